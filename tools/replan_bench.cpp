// replan_bench.cpp -- host-side cost of (re)building the render plan for a V x P additive tree (no GPU needed).
//   g++ -std=c++17 -O2 -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -o tools/_build/replan_bench tools/replan_bench.cpp
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>

#include "../libfriendship_amd/csrc/graph.cpp"
#include "../libfriendship_amd/csrc/match.cpp"
#include "../libfriendship_amd/csrc/stage.cpp"
#include "../libfriendship_amd/csrc/stagejit.cpp"

using namespace fr;
using Clock = std::chrono::steady_clock;
static double ms(Clock::time_point a) { return std::chrono::duration<double, std::milli>(Clock::now() - a).count(); }

#include <malloc.h>
int main(int argc, char **argv) {
    if (getenv("REPLAN_NOMMAP")) { mallopt(M_MMAP_THRESHOLD, 1 << 30); mallopt(M_TRIM_THRESHOLD, 1 << 30); }
    uint32_t V = argc > 1 ? atoi(argv[1]) : 64, P = argc > 2 ? atoi(argv[2]) : 4096;
    const bool effects = argc > 3 && std::string(argv[3]) == "effects";
    Mirror m;
    uint32_t next = 1;
    fr_effect prim[8]{};
    for (int k = 0; k < 7; ++k) prim[k].kind = k;
    auto node = [&](int kind) { m.add_node(next, &prim[kind]); return next++; };
    uint32_t C = node(FR_PRIM_F32CONSTANT);
    auto cst = [&](uint32_t to, float v, uint32_t slot) { m.add_edge(fr_edge{C, to, f32_to_bits(v), slot}); };
    auto t0 = Clock::now();
    std::vector<uint32_t> amp_mul;
    for (uint32_t v = 0; v < V; ++v) {
        std::vector<uint32_t> cur;
        for (uint32_t k = 0; k < P; ++k) {
            float w = 55.0f * (k + 1) / 48000.0f * (1.0f + 0.01f * v), amp = 1.0f / (k + 1);
            uint32_t x = node(FR_PRIM_MULTIPLY); m.add_edge({0, x, 0, 0}); cst(x, w, 1);
            uint32_t ph = node(FR_PRIM_MODULO); m.add_edge({x, ph, 0, 0}); cst(ph, 1.0f, 1);
            uint32_t u = node(FR_PRIM_SUM2); m.add_edge({ph, u, 0, 0}); cst(u, -0.5f, 1);
            uint32_t nu = node(FR_PRIM_MULTIPLY); cst(nu, -1.0f, 0); m.add_edge({u, nu, 0, 1});
            uint32_t mn = node(FR_PRIM_MINIMUM); m.add_edge({u, mn, 0, 0}); m.add_edge({nu, mn, 0, 1});
            uint32_t ab = node(FR_PRIM_MULTIPLY); cst(ab, -1.0f, 0); m.add_edge({mn, ab, 0, 1});
            uint32_t n1 = node(FR_PRIM_MULTIPLY); cst(n1, -1.0f, 0); m.add_edge({ab, n1, 0, 1});
            uint32_t q = node(FR_PRIM_SUM2); cst(q, 0.5f, 0); m.add_edge({n1, q, 0, 1});
            uint32_t p16 = node(FR_PRIM_MULTIPLY); cst(p16, -16.0f, 0); m.add_edge({u, p16, 0, 1});
            uint32_t y = node(FR_PRIM_MULTIPLY); m.add_edge({p16, y, 0, 0}); m.add_edge({q, y, 0, 1});
            uint32_t leaf = node(FR_PRIM_MULTIPLY); cst(leaf, amp, 0); m.add_edge({y, leaf, 0, 1});
            amp_mul.push_back(leaf);
            cur.push_back(leaf);
        }
        while (cur.size() > 1) {
            std::vector<uint32_t> nxt;
            for (size_t i = 0; i + 1 < cur.size(); i += 2) {
                uint32_t s = node(FR_PRIM_SUM2);
                m.add_edge({cur[i], s, 0, 0}); m.add_edge({cur[i + 1], s, 0, 1});
                nxt.push_back(s);
            }
            cur = nxt;
        }
        uint32_t x = cur[0];
        if (effects) {   // config D's per-voice chain: ADSR-like envelope, then 4 feed-forward delay taps
            auto bin = [&](int kind) { return node(kind); };
            uint32_t a = bin(FR_PRIM_DIVIDE); m.add_edge({0, a, 0, 0}); cst(a, 480.0f, 1);
            uint32_t rem = bin(FR_PRIM_SUM2); cst(rem, 48000.0f, 0);
            uint32_t nt = bin(FR_PRIM_MULTIPLY); cst(nt, -1.0f, 0); m.add_edge({0, nt, 0, 1}); m.add_edge({nt, rem, 0, 1});
            uint32_t rr = bin(FR_PRIM_DIVIDE); m.add_edge({rem, rr, 0, 0}); cst(rr, 4800.0f, 1);
            uint32_t mn = bin(FR_PRIM_MINIMUM); m.add_edge({a, mn, 0, 0}); m.add_edge({rr, mn, 0, 1});
            uint32_t en = bin(FR_PRIM_MULTIPLY); m.add_edge({mn, en, 0, 0}); m.add_edge({x, en, 0, 1});
            x = en;
            for (int j = 0; j < 4; ++j) {
                uint32_t dl = bin(FR_PRIM_DELAY); m.add_edge({x, dl, 0, 0}); cst(dl, 2400.0f * (j + 1), 1);
                uint32_t g = bin(FR_PRIM_MULTIPLY); cst(g, 0.5f, 0); m.add_edge({dl, g, 0, 1});
                uint32_t s = bin(FR_PRIM_SUM2); m.add_edge({x, s, 0, 0}); m.add_edge({g, s, 0, 1});
                x = s;
            }
        }
        m.add_edge({x, 0, 0, v});
    }
    std::printf("mirror build: %.1f ms (%u nodes)\n", ms(t0), next - 1);
    Lowering low;
    std::unique_ptr<BankMatcher> matcher;
    uint64_t gen = 0;
    const int reps = std::getenv("REPLAN_REPS") ? atoi(std::getenv("REPLAN_REPS")) : 6;
    for (int rep = 0; rep < reps; ++rep) {
        if (rep > 0) {   // the edit: one partial's amplitude changes (rep 0 = the initial build)
            size_t which = ((size_t)rep * 1000 + 7) % amp_mul.size();
            uint32_t leaf = amp_mul[which];
            m.del_edge(fr_edge{C, leaf, f32_to_bits(1.0f / (which % P + 1)), 0});
            m.add_edge(fr_edge{C, leaf, f32_to_bits(0.123f + rep), 0});
        }
        auto t1 = Clock::now();
        const FlatGraph &fg = low.update(m, V);
        double t_lower = ms(t1);
        if (!matcher || gen != low.generation()) { matcher.reset(new BankMatcher(fg, 20, false, true)); gen = low.generation(); }
        auto t2 = Clock::now();
        StagedPlan sp = plan_stages(fg, true, true, 20, false, true, matcher.get());
        double t_plan = ms(t2);
        auto t3 = Clock::now();
        StageJitPlan sj;
        bool jit = plan_stage_jit(sp.progs, sp.instrs, 32, false, sj);
        double t_sj = ms(t3);
        if (!sp.progs.empty()) std::printf("   stage programs %zu, instrs %zu, codegen %.2f ms (%s, %u shapes, %zu source bytes)\n", sp.progs.size(), sp.instrs.size(), t_sj, jit ? "jit" : "interp", sj.n_shapes, sj.source.size());
        size_t pbytes = 0;
        for (auto &b : sp.banks) pbytes += b.params.size() * 4;
        std::printf("%s %d: lowering %s %.2f ms (%llu nodes re-lowered, %zu flat nodes), plan_stages %.2f ms (%zu banks, %zu param bytes)\n",
                    rep ? "edit" : "build", rep, low.last_was_full() ? "full" : "incremental", t_lower, (unsigned long long)low.last_relowered(),
                    fg.nodes.size(), t_plan, sp.banks.size(), pbytes);
    }
    for (int rep = 0; rep < 4 && effects; ++rep) {   // note-on: a plain voice added as a new output slot
        std::vector<uint32_t> cur;
        for (uint32_t k = 0; k < P; ++k) {
            float w = 61.0f * (k + 1) / 48000.0f * (1.0f + 0.01f * rep), amp = 1.0f / (k + 1);
            uint32_t x = node(FR_PRIM_MULTIPLY); m.add_edge({0, x, 0, 0}); cst(x, w, 1);
            uint32_t ph = node(FR_PRIM_MODULO); m.add_edge({x, ph, 0, 0}); cst(ph, 1.0f, 1);
            uint32_t u = node(FR_PRIM_SUM2); m.add_edge({ph, u, 0, 0}); cst(u, -0.5f, 1);
            uint32_t nu = node(FR_PRIM_MULTIPLY); cst(nu, -1.0f, 0); m.add_edge({u, nu, 0, 1});
            uint32_t mn = node(FR_PRIM_MINIMUM); m.add_edge({u, mn, 0, 0}); m.add_edge({nu, mn, 0, 1});
            uint32_t ab = node(FR_PRIM_MULTIPLY); cst(ab, -1.0f, 0); m.add_edge({mn, ab, 0, 1});
            uint32_t n1 = node(FR_PRIM_MULTIPLY); cst(n1, -1.0f, 0); m.add_edge({ab, n1, 0, 1});
            uint32_t q = node(FR_PRIM_SUM2); cst(q, 0.5f, 0); m.add_edge({n1, q, 0, 1});
            uint32_t p16 = node(FR_PRIM_MULTIPLY); cst(p16, -16.0f, 0); m.add_edge({u, p16, 0, 1});
            uint32_t y = node(FR_PRIM_MULTIPLY); m.add_edge({p16, y, 0, 0}); m.add_edge({q, y, 0, 1});
            uint32_t leaf = node(FR_PRIM_MULTIPLY); cst(leaf, amp, 0); m.add_edge({y, leaf, 0, 1});
            cur.push_back(leaf);
        }
        while (cur.size() > 1) {
            std::vector<uint32_t> nxt;
            for (size_t i = 0; i + 1 < cur.size(); i += 2) {
                uint32_t s = node(FR_PRIM_SUM2);
                m.add_edge({cur[i], s, 0, 0}); m.add_edge({cur[i + 1], s, 0, 1});
                nxt.push_back(s);
            }
            cur = nxt;
        }
        m.add_edge({cur[0], 0, 0, V + rep});
        auto t1 = Clock::now();
        const FlatGraph &fg = low.update(m, V + rep + 1);
        double t_lower = ms(t1);
        auto t2 = Clock::now();
        StagedPlan sp = plan_stages(fg, true, true, 20, false, true, matcher.get());
        double t_plan = ms(t2);
        StageJitPlan sj;
        bool jit = plan_stage_jit(sp.progs, sp.instrs, 32, false, sj);
        std::printf("note-on %d: lowering %s %.2f ms (%llu nodes), plan_stages %.2f ms, %zu banks, %zu programs, jit %d shapes %u source hash %zx\n", rep,
                    low.last_was_full() ? "full" : "incremental", t_lower, (unsigned long long)low.last_relowered(), t_plan, sp.banks.size(),
                    sp.progs.size(), (int)jit, sj.n_shapes, std::hash<std::string>{}(sj.source));
    }
    {   // reference point: the same edit with from-scratch lowering and matching
        auto t1 = Clock::now();
        FlatGraph fg = lower(m, V);
        double t_lower = ms(t1);
        auto t2 = Clock::now();
        StagedPlan sp = plan_stages(fg, true, true, 20, false, true);
        std::printf("from scratch: lower %.1f ms, plan_stages %.1f ms\n", t_lower, ms(t2));
    }
    return 0;
}
