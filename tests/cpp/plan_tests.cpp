// plan_tests.cpp -- CPU tests of the engine's HOST logic (no GPU, no kernels): graph mirror, lowering, bank
// recognition, staged planning.  The lowered graph and the staged plan are executed by small interpreters written
// here (test code, mirroring what pull_kernel / stage_kernel / bank kernels do) and compared bit-for-bit with the CPU
// oracle rendering the same graph through the C ABI.  That proves the planner's output *means* what the reference
// means before any kernel runs it.
//
// Build: g++ -std=c++17 -O1 -ffp-contract=off -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -o plan_tests plan_tests.cpp -ldl
#include <dlfcn.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <random>

#include "../../libfriendship_amd/csrc/graph.cpp"
#include "../../libfriendship_amd/csrc/match.cpp"
#include "../../libfriendship_amd/csrc/stage.cpp"
#include "../../libfriendship_amd/csrc/stagejit.cpp"
#include "../../libfriendship_amd/csrc/leafjit.cpp"

using namespace fr;

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #c); throw std::runtime_error("check failed"); } } while (0)

// ---- the oracle, loaded as a plugin (same C ABI) ---------------------------------------------------------
struct Oracle {
    void *dl;
    decltype(&fr_renderer_create) create;
    decltype(&fr_renderer_destroy) destroy;
    decltype(&fr_on_add_node) add_node;
    decltype(&fr_on_add_edge) add_edge;
    decltype(&fr_fill_buffer) fill;
    Oracle() {
        const char *p = std::getenv("FRIENDSHIP_ORACLE_LIB");
        dl = dlopen(p ? p : "oracle/_build/libfr_oracle.so", RTLD_NOW | RTLD_LOCAL);
        if (!dl) throw std::runtime_error("cannot load the oracle library");
        create = (decltype(create))dlsym(dl, "fr_renderer_create");
        destroy = (decltype(destroy))dlsym(dl, "fr_renderer_destroy");
        add_node = (decltype(add_node))dlsym(dl, "fr_on_add_node");
        add_edge = (decltype(add_edge))dlsym(dl, "fr_on_add_edge");
        fill = (decltype(fill))dlsym(dl, "fr_fill_buffer");
    }
};
static Oracle &oracle() { static Oracle o; return o; }

// ---- graph builder: records primitive nodes + edges, applies them to a Mirror and to an oracle renderer ----
struct Operand { int kind; uint32_t v; };   // 0 = node handle, 1 = constant bits, 2 = input slot, 3 = unconnected
static Operand N(uint32_t h) { return {0, h}; }
static Operand Cf(float f) { return {1, f32_to_bits(f)}; }
static Operand In(uint32_t s) { return {2, s}; }
static Operand None() { return {3, 0}; }

struct Build {
    std::vector<std::pair<uint32_t, int>> nodes{{1, FR_PRIM_F32CONSTANT}};
    std::vector<fr_edge> edges;
    uint32_t next = 2;
    uint32_t op(int kind, Operand a, Operand b) {
        uint32_t h = next++;
        nodes.push_back({h, kind});
        Operand ops[2] = {a, b};
        for (uint32_t s = 0; s < 2; ++s) {
            if (ops[s].kind == 0) edges.push_back({ops[s].v, h, 0, s});
            else if (ops[s].kind == 1) edges.push_back({1, h, ops[s].v, s});
            else if (ops[s].kind == 2) edges.push_back({0, h, ops[s].v, s});
        }
        return h;
    }
    void out(Operand src, uint32_t slot) {
        if (src.kind == 0) edges.push_back({src.v, 0, 0, slot});
        else if (src.kind == 1) edges.push_back({1, 0, src.v, slot});
        else if (src.kind == 2) edges.push_back({0, 0, src.v, slot});
    }
    void apply(Mirror &m) const {
        for (auto &n : nodes) { fr_effect e{}; e.kind = n.second; m.add_node(n.first, &e); }
        for (auto &e : edges) m.add_edge(e);
    }
    fr_renderer *apply_oracle() const {
        fr_renderer *r = nullptr;
        CHECK(oracle().create(nullptr, &r) == FR_OK);
        for (auto &n : nodes) { fr_effect e{}; e.kind = n.second; CHECK(oracle().add_node(r, n.first, &e) == FR_OK); }
        for (auto &e : edges) CHECK(oracle().add_edge(r, &e) == FR_OK);
        return r;
    }
};

// the 11-node partial and the adjacent-pairs tree (same definitions as libfriendship_amd/synth.py)
static uint32_t partial(Build &b, float w, float amp) {
    uint32_t x = b.op(FR_PRIM_MULTIPLY, In(0), Cf(w));
    uint32_t ph = b.op(FR_PRIM_MODULO, N(x), Cf(1.0f));
    uint32_t u = b.op(FR_PRIM_SUM2, N(ph), Cf(-0.5f));
    uint32_t nu = b.op(FR_PRIM_MULTIPLY, Cf(-1.0f), N(u));
    uint32_t m = b.op(FR_PRIM_MINIMUM, N(u), N(nu));
    uint32_t ab = b.op(FR_PRIM_MULTIPLY, Cf(-1.0f), N(m));
    uint32_t n1 = b.op(FR_PRIM_MULTIPLY, Cf(-1.0f), N(ab));
    uint32_t q = b.op(FR_PRIM_SUM2, Cf(0.5f), N(n1));
    uint32_t p = b.op(FR_PRIM_MULTIPLY, Cf(-16.0f), N(u));
    uint32_t y = b.op(FR_PRIM_MULTIPLY, N(p), N(q));
    return b.op(FR_PRIM_MULTIPLY, Cf(amp), N(y));
}
static uint32_t sum_tree(Build &b, std::vector<uint32_t> cur) {
    while (cur.size() > 1) {
        std::vector<uint32_t> nx;
        for (size_t i = 0; i + 1 < cur.size(); i += 2) nx.push_back(b.op(FR_PRIM_SUM2, N(cur[i]), N(cur[i + 1])));
        if (cur.size() % 2) nx.push_back(cur.back());
        cur = nx;
    }
    return cur[0];
}
static uint32_t voice(Build &b, int P, float f0, std::mt19937 &rng) {
    std::vector<uint32_t> leaves;
    std::uniform_real_distribution<float> det(-0.005f, 0.005f);
    for (int k = 0; k < P; ++k) leaves.push_back(partial(b, f0 * (k + 1) * (1.0f + det(rng)) / 48000.0f, 1.0f / (k + 1)));
    return sum_tree(b, leaves);
}

// ---- interpreters of the planner's outputs ----------------------------------------------------------------
using Inputs = std::vector<std::vector<float>>;   // per slot, absolute time from 0; beyond the end -> 0
static float in_at(const Inputs &in, uint32_t slot, uint64_t t) { return slot < in.size() && t < in[slot].size() ? in[slot][t] : 0.0f; }

// value of a lowered node at time t: the reference's recursion on the flat graph
static float flat_eval(const FlatGraph &g, uint32_t id, uint64_t t, const Inputs &in) {
    const FlatNode &n = g.nodes[id];
    switch (n.op) {
    case OP_CONST: return f32_from_bits(n.a);
    case OP_INPUT: return in_at(in, n.a, t);
    case OP_DELAY: {
        float d = flat_eval(g, n.b, t, in);
        if (d >= 18446744073709551616.0f) return 0.0f;
        uint64_t fr_ = (d < 0.0f || d != d) ? 0 : (uint64_t)d;
        return fr_ > t ? 0.0f : flat_eval(g, n.a, t - fr_, in);
    }
    default: return host_binop((FlatOp)n.op, flat_eval(g, n.a, t, in), flat_eval(g, n.b, t, in));
    }
}

// the bank leaf exactly as kernels.hip's product form (EXACT = true, general fract)
static float leaf_host(float t, float w, float A4) {
    float x = t * w;
    float r = x - std::trunc(x);
    r = r < 0.0f ? r + 1.0f : r;
    float u = r - 0.5f;
    float q = 0.5f - std::fabs(u);
    float z = u * q;
    return (4.0f * A4) * z;
}
static float bank_voice_host(const BankLaunch &bl, size_t v, float t) {
    if (!bl.general) {
        size_t P = (size_t)1 << bl.log2_p;
        std::vector<float> cur(P);
        for (size_t k = 0; k < P; ++k) cur[k] = leaf_host(t, bl.params[(v * P + k) * 2], bl.params[(v * P + k) * 2 + 1]);
        while (cur.size() > 1) {
            std::vector<float> nx(cur.size() / 2);
            for (size_t i = 0; i < nx.size(); ++i) nx[i] = cur[2 * i] + cur[2 * i + 1];
            cur = nx;
        }
        return cur[0];
    }
    std::vector<float> st;
    size_t pair0 = (size_t)bl.group_off[2 * v + 1] * 8;
    for (uint32_t g = bl.group_off[2 * v]; g < bl.group_off[2 * v + 2]; ++g) {
        uint32_t j = bl.groups[g] & 15u, m = bl.groups[g] >> 4;
        CHECK(j <= GENERAL_MAX_ITEM_LOG2);
        std::vector<float> cur((size_t)1 << j);                     // the item: a complete tree over 2^j leaves
        for (size_t k = 0; k < cur.size(); ++k) cur[k] = leaf_host(t, bl.params[(pair0 + k) * 2], bl.params[(pair0 + k) * 2 + 1]);
        for (size_t k = cur.size(); k < 8; ++k) CHECK(bl.params[(pair0 + k) * 2] == 0.0f && bl.params[(pair0 + k) * 2 + 1] == 0.0f);   // padding
        pair0 += std::max<size_t>(cur.size(), 8);
        while (cur.size() > 1) {
            std::vector<float> nx(cur.size() / 2);
            for (size_t i = 0; i < nx.size(); ++i) nx[i] = cur[2 * i] + cur[2 * i + 1];
            cur = nx;
        }
        float val = cur[0];
        for (; m; --m) { val = st.back() + val; st.pop_back(); }
        st.push_back(val);
    }
    if (v + 1 < bl.rows.size()) CHECK(pair0 == (size_t)bl.group_off[2 * v + 3] * 8);
    CHECK(st.size() == 1);
    return st[0];
}

// Executes a StagedPlan call by call like engine.cpp does (rings by absolute time; level or fused form).
struct StagedSim {
    const StagedPlan &sp;
    std::vector<std::map<uint64_t, float>> rings;
    bool valid = false;
    uint64_t end = 0;
    uint64_t fused_launches = 0, strided_launches = 0;
    explicit StagedSim(const StagedPlan &p) : sp(p), rings(p.n_rings) {}

    void run_progs(uint32_t first, uint32_t count, uint64_t w0, uint64_t wlen, uint64_t idx, uint64_t T, const Inputs &in, std::vector<float> &out) {
        for (uint32_t pi = first; pi < first + count; ++pi) {
            const StageProg &pg = sp.progs[pi];
            for (uint64_t t = w0; t < w0 + wlen; ++t) {
                float tmp[STAGE_REGS] = {0};
                for (uint32_t i = 0; i < pg.n_instr; ++i) {
                    const StageInstr &x = sp.instrs[pg.first_instr + i];
                    float v = 0;
                    switch (x.op) {
                    case S_CONST: v = f32_from_bits(x.imm); break;
                    case S_INPUT: v = in_at(in, sp.input_slots[x.imm], t); break;
                    case S_READ: v = t >= x.d_lo ? rings[x.buf][t - x.d_lo] : 0.0f; break;   // (a missing entry would read 0: tests catch it)
                    case S_READ_INPUT: v = t >= x.d_lo ? in_at(in, sp.input_slots[x.imm], t - x.d_lo) : 0.0f; break;
                    case S_STEP: v = t >= x.d_lo ? f32_from_bits(x.imm) : 0.0f; break;
                    case S_STORE: rings[x.buf][t] = tmp[x.a]; continue;
                    case S_READ_DYN: case S_READ_INPUT_DYN: case S_STEP_DYN: {
                        float d = tmp[x.a];
                        v = 0.0f;
                        if (d >= 18446744073709551616.0f) break;
                        uint64_t fr_ = (d < 0.0f || d != d) ? 0 : (uint64_t)d;
                        CHECK(fr_ <= x.d_lo);   // the planner's bound holds
                        if (t < fr_) break;
                        if (x.op == S_READ_DYN) { CHECK(rings[x.buf].count(t - fr_)); v = rings[x.buf][t - fr_]; }
                        else if (x.op == S_READ_INPUT_DYN) v = in_at(in, sp.input_slots[x.imm], t - fr_);
                        else v = f32_from_bits(x.imm);
                        break;
                    }
                    case S_SUM2: v = host_binop(OP_SUM2, tmp[x.a], tmp[x.b]); break;
                    case S_MUL: v = host_binop(OP_MUL, tmp[x.a], tmp[x.b]); break;
                    case S_DIV: v = host_binop(OP_DIV, tmp[x.a], tmp[x.b]); break;
                    case S_MOD: v = host_binop(OP_MOD, tmp[x.a], tmp[x.b]); break;
                    default: v = host_binop(OP_MIN, tmp[x.a], tmp[x.b]); break;
                    }
                    tmp[x.dst] = v;
                }
                float r = tmp[pg.result_reg];
                if (pg.dst_ring != 0xFFFFFFFFu) rings[pg.dst_ring][t] = r;
                if (pg.out_row >= 0 && t >= idx) out[(size_t)pg.out_row * T + (t - idx)] = r;
            }
        }
    }

    // renders [idx, idx+T) into out [n_rows, T]; `force_levels` disables the fused form
    void call(uint64_t idx, uint64_t T, const Inputs &in, std::vector<float> &out, bool force_levels, bool *used_fused = nullptr) {
        uint64_t w0 = idx;
        if (sp.uses_rings() && !(valid && end == idx)) {
            w0 = idx > sp.lmax ? idx - sp.lmax : 0;
            for (auto &r : rings) r.clear();   // stale contents must not leak into results
            valid = false;
        }
        uint64_t wlen = idx + T - w0;
        for (const BankLaunch &bl : sp.banks)
            for (size_t v = 0; v < bl.rows.size(); ++v) {
                uint64_t b0 = bl.to_ring ? w0 : idx, blen = bl.to_ring ? wlen : T;
                for (uint64_t t = b0; t < b0 + blen; ++t) {
                    float val = bank_voice_host(bl, v, in_at(in, bl.input_slot, t));
                    if (bl.to_ring) rings[bl.rows[v]][t] = val;
                    else out[(size_t)bl.rows[v] * T + (t - idx)] = val;
                }
            }
        size_t n_levels = sp.level_first.empty() ? 0 : sp.level_first.size() - 1;
        const uint64_t fstep = std::max<uint64_t>(sp.fused_max_frames, 1);
        uint64_t n_sub = sp.fused_count ? (T - 1) / fstep + 1 : 0;
        bool fused = !force_levels && sp.fused_count && w0 == idx && valid && n_sub < n_levels;
        if (used_fused) *used_fused = fused;
        // the engine's ONE strided launch (engine.cpp execute()): frames wi, wi + stride, ... per thread.  Emulated in the
        // order least friendly to a wrong plan: programs last to first, threads last to first -- a read of a ring that another
        // program (or another thread) stores inside this launch finds nothing there yet.
        const uint64_t ssub = sp.fused_stride ? (T - 1) / sp.fused_stride + 1 : 0;
        if (!force_levels && sp.fused_count && w0 == idx && valid && sp.fused_stride >= 16 && ssub >= 2 && ssub <= 8) {
            if (used_fused) *used_fused = true;
            for (uint32_t pi = sp.fused_count; pi-- > 0;)
                for (uint64_t wi = std::min<uint64_t>(sp.fused_stride, T); wi-- > 0;)
                    for (uint64_t off = wi; off < T; off += sp.fused_stride) run_progs(sp.fused_first + pi, 1, idx + off, 1, idx, T, in, out);
            ++fused_launches;
            ++strided_launches;
        } else if (fused) {
            for (uint64_t done = 0; done < T;) {
                const uint64_t len = std::min<uint64_t>(fstep, T - done);
                run_progs(sp.fused_first, sp.fused_count, idx + done, len, idx, T, in, out);
                done += len;
                ++fused_launches;
            }
        } else {
            for (size_t l = 0; l < n_levels; ++l)
                run_progs(sp.level_first[l], sp.level_first[l + 1] - sp.level_first[l], w0, wlen, idx, T, in, out);
        }
        if (sp.uses_rings()) { valid = true; end = idx + T; }
    }
};

static bool same_bits(float a, float b) { return f32_to_bits(a) == f32_to_bits(b) || (a != a && b != b); }

// Renders the same calls on the oracle and through (lower -> plan -> simulators); compares everything.
static uint64_t g_last_strided_launches = 0;   // of the last check_graph's fused simulator
static void check_graph(const Build &b, uint32_t n_slots, uint64_t T, int calls, bool allow_banks, const char *what,
                        std::function<void(const FlatGraph &, const StagedPlan &)> inspect = nullptr) {
    Mirror m;
    b.apply(m);
    FlatGraph fg = lower(m, n_slots);
    StagedPlan sp = plan_stages(fg, allow_banks, true, 20);
    if (inspect) inspect(fg, sp);
    fr_renderer *ref = b.apply_oracle();
    StagedSim sim_levels(sp), sim_fused(sp);
    Inputs hist(2);
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.0f, 3.0f);
    bool any_fused = false;
    for (int c = 0; c < calls; ++c) {
        uint64_t idx = (uint64_t)c * T;
        std::vector<float> row0(T), row1(T);
        for (uint64_t i = 0; i < T; ++i) { row0[i] = (float)(idx + i); row1[i] = nd(rng); }
        hist[0].insert(hist[0].end(), row0.begin(), row0.end());
        hist[1].insert(hist[1].end(), row1.begin(), row1.end());
        std::vector<float> data(row0);
        data.insert(data.end(), row1.begin(), row1.end());
        uint64_t offs[3] = {0, T, 2 * T};
        std::vector<float> exp((size_t)n_slots * T), a((size_t)n_slots * T, -77.0f), f2((size_t)n_slots * T, -77.0f);
        CHECK(oracle().fill(ref, exp.data(), n_slots, T, idx, data.data(), offs, 2) == FR_OK);
        bool uf = false;
        sim_levels.call(idx, T, hist, a, true);
        sim_fused.call(idx, T, hist, f2, false, &uf);
        any_fused = any_fused || uf;
        for (uint32_t s = 0; s < n_slots; ++s) {
            bool staged_row = std::find(sp.pull_rows.begin(), sp.pull_rows.end(), s) == sp.pull_rows.end();
            for (uint64_t i = 0; i < T; ++i) {
                float e = exp[(size_t)s * T + i];
                float fl = flat_eval(fg, fg.outputs[s], idx + i, hist);
                if (!same_bits(fl, e)) { std::fprintf(stderr, "%s: lowered graph differs from oracle at slot %u t %llu: %a vs %a\n", what, s, (unsigned long long)(idx + i), fl, e); throw std::runtime_error("mismatch"); }
                if (!staged_row) continue;
                if (!same_bits(a[(size_t)s * T + i], e) || !same_bits(f2[(size_t)s * T + i], e)) {
                    std::fprintf(stderr, "%s: staged plan differs from oracle at slot %u t %llu: levels %a fused %a oracle %a\n", what, s,
                                 (unsigned long long)(idx + i), a[(size_t)s * T + i], f2[(size_t)s * T + i], e);
                    throw std::runtime_error("mismatch");
                }
            }
        }
    }
    oracle().destroy(ref);
    (void)any_fused;
    g_last_strided_launches = sim_fused.strided_launches;
}

// ---- tests ---------------------------------------------------------------------------------------------------
static void lowering_folds_constants() {
    Build b;
    uint32_t mul = b.op(FR_PRIM_MULTIPLY, Cf(0.5f), Cf(-3.0f));
    b.out(N(mul), 0);
    uint32_t dl = b.op(FR_PRIM_DELAY, Cf(0.5f), Cf(2.0f));       // a unit step is NOT a constant
    b.out(N(dl), 1);
    uint32_t d0 = b.op(FR_PRIM_DELAY, In(0), Cf(0.0f));          // zero delay folds away
    b.out(N(d0), 2);
    uint32_t big = b.op(FR_PRIM_DELAY, In(0), Cf(3.0e19f));      // >= 2^64 -> 0
    b.out(N(big), 3);
    Mirror m;
    b.apply(m);
    FlatGraph fg = lower(m, 5);
    CHECK(fg.is_const(fg.outputs[0], -1.5f));
    CHECK(fg.nodes[fg.outputs[1]].op == OP_DELAY);
    CHECK(fg.nodes[fg.outputs[2]].op == OP_INPUT);
    CHECK(fg.is_const(fg.outputs[3], 0.0f));
    CHECK(fg.is_const(fg.outputs[4], 0.0f));                      // unconnected output slot
    check_graph(b, 5, 8, 2, true, "folding");
}

static void lowering_errors() {
    auto code_of = [](const Build &b, uint32_t slots) {
        Mirror m;
        b.apply(m);
        try { lower(m, slots); } catch (const Error &e) { return (int)e.code; }
        return (int)FR_OK;
    };
    {   // a cycle the reference's RouteGraph is documented to reject
        Build b;
        uint32_t s = b.op(FR_PRIM_SUM2, None(), Cf(1.0f));
        b.edges.push_back({s, s, 0, 0});
        b.out(N(s), 0);
        CHECK(code_of(b, 1) == FR_ERR_CYCLE);
        CHECK(code_of(b, 0) == FR_OK);   // not reachable from a rendered slot: the reference never evaluates it either
    }
    {   // edge from a node the renderer was never told about (reference.rs:186 panics when it is evaluated)
        Build b;
        b.edges.push_back({99, 0, 0, 0});
        CHECK(code_of(b, 1) == FR_ERR_NO_SUCH_NODE);
    }
    {   // primitive read through output slot 1 (reference.rs:223 assert)
        Build b;
        uint32_t s = b.op(FR_PRIM_SUM2, Cf(1.0f), Cf(2.0f));
        b.edges.push_back({s, 0, 1, 0});
        CHECK(code_of(b, 1) == FR_ERR_BAD_SLOT);
    }
    {   // add_edge into an unknown node (reference.rs:145 unwrap)
        Mirror m;
        bool threw = false;
        try { m.add_edge({0, 42, 0, 0}); } catch (const Error &e) { threw = e.code == FR_ERR_NO_SUCH_NODE; }
        CHECK(threw);
    }
}

static void random_graphs_lower_correctly() {
    for (unsigned seed = 0; seed < 40; ++seed) {
        std::mt19937 rng(seed);
        Build b;
        std::vector<uint32_t> avail;
        int kinds[6] = {FR_PRIM_DELAY, FR_PRIM_SUM2, FR_PRIM_MULTIPLY, FR_PRIM_DIVIDE, FR_PRIM_MODULO, FR_PRIM_MINIMUM};
        float consts[10] = {0.0f, 1.0f, -1.0f, 0.5f, 2.0f, 3.0f, -3.5f, 7.25f, 1e-30f, NAN};
        int n = 4 + (int)(rng() % 24);
        for (int i = 0; i < n; ++i) {
            int k = kinds[rng() % 6];
            auto pick = [&](bool delay_amount) -> Operand {
                unsigned r = rng() % 10;
                if (delay_amount && (seed % 2 == 0 || r < 7)) return Cf((float)(rng() % 9));   // even seeds: constant delays only
                if (r < 1) return None();
                if (r < 3 || avail.empty()) return (rng() % 2) ? In(rng() % 2) : Cf(consts[rng() % 10]);
                return N(avail[rng() % avail.size()]);
            };
            Operand a = pick(false), bb = pick(k == FR_PRIM_DELAY);
            avail.push_back(b.op(k, a, bb));
        }
        for (uint32_t s = 0; s < 3; ++s) b.out(N(avail[avail.size() - 1 - (rng() % std::min<size_t>(avail.size(), 6))]), s);
        char what[32];
        std::snprintf(what, sizeof what, "random seed %u", seed);
        check_graph(b, 3, 24, 3, true, what);
    }
}

static void banks_are_recognised() {
    std::mt19937 rng(3);
    Build b;
    b.out(N(voice(b, 64, 55.0f, rng)), 0);
    b.out(N(voice(b, 64, 110.0f, rng)), 1);
    b.out(N(voice(b, 32, 220.0f, rng)), 2);
    b.out(N(voice(b, 100, 82.4f, rng)), 3);   // not a power of two: general schedule
    b.out(N(voice(b, 8, 82.4f, rng)), 4);     // too small to be a bank: a stage program (89 instructions)
    check_graph(b, 5, 16, 2, true, "banks", [](const FlatGraph &, const StagedPlan &sp) {
        CHECK(sp.pull_rows.empty());
        size_t balanced = 0, general = 0;
        for (auto &bl : sp.banks) (bl.general ? general : balanced) += bl.rows.size();
        CHECK(balanced == 3 && general == 1);
        CHECK(sp.progs.size() == 1 && sp.n_rings == 0);
        for (auto &bl : sp.banks)
            if (bl.general) {
                CHECK(bl.max_leaves == 100);
                uint32_t leaves = 0, merges = 0;
                for (uint32_t g : bl.groups) { leaves += 1u << (g & 15u); merges += g >> 4; }
                CHECK(leaves == 100 && merges + 1 == bl.groups.size());
            }
    });
    check_graph(b, 5, 16, 1, false, "banks off");   // same graph with banks disabled: 64-leaf trees exceed a program -> pull
}

static void effects_chain_is_staged() {
    std::mt19937 rng(9);
    Build b;
    for (uint32_t v = 0; v < 2; ++v) {
        uint32_t mix = voice(b, 32, 55.0f * (v + 1), rng);
        // envelope: max(0, min(t/20, (200 - t)/50)) built with Minimum and negations
        uint32_t a = b.op(FR_PRIM_DIVIDE, In(0), Cf(20.0f));
        uint32_t r = b.op(FR_PRIM_DIVIDE, N(b.op(FR_PRIM_SUM2, Cf(200.0f), N(b.op(FR_PRIM_MULTIPLY, Cf(-1.0f), In(0))))), Cf(50.0f));
        uint32_t mn = b.op(FR_PRIM_MINIMUM, N(a), N(r));
        uint32_t env = b.op(FR_PRIM_MULTIPLY, Cf(-1.0f), N(b.op(FR_PRIM_MINIMUM, Cf(0.0f), N(b.op(FR_PRIM_MULTIPLY, Cf(-1.0f), N(mn))))));
        uint32_t x = b.op(FR_PRIM_MULTIPLY, N(env), N(mix));
        for (int j = 0; j < 3; ++j) {
            uint32_t dl = b.op(FR_PRIM_DELAY, N(x), Cf(70.0f * (j + 1)));
            x = b.op(FR_PRIM_SUM2, N(x), N(b.op(FR_PRIM_MULTIPLY, Cf(0.5f), N(dl))));
        }
        b.out(N(x), v);
    }
    uint32_t dn = b.op(FR_PRIM_DELAY, In(1), Cf(5.0f));            // a delay of an input and of a constant
    b.out(N(b.op(FR_PRIM_SUM2, N(dn), N(b.op(FR_PRIM_DELAY, Cf(2.0f), Cf(3.0f))))), 2);
    check_graph(b, 3, 100, 5, true, "effects chain", [](const FlatGraph &, const StagedPlan &sp) {
        CHECK(sp.pull_rows.empty());
        CHECK(sp.lmax == 70 + 140 + 210);
        CHECK(sp.n_rings == 2 * 4);                 // per voice: the bank's mix + x0, x1, x2
        CHECK(sp.fused_count > 0 && sp.fused_max_frames == 70);
        CHECK(sp.fused_stride == 70);               // the taps' delays 70, 140, 210 read rings their own program stores
        CHECK(sp.level_first.size() - 1 >= 5);
    });
    CHECK(g_last_strided_launches == 4);            // calls of 100 frames > 70: the steady ones are ONE strided launch each
    check_graph(b, 3, 33, 7, true, "effects chain, short calls");
    CHECK(g_last_strided_launches == 0);
    check_graph(b, 3, 500, 3, true, "effects chain, 8 strides per call");
    CHECK(g_last_strided_launches == 2);
    {   // a tap of another row's chain: row 1 reads x0 of row 0's voice 70 frames back -- a ring ANOTHER fused program stores.
        // No stride is valid for that (two threads, no order inside one launch): sub-window launches as before.
        Build c;
        std::mt19937 rng2(10);
        uint32_t x = c.op(FR_PRIM_MULTIPLY, N(voice(c, 32, 110.0f, rng2)), N(c.op(FR_PRIM_DIVIDE, In(0), Cf(50.0f))));
        uint32_t d0 = c.op(FR_PRIM_DELAY, N(x), Cf(70.0f));
        c.out(N(c.op(FR_PRIM_SUM2, N(x), N(c.op(FR_PRIM_MULTIPLY, Cf(0.5f), N(d0))))), 0);
        uint32_t z = c.op(FR_PRIM_MULTIPLY, N(voice(c, 32, 220.0f, rng2)), Cf(0.25f));
        uint32_t d1 = c.op(FR_PRIM_DELAY, N(x), Cf(140.0f));
        c.out(N(c.op(FR_PRIM_SUM2, N(z), N(d1))), 1);
        check_graph(c, 2, 100, 5, true, "tap of another row's chain", [](const FlatGraph &, const StagedPlan &sp) {
            CHECK(sp.pull_rows.empty());
            if (sp.fused_count) CHECK(sp.fused_stride == 0);
        });
        CHECK(g_last_strided_launches == 0);
    }
}

// A non-bank root wired to two output rows with no delayed read of a program ring: the fused form has no frame limit
// (fused_max_frames used to stay at ~0 and the engine's sub-window arithmetic wrapped: idx + 1 launches per call).
static void shared_root_without_delays_is_one_launch() {
    Build b;
    uint32_t g = b.op(FR_PRIM_SUM2, N(b.op(FR_PRIM_MULTIPLY, In(0), Cf(0.5f))), In(1));   // a master gain sent to L and R
    b.out(N(g), 0);
    b.out(N(g), 1);
    Mirror m;
    b.apply(m);
    FlatGraph fg = lower(m, 2);
    StagedPlan sp = plan_stages(fg, true, true, 20);
    CHECK(sp.fused_count > 0);
    CHECK(sp.fused_max_frames >= 4800 && sp.fused_max_frames <= (1ull << 40));
    StagedSim sim(sp);
    Inputs hist(2);
    const uint64_t T = 4800;
    for (int c = 0; c < 3; ++c) {
        uint64_t idx = (uint64_t)c * T;
        for (uint64_t i = 0; i < T; ++i) { hist[0].push_back((float)(idx + i)); hist[1].push_back((float)(i % 7)); }
        std::vector<float> out(2 * T, -1.0f);
        bool uf = false;
        const uint64_t before = sim.fused_launches;
        sim.call(idx, T, hist, out, false, &uf);
        if (c > 0) { CHECK(uf); CHECK(sim.fused_launches - before == 1); }   // steady state: exactly one launch
        for (uint64_t i = 0; i < T; ++i) {
            float e = flat_eval(fg, fg.outputs[0], idx + i, hist);
            CHECK(same_bits(out[i], e) && same_bits(out[T + i], e));
        }
    }
    check_graph(b, 2, 64, 4, true, "shared root");
}

static void dynamic_delay_goes_to_pull() {
    Build b;
    uint32_t amt = b.op(FR_PRIM_MULTIPLY, In(0), Cf(0.25f));   // no bound can be proven for this amount
    uint32_t src = b.op(FR_PRIM_MULTIPLY, In(1), Cf(2.0f));    // a computed signal: delaying it needs a ring, hence a bound
    b.out(N(b.op(FR_PRIM_DELAY, N(src), N(amt))), 0);
    b.out(N(b.op(FR_PRIM_SUM2, In(1), Cf(1.0f))), 1);
    b.out(N(b.op(FR_PRIM_DELAY, In(1), N(amt))), 2);           // an input delayed by the same amount: read from the history, no bound needed
    b.out(N(b.op(FR_PRIM_DELAY, Cf(4.5f), N(amt))), 3);        // a constant delayed by it: a step at t = amount
    check_graph(b, 4, 20, 3, true, "dynamic delay", [](const FlatGraph &, const StagedPlan &sp) {
        CHECK(sp.pull_rows.size() == 1 && sp.pull_rows[0] == 0);
        CHECK(sp.progs.size() == 3);
    });
}

static void composite_instances_are_interned() {
    // MulBy2-style composite used twice: one definition, two instances, inlined correctly
    fr_effect mul{}, cst{};
    mul.kind = FR_PRIM_MULTIPLY;
    cst.kind = FR_PRIM_F32CONSTANT;
    uint32_t handles[2] = {1, 2};
    const fr_effect *effs[2] = {&mul, &cst};
    fr_edge edges[3] = {{0, 1, 0, 0}, {1, 0, 0, 0}, {2, 1, f32_to_bits(5.0f), 1}};
    fr_effect comp{};
    comp.kind = FR_EFFECT_GRAPH;
    comp.n_nodes = 2; comp.node_handles = handles; comp.node_effects = effs;
    comp.n_edges = 3; comp.edges = edges;
    Mirror m;
    m.add_node(10, &comp);
    m.add_node(11, &comp);
    CHECK(m.nodes.at(10).sub == m.nodes.at(11).sub);        // interned once
    m.add_edge({0, 10, 0, 0});
    m.add_edge({10, 11, 0, 0});
    m.add_edge({11, 0, 0, 0});
    FlatGraph fg = lower(m, 1);
    Inputs in{{1.0f, 2.0f, 3.0f}};
    for (uint64_t t = 0; t < 3; ++t) CHECK(flat_eval(fg, fg.outputs[0], t, in) == 25.0f * (t + 1));
}


// ---- compiled stage programs (stagejit.cpp) on the CPU -------------------------------------------------------------
// The generated kernel source is plain C++ apart from a few HIP keywords: with those defined away it compiles with
// g++ and jit_stage() can be driven thread by thread.  This checks the code generator (skeleton grouping, parameter
// rows, SSA wiring, literal peepholes) against the oracle without a GPU; the GPU tests check hipRTC's build of it.
static const char *kCpuPrelude = R"CPU(
#include <cmath>
#include <cstddef>
#define __device__
#define __forceinline__ inline
#define __global__
#define __launch_bounds__(x)
struct Idx3 { unsigned x, y, z; };
static Idx3 blockIdx, threadIdx;
extern "C" void set_thread(unsigned bx, unsigned by, unsigned tx) { blockIdx = Idx3{bx, by, 0}; threadIdx = Idx3{tx, 0, 0}; }
)CPU";

struct CpuJit {
    void *so = nullptr;
    void (*set_thread)(unsigned, unsigned, unsigned) = nullptr;
    void (*kernel)(JitStageArgs) = nullptr;
    explicit CpuJit(const std::string &source) {
        static int serial = 0;
        std::string base = "/tmp/fr_stagejit_" + std::to_string((long)getpid()) + "_" + std::to_string(serial++);
        FILE *f = std::fopen((base + ".cpp").c_str(), "w");
        CHECK(f != nullptr);
        std::fputs(kCpuPrelude, f);
        std::fputs(source.c_str(), f);
        std::fclose(f);
        std::string cmd = "g++ -std=c++20 -O1 -ffp-contract=off -w -shared -fPIC -o " + base + ".so " + base + ".cpp";
        CHECK(std::system(cmd.c_str()) == 0);
        so = dlopen((base + ".so").c_str(), RTLD_NOW | RTLD_LOCAL);
        CHECK(so != nullptr);
        set_thread = (void (*)(unsigned, unsigned, unsigned))dlsym(so, "set_thread");
        kernel = (void (*)(JitStageArgs))dlsym(so, "jit_stage");
        CHECK(set_thread && kernel);
        std::remove((base + ".cpp").c_str());
        std::remove((base + ".so").c_str());
    }
    ~CpuJit() { if (so) dlclose(so); }
};

// engine.cpp's execute() for the compiled form: real ring arrays (power-of-two capacity, indexed t & mask).
struct JitSim {
    const StagedPlan &sp;
    StageJitPlan sj;
    std::unique_ptr<CpuJit> jit;
    uint64_t cap = 0;
    std::vector<float> rings;
    bool valid = false;
    uint64_t end = 0;
    JitSim(const StagedPlan &p, uint64_t T) : sp(p) {
        CHECK(plan_stage_jit(sp.progs, sp.instrs, 64, true, sj));
        jit.reset(new CpuJit(sj.source));
        cap = 1024;
        while (cap < sp.lmax + T) cap <<= 1;
        rings.assign((size_t)std::max<uint32_t>(sp.n_rings, 1) * cap, -55.0f);
    }
    void launch(uint32_t first, uint32_t count, uint64_t w0, uint64_t wlen, uint64_t idx, uint64_t T, const Inputs &in, std::vector<float> &out) {
        JitStageArgs a{};
        a.ptab = sj.ptab.data();
        a.progs = sj.progs.data() + first;
        a.rings = rings.data();
        a.ring_mask = cap - 1;
        a.n_inputs = (uint32_t)sp.input_slots.size();
        CHECK(a.n_inputs <= 8);
        for (uint32_t i = 0; i < a.n_inputs; ++i) {
            uint32_t slot = sp.input_slots[i];
            if (slot < in.size()) a.inline_inputs[i] = JitInput{in[slot].data(), 0, in[slot].size()};
        }
        a.out = out.data();
        a.n_times = T; a.idx = idx; a.w0 = w0; a.w_len = wlen;
        for (uint32_t y = 0; y < count; ++y)
            for (uint32_t bx = 0; bx < (wlen + 255) / 256; ++bx)
                for (uint32_t tx = 0; tx < 256; ++tx) { jit->set_thread(bx, y, tx); jit->kernel(a); }
    }
    void call(uint64_t idx, uint64_t T, const Inputs &in, std::vector<float> &out) {
        uint64_t w0 = idx;
        if (sp.uses_rings() && !(valid && end == idx)) w0 = idx > sp.lmax ? idx - sp.lmax : 0;
        uint64_t wlen = idx + T - w0;
        for (const BankLaunch &bl : sp.banks)
            for (size_t v = 0; v < bl.rows.size(); ++v) {
                uint64_t b0 = bl.to_ring ? w0 : idx, blen = bl.to_ring ? wlen : T;
                for (uint64_t t = b0; t < b0 + blen; ++t) {
                    float val = bank_voice_host(bl, v, in_at(in, bl.input_slot, t));
                    if (bl.to_ring) rings[(size_t)bl.rows[v] * cap + (t & (cap - 1))] = val;
                    else out[(size_t)bl.rows[v] * T + (t - idx)] = val;
                }
            }
        size_t n_levels = sp.level_first.empty() ? 0 : sp.level_first.size() - 1;
        const uint64_t fstep = std::max<uint64_t>(sp.fused_max_frames, 1);
        uint64_t n_sub = sp.fused_count ? (T - 1) / fstep + 1 : 0;
        bool fused = sp.fused_count && w0 == idx && valid && n_sub < n_levels;
        if (fused) {
            for (uint64_t done = 0; done < T;) {
                const uint64_t len = std::min<uint64_t>(fstep, T - done);
                launch(sp.fused_first, sp.fused_count, idx + done, len, idx, T, in, out);
                done += len;
            }
        } else {
            for (size_t l = 0; l < n_levels; ++l)
                launch(sp.level_first[l], sp.level_first[l + 1] - sp.level_first[l], w0, wlen, idx, T, in, out);
        }
        if (sp.uses_rings()) { valid = true; end = idx + T; }
    }
};

static void check_compiled_programs(const Build &b, uint32_t n_slots, uint64_t T, const std::vector<uint64_t> &starts, const char *what,
                                    std::function<void(const StageJitPlan &)> inspect = nullptr) {
    Mirror m;
    b.apply(m);
    FlatGraph fg = lower(m, n_slots);
    StagedPlan sp = plan_stages(fg, true, true, 20);
    CHECK(!sp.progs.empty());
    JitSim sim(sp, T);
    if (std::getenv("FR_TEST_VERBOSE")) std::fprintf(stderr, "%s: progs %zu fused %u levels %zu shapes %u\n", what, sp.progs.size(), sp.fused_count, sp.level_first.size() - 1, sim.sj.n_shapes);
    if (inspect) inspect(sim.sj);
    fr_renderer *ref = b.apply_oracle();
    Inputs hist(2);
    std::mt19937 rng(11);
    std::normal_distribution<float> nd(0.0f, 3.0f);
    for (uint64_t idx : starts) {
        std::vector<float> row0(T), row1(T);
        for (uint64_t i = 0; i < T; ++i) { row0[i] = (float)(idx + i); row1[i] = nd(rng); }
        for (auto &h : hist) h.resize(idx, 0.0f);   // a seek leaves zeros behind (reference.rs:52-60); starts only grow here
        hist[0].insert(hist[0].end(), row0.begin(), row0.end());
        hist[1].insert(hist[1].end(), row1.begin(), row1.end());
        std::vector<float> data(row0);
        data.insert(data.end(), row1.begin(), row1.end());
        uint64_t offs[3] = {0, T, 2 * T};
        std::vector<float> exp((size_t)n_slots * T), got((size_t)n_slots * T, -77.0f);
        CHECK(oracle().fill(ref, exp.data(), n_slots, T, idx, data.data(), offs, 2) == FR_OK);
        sim.call(idx, T, hist, got);
        for (uint32_t s = 0; s < n_slots; ++s) {
            if (std::find(sp.pull_rows.begin(), sp.pull_rows.end(), s) != sp.pull_rows.end()) continue;
            for (uint64_t i = 0; i < T; ++i)
                if (!same_bits(got[(size_t)s * T + i], exp[(size_t)s * T + i])) {
                    std::fprintf(stderr, "%s: compiled programs differ from oracle at slot %u t %llu: %a vs %a\n", what, s,
                                 (unsigned long long)(idx + i), got[(size_t)s * T + i], exp[(size_t)s * T + i]);
                    throw std::runtime_error("mismatch");
                }
        }
    }
    oracle().destroy(ref);
}

static void stage_programs_compile_to_source() {
    // 6 voices through the same chain: one skeleton per chain position (level form) + the fused form's
    std::mt19937 rng(13);
    Build b;
    const uint32_t V = 6;
    for (uint32_t v = 0; v < V; ++v) {
        uint32_t mix = voice(b, 16, 55.0f * (v + 1), rng);
        uint32_t lfo = b.op(FR_PRIM_MODULO, N(b.op(FR_PRIM_MULTIPLY, In(0), Cf(0.01f * (v + 1)))), Cf(1.0f));   // x mod 1.0 peephole
        uint32_t a = b.op(FR_PRIM_MINIMUM, N(b.op(FR_PRIM_DIVIDE, In(0), Cf(20.0f + v))), N(lfo));
        uint32_t x = b.op(FR_PRIM_MULTIPLY, N(a), N(mix));
        for (int j = 0; j < 3; ++j) {
            uint32_t dl = b.op(FR_PRIM_DELAY, N(x), Cf(70.0f * (j + 1) + v));
            x = b.op(FR_PRIM_SUM2, N(x), N(b.op(FR_PRIM_MULTIPLY, Cf(-1.0f), N(dl))));
        }
        b.out(N(x), v);
    }
    uint32_t dn = b.op(FR_PRIM_DELAY, In(1), Cf(5.0f));
    b.out(N(b.op(FR_PRIM_MODULO, N(b.op(FR_PRIM_SUM2, N(dn), N(b.op(FR_PRIM_DELAY, Cf(2.0f), Cf(3.0f))))), Cf(0.75f))), V);
    check_compiled_programs(b, V + 1, 90, {0, 90, 180, 270, 2000, 2090}, "compiled effects chain", [&](const StageJitPlan &sj) {
        CHECK(sj.n_shapes == 4);                                            // x0, x1..x3 (one skeleton), the fused sink, the odd row
        CHECK(sj.source.find("jit_mod1(") != std::string::npos);            // Modulo(x, 1.0) was recognised
        CHECK(sj.source.find("jit_mod(") != std::string::npos);             // Modulo(x, 0.75) was not
        CHECK(sj.source.find("f32(0xbf800000u)") != std::string::npos);     // -1.0 shared by all members: a literal
    });
    check_compiled_programs(b, V + 1, 30, {0, 30, 60, 90, 120}, "compiled effects chain, calls shorter than the delays");
    // random graphs: every program its own skeleton
    for (int seed = 0; seed < 12; ++seed) {
        std::mt19937 r2(400 + seed);
        Build g;
        std::vector<uint32_t> pool;
        auto pick = [&]() -> Operand {
            uint32_t k = r2() % 10;
            if (k < 2 || pool.empty()) return k % 2 ? In(r2() % 2) : Cf((float)((int)(r2() % 41) - 20) * 0.25f);
            return N(pool[r2() % pool.size()]);
        };
        const int kinds[5] = {FR_PRIM_SUM2, FR_PRIM_MULTIPLY, FR_PRIM_DIVIDE, FR_PRIM_MODULO, FR_PRIM_MINIMUM};
        for (int i = 0; i < 30; ++i) {
            if (r2() % 5 == 0) pool.push_back(g.op(FR_PRIM_DELAY, pick(), Cf((float)(r2() % 50))));
            else pool.push_back(g.op(kinds[r2() % 5], pick(), pick()));
        }
        for (uint32_t s = 0; s < 3; ++s) g.out(N(pool[pool.size() - 1 - s]), s);
        check_compiled_programs(g, 3, 64, {0, 64, 128, 900}, "compiled random graph");
    }
}

// ---- incremental lowering (graph.hpp Lowering) ----------------------------------------------------------------------
// A Lowering kept across random edit sequences must describe the same function as lowering the mirror from scratch
// after every batch of edits: same error (if the graph is broken at that point), else bit-identical output values.
static void incremental_lowering_equals_from_scratch() {
    const int kinds[6] = {FR_PRIM_DELAY, FR_PRIM_SUM2, FR_PRIM_MULTIPLY, FR_PRIM_DIVIDE, FR_PRIM_MODULO, FR_PRIM_MINIMUM};
    const float consts[8] = {0.0f, 1.0f, -1.0f, 0.5f, 2.0f, 3.0f, -3.5f, 7.25f};
    // composite: out0 = in0 * C(5) + in1, out1 = in1 (a pure pass-through: readers reach the instance's inbound edge directly)
    fr_effect mul{}, add{}, cst{};
    mul.kind = FR_PRIM_MULTIPLY; add.kind = FR_PRIM_SUM2; cst.kind = FR_PRIM_F32CONSTANT;
    uint32_t ch[3] = {1, 2, 3};
    const fr_effect *ce[3] = {&mul, &cst, &add};
    fr_edge cedges[6] = {{0, 1, 0, 0}, {2, 1, f32_to_bits(5.0f), 1}, {1, 3, 0, 0}, {0, 3, 1, 1}, {3, 0, 0, 0}, {0, 0, 1, 1}};
    fr_effect comp{};
    comp.kind = FR_EFFECT_GRAPH;
    comp.n_nodes = 3; comp.node_handles = ch; comp.node_effects = ce; comp.n_edges = 6; comp.edges = cedges;
    // a second definition for "replace the instance by another effect under the same handle": out0 = in0 + in1
    fr_edge c2edges[3] = {{0, 3, 0, 0}, {0, 3, 1, 1}, {3, 0, 0, 0}};
    uint32_t c2h[1] = {3};
    const fr_effect *c2e[1] = {&add};
    fr_effect comp2{};
    comp2.kind = FR_EFFECT_GRAPH;
    comp2.n_nodes = 1; comp2.node_handles = c2h; comp2.node_effects = c2e; comp2.n_edges = 3; comp2.edges = c2edges;

    const uint32_t CONST = 100000, N_OUT = 3;
    uint64_t n_incremental = 0, n_full = 0, n_errors = 0;
    for (unsigned seed = 0; seed < 30; ++seed) {
        std::mt19937 rng(900 + seed);
        Mirror m;
        Lowering low;
        fr_effect prim{};
        prim.kind = FR_PRIM_F32CONSTANT;
        m.add_node(CONST, &prim);
        std::vector<uint32_t> live;          // handles in creation order; edges only go from earlier to later handles
        std::map<uint32_t, bool> is_comp;
        uint32_t next = 1;
        auto source = [&](uint32_t before_handle, uint32_t &from, uint32_t &from_slot) {
            unsigned r = rng() % 10;
            std::vector<uint32_t> cand;
            for (uint32_t h : live) if (h < before_handle) cand.push_back(h);
            if (r < 2 || cand.empty()) {
                if (rng() % 2) { from = 0; from_slot = rng() % 2; }
                else { from = CONST; from_slot = f32_to_bits(consts[rng() % 8]); }
                return;
            }
            from = cand[rng() % cand.size()];
            from_slot = is_comp[from] ? rng() % 2 : 0;
        };
        auto connect = [&](uint32_t h, uint32_t slot) {
            uint32_t from, fs;
            source(h, from, fs);
            if (!is_comp[h] && slot == 1 && rng() % 3 == 0) { from = CONST; fs = f32_to_bits((float)(rng() % 7)); }   // constant delay amounts etc.
            m.add_edge(fr_edge{from, h, fs, slot});
        };
        auto edit = [&]() {
            unsigned r = rng() % 12;
            if (live.size() < 4) r = 0;
            if (r < 3) {                                   // new primitive node
                uint32_t h = next++;
                prim.kind = kinds[rng() % 6];
                m.add_node(h, &prim);
                is_comp[h] = false;
                connect(h, 0); connect(h, 1);
                live.push_back(h);
            } else if (r < 4) {                            // new composite instance
                uint32_t h = next++;
                m.add_node(h, &comp);
                is_comp[h] = true;
                connect(h, 0); connect(h, 1);
                live.push_back(h);
            } else if (r < 7) {                            // rewire one inbound edge
                uint32_t h = live[rng() % live.size()];
                connect(h, rng() % 2);
            } else if (r < 8) {                            // drop one inbound edge
                uint32_t h = live[rng() % live.size()];
                m.del_edge(fr_edge{0, h, 0, (uint32_t)(rng() % 2)});
            } else if (r < 9 && rng() % 3 == 0) {          // delete a node (its readers now dangle), sometimes bring the handle back
                size_t i = rng() % live.size();
                uint32_t h = live[i];
                m.del_node(h);
                if (rng() % 2) {
                    bool c = rng() % 3 == 0;
                    prim.kind = kinds[rng() % 6];
                    m.add_node(h, c ? (rng() % 2 ? &comp : &comp2) : &prim);
                    is_comp[h] = c;
                    connect(h, 0); connect(h, 1);
                } else {
                    live.erase(live.begin() + i);
                }
            } else if (r < 10) {                           // replace a node in place (HashMap::insert semantics), inbound edges start empty
                uint32_t h = live[rng() % live.size()];
                for (int tries = 0; tries < 4 && !is_comp[h]; ++tries) h = live[rng() % live.size()];   // prefer instances:
                bool c = is_comp[h] ? rng() % 4 != 0 : rng() % 3 == 0;                                  // swap their definition
                prim.kind = kinds[rng() % 6];
                m.add_node(h, c ? (rng() % 2 ? &comp : &comp2) : &prim);
                is_comp[h] = c;
                connect(h, 0);
                if (rng() % 2) connect(h, 1);
            } else {                                       // output edge
                uint32_t from, fs;
                source(~0u, from, fs);
                m.add_edge(fr_edge{from, 0, fs, (uint32_t)(rng() % N_OUT)});
            }
        };
        Inputs in(2);
        std::normal_distribution<float> nd(0.0f, 3.0f);
        for (int t = 0; t < 48; ++t) { in[0].push_back((float)t); in[1].push_back(nd(rng)); }
        for (int step = 0; step < 60; ++step) {
            int n_edits = 1 + (int)(rng() % 3);
            if (step == 0) n_edits = 12;
            for (int e = 0; e < n_edits; ++e) edit();
            bool expect_full = step == 0;
            if (seed % 4 == 3 && step == 30) {             // the constant node itself is replaced: not journalled per reader
                expect_full = true;
                m.del_node(CONST);
                prim.kind = FR_PRIM_F32CONSTANT;
                m.add_node(CONST, &prim);
            }
            if (seed % 4 == 0 && step == 30) {             // more edits than the journal holds
                expect_full = true;
                for (unsigned i = 0; i < Mirror::JOURNAL_LIMIT + 10; ++i) connect(live[i % live.size()], i % 2);
            }
            int code_inc = 0, code_fresh = 0;
            const FlatGraph *inc = nullptr;
            FlatGraph fresh;
            try { inc = &low.update(m, N_OUT); } catch (const Error &e) { code_inc = e.code; }
            try { fresh = lower(m, N_OUT); } catch (const Error &e) { code_fresh = e.code; }
            if (code_inc != code_fresh) {
                std::fprintf(stderr, "seed %u step %d: incremental lowering status %d, from scratch %d\n", seed, step, code_inc, code_fresh);
                throw std::runtime_error("status mismatch");
            }
            if (code_inc) { ++n_errors; continue; }
            (low.last_was_full() ? n_full : n_incremental) += 1;
            CHECK(low.last_was_full() == expect_full);
            for (uint32_t s = 0; s < N_OUT; ++s)
                for (uint64_t t : {0ull, 1ull, 5ull, 17ull, 40ull}) {
                    float a = flat_eval(*inc, inc->outputs[s], t, in), b = flat_eval(fresh, fresh.outputs[s], t, in);
                    if (!same_bits(a, b)) {
                        std::fprintf(stderr, "seed %u step %d slot %u t %llu: incremental %a, from scratch %a\n", seed, step, s, (unsigned long long)t, a, b);
                        throw std::runtime_error("mismatch");
                    }
                }
        }
    }
    if (std::getenv("FR_TEST_VERBOSE")) std::fprintf(stderr, "incremental %llu, full %llu, failing graphs %llu\n", (unsigned long long)n_incremental, (unsigned long long)n_full, (unsigned long long)n_errors);
    CHECK(n_incremental > 20 * n_full && n_full > 30 && n_errors > 0 && n_errors < n_incremental);
}

static void bounded_signal_delays_are_staged() {
    // chorus / flanger shapes: Delay(x, amount(t)) with a provably bounded amount
    std::mt19937 rng(21);
    Build b;
    const uint32_t V = 3;
    for (uint32_t v = 0; v < V; ++v) {
        uint32_t mix = voice(b, 32, 110.0f * (v + 1), rng);
        uint32_t lfo = b.op(FR_PRIM_MODULO, N(b.op(FR_PRIM_MULTIPLY, In(0), Cf(0.013f * (v + 1)))), Cf(1.0f));       // [0, 1]
        uint32_t amt = b.op(FR_PRIM_SUM2, Cf(3.0f + v), N(b.op(FR_PRIM_MULTIPLY, Cf(40.0f), N(lfo))));                // [3, 43+]
        uint32_t wet = b.op(FR_PRIM_DELAY, N(mix), N(amt));                                                          // bank ring, signal amount
        uint32_t x = b.op(FR_PRIM_SUM2, N(mix), N(b.op(FR_PRIM_MULTIPLY, Cf(0.5f), N(wet))));
        uint32_t echo = b.op(FR_PRIM_DELAY, N(x), Cf(70.0f));                                                        // + a constant tap on top
        // a second modulated tap whose source is a program cut node (x), amount clamped by Minimum: may be 0 frames
        uint32_t amt2 = b.op(FR_PRIM_MINIMUM, Cf(25.0f), N(b.op(FR_PRIM_MULTIPLY, In(1), In(1))));                    // min(25, noise^2): [.., 25]
        uint32_t wob = b.op(FR_PRIM_DELAY, N(x), N(amt2));
        b.out(N(b.op(FR_PRIM_SUM2, N(b.op(FR_PRIM_SUM2, N(x), N(echo))), N(wob))), v);
    }
    // delayed input and delayed constant with signal amounts; an amount that can be negative / NaN (-> 0 frames)
    uint32_t a3 = b.op(FR_PRIM_MODULO, In(1), Cf(9.0f));                                                              // [0, 9], NaN when the input is inf
    uint32_t din = b.op(FR_PRIM_DELAY, In(1), N(a3));
    uint32_t dct = b.op(FR_PRIM_DELAY, Cf(2.5f), N(b.op(FR_PRIM_MODULO, In(0), Cf(7.0f))));
    uint32_t neg = b.op(FR_PRIM_DELAY, In(0), N(b.op(FR_PRIM_SUM2, Cf(-4.0f), N(a3))));                               // [-4, 5]
    b.out(N(b.op(FR_PRIM_SUM2, N(din), N(b.op(FR_PRIM_SUM2, N(dct), N(neg))))), V);
    // unbounded amount on a computed signal: stays with the pull interpreter
    b.out(N(b.op(FR_PRIM_DELAY, N(b.op(FR_PRIM_SUM2, In(1), Cf(1.0f))), N(b.op(FR_PRIM_MULTIPLY, In(0), Cf(0.5f))))), V + 1);
    auto inspect = [&](const FlatGraph &, const StagedPlan &sp) {
        CHECK(sp.pull_rows.size() == 1 && sp.pull_rows[0] == V + 1);
        CHECK(sp.lmax >= 70 + 43 && sp.lmax <= 70 + 50);
        CHECK(sp.fused_count == 0);        // a signal-delayed read of a program's ring can land inside the current launch
        size_t dyn = 0;
        for (const StageInstr &in : sp.instrs) dyn += (in.op == S_READ_DYN || in.op == S_READ_INPUT_DYN || in.op == S_STEP_DYN) ? 1 : 0;
        CHECK(dyn == 2 * V + 3);   // (the pulled row's Delay is not an instruction)
    };
    check_graph(b, V + 2, 100, 5, true, "bounded signal delays", inspect);
    check_graph(b, V + 2, 17, 9, true, "bounded signal delays, short calls");
    check_compiled_programs(b, V + 2, 64, {0, 64, 128, 192, 5000, 5064}, "bounded signal delays, compiled");
}

static void value_ranges_are_sound() {
    // Planner::range (the interval analysis that licenses staging a signal-amount Delay) against brute-force evaluation
    const int kinds[6] = {FR_PRIM_DELAY, FR_PRIM_SUM2, FR_PRIM_MULTIPLY, FR_PRIM_DIVIDE, FR_PRIM_MODULO, FR_PRIM_MINIMUM};
    const float consts[14] = {0.0f, -0.0f, 1.0f, -1.0f, 0.5f, 2.0f, 3.0f, -3.5f, 7.25f, 1e-30f, 1e30f, -1e30f, 48000.0f, NAN};
    const float specials[12] = {0.0f, -0.0f, 1.0f, -2.5f, 1e30f, -1e30f, INFINITY, -INFINITY, NAN, 3.5f, 1e-40f, 16777216.0f};
    uint64_t bounded_nodes = 0, checked = 0;
    for (unsigned seed = 0; seed < 150; ++seed) {
        std::mt19937 rng(3000 + seed);
        Build b;
        std::vector<uint32_t> avail;
        int n = 6 + (int)(rng() % 30);
        for (int i = 0; i < n; ++i) {
            int k = kinds[rng() % 6];
            auto pick = [&]() -> Operand {
                unsigned r = rng() % 10;
                if (r < 4 || avail.empty()) return (rng() % 3 == 0) ? In(rng() % 2) : Cf(consts[rng() % 14]);
                return N(avail[rng() % avail.size()]);
            };
            Operand a = pick(), bb = pick();
            if (k == FR_PRIM_DELAY) bb = Cf((float)(rng() % 5));
            avail.push_back(b.op(k, a, bb));
        }
        for (uint32_t s = 0; s < 4; ++s) b.out(N(avail[avail.size() - 1 - (rng() % std::min<size_t>(avail.size(), 8))]), s);
        Mirror m;
        b.apply(m);
        FlatGraph fg = lower(m, 4);
        Planner P(fg, nullptr);
        Inputs in(2);
        for (int t = 0; t < 40; ++t) {
            in[0].push_back(t % 7 == 6 ? specials[rng() % 12] : (float)t * 1000.0f);
            in[1].push_back(rng() % 3 ? specials[rng() % 12] : (float)((int)(rng() % 2001) - 1000) * 0.01f);
        }
        for (uint32_t id = 0; id < fg.nodes.size(); ++id) {
            Planner::Range r = P.range(id);
            if (std::isfinite(r.hi) || std::isfinite(r.lo)) ++bounded_nodes;
            for (uint64_t t = 0; t < 40; ++t) {
                float val = flat_eval(fg, id, t, in);
                ++checked;
                bool ok = val != val ? r.nan : ((double)val >= r.lo && (double)val <= r.hi);
                if (!ok) {
                    std::fprintf(stderr, "seed %u node %u (op %u) t %llu: value %a outside [%g, %g] nan=%d\n", seed, id, fg.nodes[id].op,
                                 (unsigned long long)t, val, r.lo, r.hi, (int)r.nan);
                    throw std::runtime_error("unsound range");
                }
            }
        }
    }
    if (std::getenv("FR_TEST_VERBOSE")) std::fprintf(stderr, "%llu values checked, %llu nodes with a finite bound\n", (unsigned long long)checked, (unsigned long long)bounded_nodes);
    CHECK(bounded_nodes > 500);
}

static void oversized_expressions_are_split() {
    // one output whose expression is far beyond a single program (4096 instructions / 48 registers): a 3000-long
    // accumulation chain over a wide tree of non-template terms, with shared sub-expressions and a Delay in the middle
    std::mt19937 rng(31);
    Build b;
    std::vector<uint32_t> terms;
    for (int i = 0; i < 700; ++i) {
        uint32_t x = b.op(FR_PRIM_MULTIPLY, In(i % 2), Cf(0.001f * (float)(i + 1)));
        uint32_t y = b.op(FR_PRIM_MODULO, N(x), Cf(1.0f + (float)(i % 5)));
        uint32_t z = b.op(FR_PRIM_MINIMUM, N(y), N(b.op(FR_PRIM_DIVIDE, N(x), Cf(3.0f + (float)(i % 7)))));
        terms.push_back(b.op(FR_PRIM_SUM2, N(z), N(y)));
    }
    uint32_t tree = sum_tree(b, terms);                       // ~700 * 5 + 699 nodes in one expression
    uint32_t acc = tree;
    for (int i = 0; i < 3000; ++i) {                          // left-deep chain re-using the terms
        acc = b.op(i % 3 ? FR_PRIM_SUM2 : FR_PRIM_MINIMUM, N(acc), N(terms[(size_t)(rng() % terms.size())]));
        if (i == 1500) acc = b.op(FR_PRIM_SUM2, N(acc), N(b.op(FR_PRIM_DELAY, N(acc), Cf(9.0f))));
    }
    b.out(N(acc), 0);
    b.out(N(tree), 1);
    check_graph(b, 2, 20, 3, true, "oversized expression", [](const FlatGraph &, const StagedPlan &sp) {
        CHECK(sp.pull_rows.empty());                          // (before splitting existed both rows fell back to the pull interpreter)
        CHECK(sp.progs.size() > 6);
        for (const StageProg &pg : sp.progs) CHECK(pg.n_instr <= 4096);
    });
}

// ---- generated leaves (leafjit.cpp) on the CPU -----------------------------------------------------------------------
// The leaf function the hipRTC bank kernel is built around, compiled with g++ and compared with the lowered graph's
// own value for every leaf of a voice, on ordinary and hostile inputs: the general body always, the v_fract body
// (FAST) wherever the kernel would select it.  Pins the peepholes (x mod 1 as fract, Minimum(u, -u) as -|u|).
// Voices whose leaves read per-leaf TRACK rows (leafshape.hpp LEAF_TRACK): matched only when the slots are declared tracks,
// the slot numbers travel as per-leaf parameters in leaf order, and the generated leaf -- built here with g++ against a small
// dense matrix -- equals the graph's own evaluation, including slots the call did not supply (beyond `limit`: +0).
static void track_leaves_are_matched_and_generated() {
    const uint32_t first = 10;
    Build b;
    std::vector<uint32_t> leaves;
    for (uint32_t k = 0; k < 32; ++k) {
        uint32_t x = b.op(FR_PRIM_MULTIPLY, In(0), In(first + 2 * k));
        uint32_t ph = b.op(FR_PRIM_MODULO, N(x), Cf(1.0f));
        uint32_t u = b.op(FR_PRIM_SUM2, N(ph), Cf(-0.5f));
        uint32_t m = b.op(FR_PRIM_MINIMUM, N(u), N(b.op(FR_PRIM_MULTIPLY, Cf(-1.0f), N(u))));
        uint32_t q = b.op(FR_PRIM_SUM2, Cf(0.5f), N(m));
        uint32_t y = b.op(FR_PRIM_MULTIPLY, N(b.op(FR_PRIM_MULTIPLY, Cf(-16.0f), N(u))), N(q));
        leaves.push_back(b.op(FR_PRIM_MULTIPLY, In(first + 2 * k + 1), N(y)));
    }
    b.out(N(sum_tree(b, leaves)), 0);
    Mirror m;
    b.apply(m);
    FlatGraph fg = lower(m, 1);
    {   // not declared: every leaf reads other input slots, so the leaves are different shapes -- no voice
        BankMatcher plain(fg, 20, true, false);
        VoiceMatch vm;
        CHECK(!plain.try_voice(fg.outputs[0], vm));
    }
    BankMatcher bm(fg, 20, true, false, first);
    VoiceMatch vm;
    CHECK(bm.try_voice(fg.outputs[0], vm) && vm.jit && vm.tracks && vm.log2_p == 5 && vm.k == 2 && !vm.fast_ok);
    CHECK(vm.max_track_slot == first + 63 && vm.shape.input_slots.size() == 1 && vm.shape.input_slots[0] == 0);
    // parameters: the two slot numbers of every leaf, leaves left to right
    for (uint32_t k = 0; k < 32; ++k) {
        uint32_t a = f32_to_bits(vm.params[k * 2]), c = f32_to_bits(vm.params[k * 2 + 1]);
        if (a > c) std::swap(a, c);
        CHECK(a == first + 2 * k && c == first + 2 * k + 1);
    }
    LeafSource ls = generate_leaf_source(vm.shape, vm.varying, vm.literal_bits, vm.alias);
    CHECK(ls.tracks && ls.k == 2 && ls.track_params.size() == 2);   // both parameters are per-leaf tracks: the leaf takes the rows' values
    std::string base = "/tmp/fr_trackleaf_" + std::to_string((long)getpid());
    FILE *f = std::fopen((base + ".cpp").c_str(), "w");
    CHECK(f != nullptr);
    std::fputs("#include <cmath>\n#define __device__\n#define __forceinline__ inline\n"
               "static inline float __builtin_amdgcn_fractf(float a) { return a - floorf(a); }\n", f);
    std::fputs(ls.text.c_str(), f);
    std::fputs("extern \"C\" float leaf_general(const float *x, const float *trk, unsigned long long stride, unsigned limit, unsigned long long t, const float *p) "
               "{ return leaf<false>(x, trk, stride, limit, t, p[0], p[1]); }\n", f);
    std::fclose(f);
    CHECK(std::system(("g++ -std=c++17 -O1 -ffp-contract=off -w -shared -fPIC -o " + base + ".so " + base + ".cpp").c_str()) == 0);
    void *so = dlopen((base + ".so").c_str(), RTLD_NOW | RTLD_LOCAL);
    CHECK(so != nullptr);
    auto leaf = (float (*)(const float *, const float *, unsigned long long, unsigned, unsigned long long, const float *))dlsym(so, "leaf_general");
    CHECK(leaf != nullptr);
    std::remove((base + ".cpp").c_str());
    std::remove((base + ".so").c_str());
    const uint32_t T = 24, R = first + 64;
    std::mt19937 rng(5);
    std::uniform_real_distribution<float> uni(-0.2f, 1.0f);
    std::vector<float> mat((size_t)R * T);
    for (float &v : mat) v = uni(rng);
    std::vector<uint32_t> leaf_ids;
    std::function<void(uint32_t, int)> walk = [&](uint32_t id, int h) {
        if (h == 0) { leaf_ids.push_back(id); return; }
        walk(fg.nodes[id].a, h - 1);
        walk(fg.nodes[id].b, h - 1);
    };
    walk(fg.outputs[0], 5);
    for (uint32_t limit : {R, first + 20u}) {   // the second: rows from slot first + 20 on were not supplied
        Inputs in(R);
        for (uint32_t s2 = 0; s2 < R; ++s2)
            if (s2 < limit) in[s2].assign(mat.begin() + (size_t)s2 * T, mat.begin() + (size_t)(s2 + 1) * T);
        for (uint32_t t = 0; t < T; ++t)
            for (size_t li = 0; li < leaf_ids.size(); ++li) {
                const float x0 = mat[t];   // slot 0's row
                float vals[2];   // what the kernel's TRACK_LOADS hands the leaf: the row's value at t, +0 for a row not supplied
                for (int q = 0; q < 2; ++q) {
                    const uint32_t slot = f32_to_bits(vm.params[li * 2 + q]);
                    vals[q] = slot < limit ? mat[(size_t)slot * T + t] : 0.0f;
                }
                const float got = leaf(&x0, mat.data(), T, limit, t, vals);
                const float expect = flat_eval(fg, leaf_ids[li], t, in);
                CHECK(f32_to_bits(got) == f32_to_bits(expect) || (got != got && expect != expect));
            }
    }
    dlclose(so);
}

static void generated_leaves_equal_the_graph() {
    const float hostile[] = {0.0f, -0.0f, 1.0f, 0.5f, 0.25f, 3.0f, 1e-30f, 1e-42f, 48000.0f, 16777216.0f, 4294967296.0f, 4294967808.0f,
                             1e30f, 3e38f, -1.0f, -0.5f, -2.75f, -1e30f, INFINITY, -INFINITY, NAN};
    struct Case { const char *name; std::function<uint32_t(Build &, float, float)> leaf; bool expect_fast, expect_abs; };
    std::vector<Case> cases;
    cases.push_back({"N1 partial", [](Build &b, float w, float amp) { return partial(b, w, amp); }, true, true});
    cases.push_back({"triangle x input 1", [](Build &b, float w, float amp) {
        uint32_t ph = b.op(FR_PRIM_MODULO, N(b.op(FR_PRIM_MULTIPLY, In(0), Cf(w))), Cf(1.0f));
        uint32_t u = b.op(FR_PRIM_SUM2, N(ph), Cf(-0.5f));
        uint32_t au = b.op(FR_PRIM_MULTIPLY, Cf(-1.0f), N(b.op(FR_PRIM_MINIMUM, N(u), N(b.op(FR_PRIM_MULTIPLY, Cf(-1.0f), N(u))))));
        uint32_t tri = b.op(FR_PRIM_SUM2, Cf(1.0f), N(b.op(FR_PRIM_MULTIPLY, Cf(-4.0f), N(au))));
        return b.op(FR_PRIM_MULTIPLY, N(b.op(FR_PRIM_MULTIPLY, Cf(amp), N(tri))), In(1)); }, true, true});
    cases.push_back({"phase offset that can go negative", [](Build &b, float w, float amp) {
        uint32_t x = b.op(FR_PRIM_SUM2, N(b.op(FR_PRIM_MULTIPLY, In(0), Cf(w))), Cf(amp - 0.3f));   // offset in (-0.3, 0.7]
        uint32_t ph = b.op(FR_PRIM_MODULO, N(x), Cf(1.0f));
        return b.op(FR_PRIM_MULTIPLY, N(ph), Cf(amp)); }, false, false});
    cases.push_back({"abs of a product (can be -0), other divisors", [](Build &b, float w, float amp) {
        uint32_t x = b.op(FR_PRIM_MULTIPLY, In(0), Cf(w));
        uint32_t m = b.op(FR_PRIM_MINIMUM, N(x), N(b.op(FR_PRIM_MULTIPLY, N(x), Cf(-1.0f))));       // -|x| only if x is never -0: it can be
        uint32_t r = b.op(FR_PRIM_MODULO, N(m), Cf(0.75f));
        return b.op(FR_PRIM_DIVIDE, N(r), Cf(amp + 0.5f)); }, false, false});
    cases.push_back({"sawtooth whose phase hits -0 and the negative integers", [](Build &b, float, float amp) {
        const float wq = 0.25f * std::round(1.0f / amp);   // quarter-integer rates: t * wq is often integral; fmodf(-3, 1) is -0
        uint32_t ph = b.op(FR_PRIM_MODULO, N(b.op(FR_PRIM_MULTIPLY, In(0), Cf(wq))), Cf(1.0f));
        return b.op(FR_PRIM_MULTIPLY, N(ph), Cf(amp)); }, true, false});
    for (const Case &cs : cases) {
        std::mt19937 rng(77);
        Build b;
        std::vector<uint32_t> leaves;
        for (int k = 0; k < 32; ++k) leaves.push_back(cs.leaf(b, 0.0013f * (float)(k + 1), 1.0f / (float)(k + 1)));
        b.out(N(sum_tree(b, leaves)), 0);
        Mirror m;
        b.apply(m);
        FlatGraph fg = lower(m, 1);
        BankMatcher bm(fg, 20, true, false);
        VoiceMatch vm;
        CHECK(bm.try_voice(fg.outputs[0], vm) && vm.jit && vm.log2_p == 5);
        LeafSource ls = generate_leaf_source(vm.shape, vm.varying, vm.literal_bits, vm.alias);
        CHECK(vm.fast_ok == cs.expect_fast);
        CHECK((ls.text.find("__builtin_fabsf") != std::string::npos) == cs.expect_abs);
        // 1 + (-4 * |u|) of the triangle: the product by a power of two is exact, one fused multiply-add replaces both operations
        if (std::string(cs.name).find("triangle") == 0) CHECK(ls.text.find("__builtin_fmaf(__builtin_bit_cast(float, 0xc0800000u)") != std::string::npos);
        if (std::string(cs.name).find("N1") == 0) CHECK(ls.text.find("__builtin_fmaf") == std::string::npos);   // (nothing to fold there)
        // lowered ids of the leaves, in parameter order (left to right)
        std::vector<uint32_t> leaf_ids;
        std::function<void(uint32_t, int)> walk = [&](uint32_t id, int h) {
            if (h == 0) { leaf_ids.push_back(id); return; }
            walk(fg.nodes[id].a, h - 1);
            walk(fg.nodes[id].b, h - 1);
        };
        walk(fg.outputs[0], 5);
        // build the leaf for the CPU
        static int serial = 0;
        std::string base = "/tmp/fr_leafjit_" + std::to_string((long)getpid()) + "_" + std::to_string(serial++);
        FILE *f = std::fopen((base + ".cpp").c_str(), "w");
        CHECK(f != nullptr);
        std::fputs("#include <cmath>\n#define __device__\n#define __forceinline__ inline\n"
                   "static inline float __builtin_amdgcn_fractf(float a) { return a - floorf(a); }\n", f);
        std::fputs(ls.text.c_str(), f);
        for (int fast = 0; fast < 2; ++fast) {
            std::fprintf(f, "extern \"C\" float leaf_%s(const float *x, const float *p) { return leaf<%s>(x", fast ? "fast" : "general", fast ? "true" : "false");
            for (uint32_t i = 0; i < ls.k; ++i) std::fprintf(f, ", p[%u]", i);
            std::fputs("); }\n", f);
        }
        std::fclose(f);
        CHECK(std::system(("g++ -std=c++17 -O1 -ffp-contract=off -w -shared -fPIC -o " + base + ".so " + base + ".cpp").c_str()) == 0);
        void *so = dlopen((base + ".so").c_str(), RTLD_NOW | RTLD_LOCAL);
        CHECK(so != nullptr);
        auto general = (float (*)(const float *, const float *))dlsym(so, "leaf_general");
        auto fastf = (float (*)(const float *, const float *))dlsym(so, "leaf_fast");
        CHECK(general && fastf);
        std::remove((base + ".cpp").c_str());
        std::remove((base + ".so").c_str());
        const size_t nin = vm.shape.input_slots.size();
        uint64_t n_fast = 0, n_general = 0;
        std::uniform_real_distribution<float> uni(0.0f, 100000.0f);
        for (int trial = 0; trial < 600; ++trial) {
            float xin[4] = {0, 0, 0, 0};
            Inputs in(4);
            for (size_t i = 0; i < nin; ++i) {
                float val = trial < 200 ? hostile[(trial * (i + 3) + i) % (sizeof hostile / sizeof hostile[0])]
                                        : (trial % 3 ? std::floor(uni(rng)) : uni(rng) - 20000.0f);
                xin[i] = val;
                uint32_t slot = vm.shape.input_slots[i];
                in[slot].assign(1, val);
            }
            bool in_range = vm.fast_ok && ls.has_mod1;
            for (size_t i = 0; i < nin; ++i)
                if ((ls.fract_inputs >> i) & 1u) in_range = in_range && f32_to_bits(xin[i]) <= 0x4F800000u;
            for (size_t li = 0; li < leaf_ids.size(); ++li) {
                float expect = flat_eval(fg, leaf_ids[li], 0, in);
                const float *prm = vm.params.data() + li * ls.k;
                float got = general(xin, prm);
                ++n_general;
                if (!same_bits(got, expect)) {
                    std::fprintf(stderr, "%s: general body, leaf %zu, x0=%a: %a vs graph %a\n", cs.name, li, xin[0], got, expect);
                    throw std::runtime_error("mismatch");
                }
                if (in_range) {
                    float gf = fastf(xin, prm);
                    ++n_fast;
                    if (!same_bits(gf, expect)) {
                        std::fprintf(stderr, "%s: fract body, leaf %zu, x0=%a: %a vs graph %a\n", cs.name, li, xin[0], gf, expect);
                        throw std::runtime_error("mismatch");
                    }
                }
            }
        }
        dlclose(so);
        CHECK(n_general == 600 * 32 && (cs.expect_fast ? n_fast > 5000 : n_fast == 0));
    }
}

// The parallel from-scratch lowering (rows on several threads, graph.cpp Lowering::Impl::lower_rows_parallel) against the
// one-thread form on graphs built to make threads meet: many rows, each with a part of its own and a part SHARED with other
// rows (an LFO-like sub-graph feeding several voices, whole rows that repeat another row's expression), a row that fails,
// a row through a composite-free cycle.  Same errors; where it lowers, the same values at every sampled time; operands
// precede users in the flat graph; and a Lowering that started in parallel takes incremental edits like any other.
static void parallel_lowering_equals_sequential() {
    std::mt19937 rng(77);
    for (int round = 0; round < 6; ++round) {
        Build b;
        const int rows = 24 + round * 7;
        std::vector<uint32_t> shared;
        for (int i = 0; i < 6; ++i) {   // shared sub-graphs over inputs and constants
            uint32_t x = b.op(FR_PRIM_MULTIPLY, In(i % 2), Cf(0.01f * (i + 1)));
            uint32_t y = b.op(FR_PRIM_MODULO, N(x), Cf(1.0f));
            shared.push_back(b.op(FR_PRIM_SUM2, N(y), i ? N(shared[i - 1]) : Cf(0.25f)));
        }
        std::vector<uint32_t> roots;
        for (int r = 0; r < rows; ++r) {
            uint32_t own = voice(b, 8 + (int)(rng() % 3) * 8, 40.0f + r, rng);
            uint32_t mix = b.op(FR_PRIM_MULTIPLY, N(own), N(shared[rng() % shared.size()]));
            if (r % 5 == 0) mix = b.op(FR_PRIM_DELAY, N(mix), Cf((float)(r + 1)));
            if (r % 7 == 3 && !roots.empty()) mix = b.op(FR_PRIM_SUM2, N(mix), N(roots[rng() % roots.size()]));   // reads another row's root
            roots.push_back(mix);
            b.out(N(mix), (uint32_t)r);
        }
        if (round % 2) b.out(N(roots[3]), (uint32_t)rows);        // a row that repeats another row's expression
        const uint32_t n_slots = (uint32_t)rows + (round % 2 ? 1u : 0u);
        Mirror m;
        b.apply(m);
        setenv("FR_LOWER_PAR_MIN_NODES", "0", 1);
        setenv("FR_LOWER_THREADS", "1", 1);
        FlatGraph seq = lower(m, n_slots);
        setenv("FR_LOWER_THREADS", round % 2 ? "3" : "8", 1);
        FlatGraph par = lower(m, n_slots);
        CHECK(par.outputs.size() == seq.outputs.size());
        for (uint32_t id = 0; id < par.nodes.size(); ++id) {
            const FlatNode &n = par.nodes[id];
            if (n.op != OP_CONST && n.op != OP_INPUT) CHECK(n.a < id && n.b < id);
        }
        Inputs hist(2);
        for (int t = 0; t < 160; ++t) { hist[0].push_back((float)t); hist[1].push_back(0.37f * (float)t - 3.0f); }
        for (uint32_t s2 = 0; s2 < n_slots; ++s2)
            for (uint64_t t : {0ull, 1ull, 17ull, 64ull, 159ull})
                CHECK(same_bits(flat_eval(seq, seq.outputs[s2], t, hist), flat_eval(par, par.outputs[s2], t, hist)));
        // a Lowering that began in parallel, then edits (incremental, sequential), against from-scratch on one thread
        Lowering low;
        const FlatGraph &f0 = low.update(m, n_slots);
        CHECK(low.last_was_full() && low.last_parallel_subtrees() >= 8);
        (void)f0;
        for (int e = 0; e < 20; ++e) {
            const uint32_t victim = roots[rng() % roots.size()];
            fr_edge old_e{}, new_e{};
            bool found = false;
            for (const fr_edge &x : b.edges) if (x.to == victim && x.to_slot == 1) { old_e = x; found = true; }
            if (!found) continue;
            m.del_edge(old_e);
            new_e = fr_edge{1, victim, f32_to_bits(0.5f + 0.01f * e), 1};
            m.add_edge(new_e);
            for (fr_edge &x : b.edges) if (x.to == victim && x.to_slot == 1) x = new_e;
            const FlatGraph &fi = low.update(m, n_slots);
            CHECK(!low.last_was_full());
            setenv("FR_LOWER_THREADS", "1", 1);
            FlatGraph ref = lower(m, n_slots);
            setenv("FR_LOWER_THREADS", "8", 1);
            for (uint32_t s2 = 0; s2 < n_slots; ++s2)
                for (uint64_t t : {0ull, 33ull, 159ull}) CHECK(same_bits(flat_eval(ref, ref.outputs[s2], t, hist), flat_eval(fi, fi.outputs[s2], t, hist)));
        }
        // a note-on: a whole new voice arrives between two updates -- the incremental update lowers its sub-trees on threads
        {
            setenv("FR_LOWER_PAR_MIN_EDIT", "0", 1);
            setenv("FR_LOWER_THREADS", "8", 1);
            Build nb;
            nb.next = b.next + 1000;     // fresh handles
            nb.nodes.clear();
            const uint32_t v2 = voice(nb, 64, 61.7f + round, rng);
            for (auto &n : nb.nodes) { fr_effect e{}; e.kind = n.second; m.add_node(n.first, &e); }
            for (auto &e : nb.edges) m.add_edge(e);
            m.add_edge(fr_edge{v2, 0, 0, n_slots});
            const FlatGraph &fi = low.update(m, n_slots + 1);
            CHECK(!low.last_was_full() && low.last_relowered() >= 64 * 11 && low.last_parallel_subtrees() >= 8);
            setenv("FR_LOWER_THREADS", "1", 1);
            FlatGraph ref = lower(m, n_slots + 1);
            for (uint32_t s2 = 0; s2 <= n_slots; ++s2)
                for (uint64_t t : {0ull, 33ull, 159ull}) CHECK(same_bits(flat_eval(ref, ref.outputs[s2], t, hist), flat_eval(fi, fi.outputs[s2], t, hist)));
            for (uint32_t id = 0; id < fi.nodes.size(); ++id) {
                const FlatNode &n = fi.nodes[id];
                if (n.op != OP_CONST && n.op != OP_INPUT) CHECK(n.a < id && n.b < id);
            }
            unsetenv("FR_LOWER_PAR_MIN_EDIT");
        }
        // errors: an edge from a node that does not exist, in the middle of the rows -> the same error as on one thread
        Mirror bad;
        b.apply(bad);
        bad.add_edge(fr_edge{999999, roots[rows / 2], 0, 0});
        for (const char *threads : {"1", "8"}) {
            setenv("FR_LOWER_THREADS", threads, 1);
            bool threw = false;
            try { (void)lower(bad, n_slots); } catch (const Error &er) { threw = er.code == FR_ERR_NO_SUCH_NODE; }
            CHECK(threw);
        }
        // a cycle between two rows' nodes -> FR_ERR_CYCLE either way (the threads hand it to the sequential pass)
        Mirror cyc;
        b.apply(cyc);
        {
            // roots[1] = own * shared: rewire its slot 1 to read roots[2], and roots[2]'s slot 1 to read roots[1]
            fr_edge e1{}, e2{};
            for (const fr_edge &x : b.edges) { if (x.to == roots[1] && x.to_slot == 1) e1 = x; if (x.to == roots[2] && x.to_slot == 1) e2 = x; }
            cyc.del_edge(e1); cyc.del_edge(e2);
            cyc.add_edge(fr_edge{roots[2], roots[1], 0, 1});
            cyc.add_edge(fr_edge{roots[1], roots[2], 0, 1});
        }
        for (const char *threads : {"1", "8"}) {
            setenv("FR_LOWER_THREADS", threads, 1);
            bool threw = false;
            try { (void)lower(cyc, n_slots); } catch (const Error &er) { threw = er.code == FR_ERR_CYCLE; }
            CHECK(threw);
        }
    }
    unsetenv("FR_LOWER_THREADS");
    unsetenv("FR_LOWER_PAR_MIN_NODES");
}

int main(int argc, char **argv) {
    std::vector<std::pair<const char *, std::function<void()>>> tests = {
        {"parallel_lowering_equals_sequential", parallel_lowering_equals_sequential},
        {"lowering_folds_constants", lowering_folds_constants}, {"lowering_errors", lowering_errors},
        {"random_graphs_lower_correctly", random_graphs_lower_correctly}, {"banks_are_recognised", banks_are_recognised},
        {"effects_chain_is_staged", effects_chain_is_staged}, {"dynamic_delay_goes_to_pull", dynamic_delay_goes_to_pull},
        {"shared_root_without_delays_is_one_launch", shared_root_without_delays_is_one_launch},
        {"composite_instances_are_interned", composite_instances_are_interned},
        {"stage_programs_compile_to_source", stage_programs_compile_to_source},
        {"incremental_lowering_equals_from_scratch", incremental_lowering_equals_from_scratch},
        {"bounded_signal_delays_are_staged", bounded_signal_delays_are_staged},
        {"value_ranges_are_sound", value_ranges_are_sound},
        {"oversized_expressions_are_split", oversized_expressions_are_split},
        {"generated_leaves_equal_the_graph", generated_leaves_equal_the_graph},
        {"track_leaves_are_matched_and_generated", track_leaves_are_matched_and_generated}};
    int failed = 0, ran = 0;
    for (auto &t : tests) {
        if (argc > 1 && std::string(argv[1]) != t.first) continue;
        ++ran;
        try {
            t.second();
            std::printf("test %s ... ok\n", t.first);
        } catch (const std::exception &e) {
            std::printf("test %s ... FAILED: %s\n", t.first, e.what());
            ++failed;
        }
    }
    std::printf("test result: %s. %d passed; %d failed\n", failed ? "FAILED" : "ok", ran - failed, failed);
    return failed ? 1 : 0;
}
