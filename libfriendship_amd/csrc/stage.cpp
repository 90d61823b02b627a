// stage.cpp -- plans the staged (materialised) evaluation of a lowered graph.
//
// The reference implements Delay by re-evaluating the whole upstream sub-graph at t - d
// (reference src/render/reference.rs:197-216), so K chained delays cost 2^K upstream evaluations.  On the
// GPU the upstream value is materialised once per frame instead: every node some constant Delay reads back
// in time becomes a *cut node* with a ring buffer in HBM, computed level by level per call; rings persist
// across contiguous calls (that is the delay line) and are recomputed from the input history after a seek
// or a graph edit -- exactly the reference's "recompute from input history with the current graph".
// This is equivalent only because the evaluator is a pure function of (graph, input history, t).
//
// Not staged (left to the pull interpreter, which handles everything): signal-dependent delay amounts,
// delays of 2^31 frames or more, programs that outgrow the register/instruction budget.
#include "stage.hpp"

#include "range.hpp"

#include <algorithm>
#include <cmath>
#include <memory>
#include <string>
#include <unordered_map>
#include <unordered_set>

#include <chrono>
#include <cstdio>
#include <cstdlib>
namespace fr {
namespace {
struct PlanTrace {
    bool on = std::getenv("FR_PLAN_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(const char *what) {
        if (!on) return;
        auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "  plan_stages: -> %s after %.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};
#define FR_PT(x) plan_trace.mark(x)

constexpr uint32_t NO_RING = 0xFFFFFFFFu;
constexpr size_t MAX_PROG_INSTR = 4096;
constexpr uint64_t FUSED_UNBOUNDED = 1ull << 31;   // (delays of 2^31 frames or more are never staged)

struct Planner {
    const FlatGraph &g;
    BankMatcher *matcher;   // null: no fused banks
    std::unordered_map<uint32_t, int8_t> supported_memo;
    std::unordered_map<uint32_t, const VoiceMatch *> bank_of;   // node -> voice (owned by the matcher: 32 KB of parameters each at config C)
    std::unordered_set<uint32_t> cut;                     // program cut nodes (non-bank)
    std::unordered_set<uint32_t> visited;

    Planner(const FlatGraph &fg, BankMatcher *m) : g(fg), matcher(m) {}

    static bool delay_frames_ok(const FlatGraph &g, const FlatNode &n, uint64_t &frames) {
        if (!g.is_const(n.b)) return false;
        float d = g.const_val(n.b);
        if (!(d >= 1.0f) || d >= 2147483648.0f) return false;   // 0 / negative / NaN / >= 2^64 were folded at lowering
        frames = (uint64_t)d;
        return true;
    }

    // ---- value ranges -------------------------------------------------------------------------------------
    // A Delay whose amount is a signal can still be staged when the amount is provably bounded: the source is
    // materialised in a ring as for a constant delay, the look-back is the bound, and the program computes the
    // frame offset per sample.  Bounds come from interval arithmetic over the amount's expression (doubles, widened
    // outward after every step so that f32 rounding cannot escape them).  `nan` = the value may also be NaN (which
    // as a delay amount means 0 frames, reference.rs:206-211).  Infinite bounds make everything downstream unbounded.
    using Range = fr::Range;
    std::unordered_map<uint32_t, Range> range_memo;
    std::unordered_map<uint32_t, uint64_t> dyn_max;   // Delay node with a signal amount -> bound on the delay in frames

    Range range(uint32_t root) {
        std::vector<uint32_t> st{root};
        while (!st.empty()) {
            uint32_t n = st.back();
            if (range_memo.count(n)) { st.pop_back(); continue; }
            const FlatNode &x = g.nodes[n];
            if (x.op == OP_CONST) {
                float c = g.const_val(n);
                range_memo[n] = Range::exactly(c);
                st.pop_back();
                continue;
            }
            if (x.op == OP_INPUT || x.op == OP_FBREF) { range_memo[n] = Range::unbounded(); st.pop_back(); continue; }   // (feedback: no bound known)
            bool need_a = !range_memo.count(x.a);
            bool need_b = x.op != OP_DELAY && !range_memo.count(x.b);
            if (need_a) st.push_back(x.a);
            if (need_b) st.push_back(x.b);
            if (need_a || need_b) continue;
            if (x.op == OP_DELAY) {   // the source at some other time, or 0
                Range s = range_memo[x.a];
                range_memo[n] = Range{std::min(s.lo, 0.0), std::max(s.hi, 0.0), s.nan};
            } else {
                range_memo[n] = Range::combine(x.op, range_memo[x.a], range_memo[x.b], g.sparkle);
            }
            st.pop_back();
        }
        return range_memo[root];
    }
    // Delay node n with a non-constant amount: can it be staged, and with what bound?
    bool dynamic_delay_ok(uint32_t n) {
        if (dyn_max.count(n)) return true;
        if (is_leaf(g.nodes[n].a)) {   // an input or a constant delayed by a signal: read from the input history (kept in
            dyn_max[n] = 0xFFFFFFFFull;   // full), no ring, so no bound is needed
            return true;
        }
        Range r = range(g.nodes[n].b);
        if (!(r.hi < 2147483648.0)) return false;   // unbounded (or NaN bound)
        dyn_max[n] = r.hi <= 0.0 ? 0 : (uint64_t)r.hi;
        return true;
    }

    // iterative post-order "is everything under n stageable"
    bool supported(uint32_t root) {
        std::vector<uint32_t> st{root};
        while (!st.empty()) {
            uint32_t n = st.back();
            if (supported_memo.count(n)) { st.pop_back(); continue; }
            const FlatNode &x = g.nodes[n];
            if (x.op == OP_CONST || x.op == OP_INPUT || x.op == OP_FBREF) { supported_memo[n] = 1; st.pop_back(); continue; }   // (what an OP_FBREF stands for is checked by plan_stages)
            if (is_voice(n)) { supported_memo[n] = 1; st.pop_back(); continue; }   // a bank computes it: nothing below matters
            uint64_t fr_;
            const bool const_delay = x.op == OP_DELAY && delay_frames_ok(g, x, fr_);
            if (x.op == OP_DELAY && !const_delay && (g.is_const(x.b) || !dynamic_delay_ok(n))) { supported_memo[n] = 0; st.pop_back(); continue; }
            bool need_a = !supported_memo.count(x.a);
            bool need_b = !const_delay && !supported_memo.count(x.b);   // a signal amount is computed by the program
            if (need_a) st.push_back(x.a);
            if (need_b) st.push_back(x.b);
            if (need_a || need_b) continue;
            supported_memo[n] = supported_memo[x.a] && (const_delay || supported_memo[x.b]);
            st.pop_back();
        }
        return supported_memo[root] != 0;
    }

    bool is_leaf(uint32_t n) const { return g.nodes[n].op == OP_CONST || g.nodes[n].op == OP_INPUT; }

    bool is_voice(uint32_t n) {
        if (!matcher || g.nodes[n].op != OP_SUM2) return false;
        if (bank_of.count(n)) return true;
        const VoiceMatch *vm = matcher->match(n);
        if (!vm) return false;
        bank_of.emplace(n, vm);
        return true;
    }

    // Splits the program region of cut node m (everything under m down to leaves, banks, Delays and other cut nodes)
    // so that no piece has more than `limit` nodes counted as a tree (shared nodes with multiplicity: an over-estimate
    // of the program's instruction count): whenever a node's count exceeds the limit its bigger operand becomes a cut.
    void split_region(uint32_t m, uint64_t limit) {
        std::unordered_map<uint32_t, uint64_t> size;
        auto atomic = [&](uint32_t n) { return n != m && (is_leaf(n) || bank_of.count(n) || cut.count(n)); };
        std::vector<std::pair<uint32_t, int>> st{{m, 0}};
        while (!st.empty()) {
            auto [n, state] = st.back();
            st.pop_back();
            if (size.count(n)) continue;
            if (atomic(n)) { size[n] = 1; continue; }
            const FlatNode &x = g.nodes[n];
            const bool delay = x.op == OP_DELAY;
            if (delay && !dyn_max.count(n)) { size[n] = 1; continue; }   // a ring / input read
            if (state == 0) {
                st.push_back({n, 1});
                st.push_back({x.b, 0});
                if (!delay) st.push_back({x.a, 0});                      // (a signal-amount Delay computes only its amount here)
                continue;
            }
            for (;;) {
                uint64_t sa = delay ? 0 : size[x.a], sb = size[x.b];
                if (1 + sa + sb <= limit) { size[n] = 1 + sa + sb; break; }
                uint32_t big = (!delay && sa >= sb) ? x.a : x.b;
                if (size[big] <= 1) { size[n] = 1 + sa + sb; break; }    // both operands already atomic
                cut.insert(big);
                size[big] = 1;
            }
        }
    }

    // marks banks and cut nodes under a root
    void explore(uint32_t root) {
        std::vector<uint32_t> st{root};
        while (!st.empty()) {
            uint32_t n = st.back();
            st.pop_back();
            if (is_leaf(n) || !visited.insert(n).second) continue;
            const FlatNode &x = g.nodes[n];
            if (is_voice(n)) continue;
            if (x.op == OP_DELAY) {
                const uint32_t src = g.src_of(x.a);   // (a feedback Delay reads a node lowered after it)
                if (!is_leaf(src)) cut.insert(src);
                st.push_back(src);
                if (dyn_max.count(n)) st.push_back(x.b);   // the amount's own expression
            } else {
                st.push_back(x.a);
                st.push_back(x.b);
            }
        }
    }
};

// Adds a voice to the launch that shares its kind (balanced: partial count; general: all together), time slot
// and destination kind.
std::string voice_key(const VoiceMatch &vm, bool ring, bool ws) {
    std::string key = std::to_string(vm.general ? 63u : vm.log2_p) + "/" + std::to_string(vm.input_slot) + (vm.general ? "g" : "") + (ring ? "r" : "") + (ws ? "w" : "");
    if (vm.jit) {   // same generated source (shape, which columns vary, literal values) and same inputs share a launch
        key += "j" + vm.shape.key();
        for (uint32_t sl : vm.shape.input_slots) key += "s" + std::to_string(sl);
        for (size_t c = 0; c < vm.varying.size(); ++c)
            key += vm.varying[c] ? ("v" + std::to_string(vm.alias[c])) : ("l" + std::to_string(vm.literal_bits[c]));
    }
    return key;
}

// `totals` (optional): parameter floats each launch will hold in the end, so that its array is allocated once -- grown by
// doubling, a megabyte of parameters costs a second megabyte of first-touch page faults on every re-plan.
void add_voice(std::vector<BankLaunch> &banks, std::unordered_map<std::string, size_t> &grp, const VoiceMatch &vm, uint32_t row, bool ring,
               bool ws = false, const std::unordered_map<std::string, size_t> *totals = nullptr) {
    const std::string key = voice_key(vm, ring, ws);
    auto gi = grp.find(key);
    if (gi == grp.end()) {
        gi = grp.emplace(key, banks.size()).first;
        BankLaunch bl;
        bl.log2_p = vm.log2_p; bl.input_slot = vm.input_slot; bl.to_ring = ring; bl.to_ws = ws; bl.general = vm.general;
        if (vm.general) { bl.group_off.push_back(0); bl.group_off.push_back(0); }
        if (vm.jit) { bl.jit = true; bl.shape = vm.shape; bl.varying = vm.varying; bl.literal_bits = vm.literal_bits; bl.alias = vm.alias; bl.k = vm.k; bl.tracks = vm.tracks; }
        if (totals) {
            auto ti = totals->find(key);
            if (ti != totals->end()) bl.params.reserve(ti->second);
        }
        banks.push_back(std::move(bl));
    }
    BankLaunch &bl = banks[gi->second];
    bl.rows.push_back(row);
    bl.params.insert(bl.params.end(), vm.params.begin(), vm.params.end());
    bl.fast_ok = bl.fast_ok && vm.fast_ok;
    bl.max_track_slot = std::max(bl.max_track_slot, vm.max_track_slot);
    if (vm.general) {
        bl.groups.insert(bl.groups.end(), vm.groups.begin(), vm.groups.end());
        bl.group_off.push_back((uint32_t)bl.groups.size());
        bl.group_off.push_back((uint32_t)(bl.params.size() / 16));
        bl.max_leaves = std::max(bl.max_leaves, vm.n_leaves);
    }
}

struct ProgBuild {
    std::vector<StageInstr> instrs;   // `buf` holds the cut NODE id until rings are assigned
    uint32_t result_reg = 0;
    uint32_t n_loads = 0;             // leading instructions without register operands (hoisted loads)
    std::vector<std::pair<uint32_t, uint64_t>> reads;   // (cut node, delay -- for a signal amount its upper bound)
    std::unordered_set<uint32_t> dynamic_reads;         // cut nodes read with a signal amount (possibly 0 frames back)
};

// Emits the program computing cut node `m`.  Returns false if it does not fit the budgets.
// fuse == false: other cut nodes are read from their rings (level-by-level mode).
// fuse == true:  program cut nodes used at the SAME frame are computed inline (and stored to their ring with S_STORE
//                when `stored` says they have one), so that only delayed reads touch rings; `min_delay` receives the
//                smallest delay with which a program cut node's ring is read (the fused mode is valid for calls of at
//                most that many frames, in steady state).
bool build_program(const FlatGraph &g, const Planner &P, uint32_t m, std::unordered_map<uint32_t, uint32_t> &dense_input,
                   std::vector<uint32_t> &input_slots, ProgBuild &out, bool fuse = false,
                   const std::unordered_set<uint32_t> *stored = nullptr, uint64_t *min_delay = nullptr, uint64_t *gcd_delay = nullptr) {
    auto is_boundary = [&](uint32_t n) {
        if (n == m) return false;
        if (P.bank_of.count(n)) return true;
        return !fuse && P.cut.count(n) != 0;
    };
    // post-order over the expression DAG inside this stage
    std::vector<uint32_t> order;
    std::unordered_map<uint32_t, uint32_t> uses;
    {
        std::unordered_set<uint32_t> seen;
        std::vector<std::pair<uint32_t, int>> st{{m, 0}};
        while (!st.empty()) {
            auto [n, state] = st.back();
            st.pop_back();
            if (state == 1) { order.push_back(n); continue; }
            if (!seen.insert(n).second) continue;
            st.push_back({n, 1});
            const FlatNode &x = g.nodes[n];
            if (is_boundary(n) || x.op == OP_CONST || x.op == OP_INPUT) continue;   // leaves of the program
            if (x.op == OP_DELAY) {   // the source is read from a ring / input row; a signal amount is computed here, at t
                if (P.dyn_max.count(n)) { ++uses[x.b]; st.push_back({x.b, 0}); }
                continue;
            }
            ++uses[x.a];
            ++uses[x.b];
            st.push_back({x.b, 0});
            st.push_back({x.a, 0});
        }
    }
    // Loads first (they have no register operands), so the kernel can overlap their latencies; only when that does
    // not inflate register pressure beyond the hoisting window.
    {
        auto is_load = [&](uint32_t n) {
            const FlatNode &x = g.nodes[n];
            return is_boundary(n) || x.op == OP_CONST || x.op == OP_INPUT || (x.op == OP_DELAY && !P.dyn_max.count(n));
        };
        size_t n_loads = 0;
        for (uint32_t n : order) n_loads += is_load(n) ? 1 : 0;
        if (n_loads <= STAGE_MAX_HOISTED) {
            std::stable_partition(order.begin(), order.end(), is_load);
            out.n_loads = (uint32_t)n_loads;
        }
    }
    std::vector<uint8_t> free_regs;
    for (int r = STAGE_REGS - 1; r >= 0; --r) free_regs.push_back((uint8_t)r);
    std::unordered_map<uint32_t, uint8_t> reg_of;
    auto dense = [&](uint32_t slot) {
        auto it = dense_input.emplace(slot, (uint32_t)input_slots.size());
        if (it.second) input_slots.push_back(slot);
        return it.first->second;
    };
    for (uint32_t n : order) {
        if (out.instrs.size() >= MAX_PROG_INSTR || free_regs.empty()) return false;
        const FlatNode &x = g.nodes[n];
        StageInstr in{};
        if (is_boundary(n)) {
            in.op = S_READ; in.buf = n; in.d_lo = 0;
            out.reads.push_back({n, 0});
        } else if (x.op == OP_CONST) {
            in.op = S_CONST; in.imm = x.a;
        } else if (x.op == OP_INPUT) {
            in.op = S_INPUT; in.imm = dense(x.a);
        } else if (x.op == OP_DELAY && P.dyn_max.count(n)) {
            const uint64_t bound = P.dyn_max.at(n);
            const FlatNode &src = g.nodes[x.a];
            in.a = reg_of.at(x.b);
            in.d_lo = (uint32_t)bound;   // (informational: the proven bound)
            if (src.op == OP_CONST) { in.op = S_STEP_DYN; in.imm = src.a; }
            else if (src.op == OP_INPUT) { in.op = S_READ_INPUT_DYN; in.imm = dense(src.a); }
            else {
                in.op = S_READ_DYN; in.buf = x.a;
                out.reads.push_back({x.a, bound});
                out.dynamic_reads.insert(x.a);
                // the offset can be anything in [0, bound]: a ring filled by this very launch is not safe to read
                if (min_delay && !P.bank_of.count(x.a)) *min_delay = 0;
            }
            if (--uses[x.b] == 0) free_regs.push_back(reg_of.at(x.b));
        } else if (x.op == OP_DELAY) {
            uint64_t d;
            Planner::delay_frames_ok(g, x, d);
            const uint32_t sa = g.src_of(x.a);
            const FlatNode &src = g.nodes[sa];
            in.d_lo = (uint32_t)d;
            if (src.op == OP_CONST) { in.op = S_STEP; in.imm = src.a; }
            else if (src.op == OP_INPUT) { in.op = S_READ_INPUT; in.imm = dense(src.a); }
            else {
                in.op = S_READ; in.buf = sa; out.reads.push_back({sa, d});
                if (min_delay && !P.bank_of.count(sa)) *min_delay = std::min(*min_delay, d);
                if (gcd_delay && !P.bank_of.count(sa)) { uint64_t a_ = *gcd_delay, b_ = d; while (b_) { const uint64_t r_ = a_ % b_; a_ = b_; b_ = r_; } *gcd_delay = a_; }
            }
        } else {
            switch (x.op) {
            case OP_SUM2: in.op = S_SUM2; break;
            case OP_MUL: in.op = S_MUL; break;
            case OP_DIV: in.op = S_DIV; break;
            case OP_MOD: in.op = S_MOD; break;
            default: in.op = S_MIN; break;
            }
            in.a = reg_of.at(x.a);
            in.b = reg_of.at(x.b);
            // operands die after their last use inside this program
            if (--uses[x.a] == 0) free_regs.push_back(reg_of.at(x.a));
            if (--uses[x.b] == 0) free_regs.push_back(reg_of.at(x.b));   // x.a == x.b: counted twice, freed once
        }
        in.dst = free_regs.back();
        free_regs.pop_back();
        reg_of[n] = in.dst;
        out.instrs.push_back(in);
        if (fuse && n != m && P.cut.count(n) && stored && stored->count(n)) {   // an inlined cut node still feeds its ring
            StageInstr stx{};
            stx.op = S_STORE; stx.a = in.dst; stx.buf = n;
            out.instrs.push_back(stx);
        }
    }
    out.result_reg = reg_of.at(m);
    // the hoisted prefix is whatever leading run of loads the final instruction list has (an S_STORE of an inlined
    // Delay-rooted cut node may sit among them and ends the prefix)
    uint32_t lead = 0;
    while (lead < out.instrs.size() && lead < out.n_loads && out.instrs[lead].op <= S_STEP) ++lead;
    out.n_loads = lead;
    return true;
}

// Tracks (fr_set_track_inputs) are never stored: only a voice leaf of the call that supplies them can read one.
void check_tracks(const FlatGraph &g, const StagedPlan &sp, uint32_t track_from) {
    if (track_from != 0xFFFFFFFFu && g.has_input && g.max_input_slot >= track_from) {
        // tracks are never stored: only a voice leaf of the call that supplies them can read one
        auto refuse = [](const char *who) { throw Error(FR_ERR_UNSUPPORTED, std::string("a track input (fr_set_track_inputs) is read by ") + who +
                                                                                ": only the leaves of shape-matched voices can read tracks"); };
        for (uint32_t sl : sp.input_slots) if (sl >= track_from) refuse("a stage program");
        for (const BankLaunch &bl : sp.banks) {
            if (!bl.jit && bl.input_slot >= track_from) refuse("a template voice as its time input");
            if (bl.tracks && (bl.to_ring || bl.to_ws)) refuse("a voice that feeds a delay line or is split across GPUs");
        }
        std::unordered_set<uint32_t> seen;
        std::vector<uint32_t> st;
        for (uint32_t row : sp.pull_rows) st.push_back(g.outputs[row]);
        while (!st.empty()) {
            const uint32_t n = st.back();
            st.pop_back();
            if (!seen.insert(n).second) continue;
            const FlatNode &x = g.nodes[n];
            if (x.op == OP_INPUT && x.a >= track_from) refuse("a row left to the pull interpreter");
            if (x.op == OP_CONST || x.op == OP_INPUT || x.op == OP_FBREF) continue;
            st.push_back(x.a);
            st.push_back(x.b);
        }
    }
}

}  // namespace

StagedPlan plan_stages(const FlatGraph &g, bool allow_banks, bool allow_programs, uint32_t max_log2_p, bool allow_jit, bool allow_template,
                       BankMatcher *reuse, const ShardSpec *shard, uint32_t track_from) {
    StagedPlan sp;
    PlanTrace plan_trace;
    const uint32_t n_rows = (uint32_t)g.outputs.size();
    // Sharding (friendship_render.h fr_shard).  FR_SHARD_VOICES: plan only what this rank's rows need.  FR_SHARD_PARTIALS:
    // analyse the whole graph (every rank must arrive at the same list of split voices and the same look-back), then keep
    // the programs this rank's rows need, its own sub-tree of every split voice, and the unsplit voices it needs.
    const bool sharded = shard && shard->world > 1 && shard->mode != FR_SHARD_NONE;
    const bool partials = sharded && shard->mode == FR_SHARD_PARTIALS;
    const uint32_t my_rank = sharded ? shard->rank : 0, world = sharded ? shard->world : 1;
    uint32_t my_lo = 0, my_hi = n_rows;
    if (sharded) shard_row_range(my_rank, world, n_rows, my_lo, my_hi);
    auto mine = [&](uint32_t row) { return row >= my_lo && row < my_hi; };
    std::unique_ptr<BankMatcher> own;
    BankMatcher *matcher = nullptr;
    if (allow_banks) {
        if (reuse) matcher = reuse;
        else { own.reset(new BankMatcher(g, max_log2_p, allow_jit, allow_template, track_from)); matcher = own.get(); }
        matcher->begin_plan();
    }
    struct Retain { BankMatcher *m; ~Retain() { if (m) m->retain_used(); } } retain{reuse ? matcher : nullptr};
    Planner P(g, matcher);

    FR_PT("A_rows");
    std::vector<uint32_t> staged_rows;
    for (uint32_t row = 0; row < n_rows; ++row) {
        if (sharded && !partials && !mine(row)) continue;
        uint32_t root = g.outputs[row];
        bool root_is_bank = P.is_voice(root);
        if (root_is_bank || (allow_programs && P.supported(root))) staged_rows.push_back(row);
        else if (mine(row)) sp.pull_rows.push_back(row);   // (another rank's pull row: that rank evaluates all of it)
    }
    // Feedback (graph.hpp OP_FBREF): what the cut Delays read must be stageable too, nothing the pull interpreter evaluates
    // may reach one (its stack is sized by the graph's depth; a loop has none), and the plan has only its fused form --
    // below -- run with threads striding by the gcd of the loop delays, each thread reading what it stored itself.
    // (only the loops the rendered rows reach: after edits the graph may hold OP_FBREF leaves of loops that are gone)
    std::vector<uint32_t> live_targets;
    if (!g.fb_target.empty()) {
        std::unordered_set<uint32_t> seen;
        std::vector<uint32_t> st;
        for (uint32_t row = 0; row < n_rows; ++row)
            if (!sharded || partials || mine(row)) st.push_back(g.outputs[row]);
        while (!st.empty()) {
            const uint32_t n = st.back();
            st.pop_back();
            if (!seen.insert(n).second) continue;
            const FlatNode &x = g.nodes[n];
            if (x.op == OP_CONST || x.op == OP_INPUT) continue;
            if (x.op == OP_FBREF) { live_targets.push_back(g.fb_target[x.a]); st.push_back(g.fb_target[x.a]); continue; }
            st.push_back(x.a);
            st.push_back(x.b);
        }
    }
    const bool feedback = !live_targets.empty();
    sp.feedback_loops = (uint32_t)live_targets.size();
    if (feedback) {
        if (partials) throw Error(FR_ERR_UNSUPPORTED, "feedback through Delay is not available under partial-block sharding");
        for (uint32_t t : live_targets)
            if (!allow_programs || !P.supported(t))
                throw Error(FR_ERR_UNSUPPORTED, "a feedback loop needs the staged evaluator and contains something it cannot take "
                                                "(a Delay by an unbounded signal or by 2^31 frames or more, or FR_MODE_PULL)");
        std::unordered_set<uint32_t> seen;
        std::vector<uint32_t> st;
        for (uint32_t row : sp.pull_rows) st.push_back(g.outputs[row]);
        while (!st.empty()) {
            const uint32_t n = st.back();
            st.pop_back();
            if (!seen.insert(n).second) continue;
            const FlatNode &x = g.nodes[n];
            if (x.op == OP_FBREF)
                throw Error(FR_ERR_UNSUPPORTED, "an output row that the pull interpreter must evaluate (signal-dependent Delay without a bound) "
                                                "reaches a feedback loop");
            if (x.op == OP_CONST || x.op == OP_INPUT) continue;
            st.push_back(x.a);
            st.push_back(x.b);
        }
    }
    FR_PT("B_explore");
    for (uint32_t row : staged_rows) P.explore(g.outputs[row]);
    for (auto &kv : P.bank_of) P.cut.erase(kv.first);   // a Delay's source that is a voice is computed by the bank kernel

    // output rows per root
    std::unordered_map<uint32_t, std::vector<uint32_t>> rows_of;
    for (uint32_t row : staged_rows) rows_of[g.outputs[row]].push_back(row);
    for (auto &kv : rows_of)
        if (!P.bank_of.count(kv.first)) P.cut.insert(kv.first);   // a root that is a leaf gets a trivial program too

    FR_PT("C_build");
    // programs (cut node ids ascending == topological).  An expression too big for one program (instructions or
    // registers) is split: nodes inside it become extra cut nodes -- materialised in a ring, read back at the same
    // frame -- until every piece fits; the levels below order them.
    std::vector<uint32_t> cuts;
    std::unordered_map<uint32_t, uint32_t> dense_input;
    std::unordered_map<uint32_t, ProgBuild> built;
    bool ok = allow_programs || P.cut.empty();
    for (uint64_t limit = 1024; ok;) {
        cuts.assign(P.cut.begin(), P.cut.end());
        std::sort(cuts.begin(), cuts.end());
        uint32_t failed = 0;
        bool all = true;
        for (uint32_t m : cuts) {
            if (built.count(m)) continue;
            ProgBuild pb;
            if (!build_program(g, P, m, dense_input, sp.input_slots, pb)) { all = false; failed = m; break; }
            built.emplace(m, std::move(pb));
        }
        if (all) break;
        size_t before = P.cut.size();
        P.split_region(failed, limit);
        if (P.cut.size() == before) limit /= 2;    // nothing left to cut at this size: try smaller pieces
        else built.clear();                        // programs that inlined a node that is now a cut read its ring instead
        if (limit < 8) ok = false;
    }
    if (!ok && feedback) throw Error(FR_ERR_UNSUPPORTED, "a feedback loop's expression exceeds the stage programs' budget");
    if (!ok) {
        // budget exceeded somewhere: keep only rows whose root is itself a bank (direct launches), pull the rest
        StagedPlan fb;
        fb.pull_rows = sp.pull_rows;
        std::unordered_map<std::string, size_t> grp;
        for (uint32_t row : staged_rows) {
            if (!mine(row)) continue;
            auto it = P.bank_of.find(g.outputs[row]);
            if (it == P.bank_of.end()) { fb.pull_rows.push_back(row); continue; }
            const VoiceMatch &vm = *it->second;
            add_voice(fb.banks, grp, vm, row, false);
        }
        std::sort(fb.pull_rows.begin(), fb.pull_rows.end());
        check_tracks(g, fb, track_from);
        return fb;
    }

    FR_PT("D_rings");
    // who needs a ring: anything read by a program, or feeding more than one output row
    std::unordered_set<uint32_t> needs_ring;
    for (auto &kv : built)
        for (auto &rd : kv.second.reads) needs_ring.insert(rd.first);
    for (auto &kv : rows_of)
        if (kv.second.size() > 1) needs_ring.insert(kv.first);
    // program cut nodes another program uses at the SAME frame (the fused form computes them inline)
    std::unordered_set<uint32_t> same_frame_used;
    for (auto &kv : built)
        for (auto &rd : kv.second.reads)
            if ((rd.second == 0 || kv.second.dynamic_reads.count(rd.first)) && !P.bank_of.count(rd.first)) same_frame_used.insert(rd.first);
    if (feedback)   // every row root lives in a ring: rows the fused programs do not write themselves (a node another program
        for (auto &kv : rows_of)   // computes inline, a second row of the same node) are copied from it after the launch (post_first),
            if (!P.bank_of.count(kv.first)) needs_ring.insert(kv.first);   // and programs merged into one hand their results over through it

    // look-back and levels, consumers before producers (descending node id)
    std::unordered_map<uint32_t, uint64_t> L;
    std::unordered_map<uint32_t, uint32_t> level;
    for (auto it = cuts.rbegin(); it != cuts.rend(); ++it) {
        uint64_t lm = L[*it];
        for (auto &rd : built[*it].reads) L[rd.first] = std::max(L[rd.first], lm + rd.second);
    }
    for (uint32_t m : cuts) {   // ascending: producers first
        uint32_t lv = 1;
        for (auto &rd : built[m].reads) lv = std::max(lv, (P.bank_of.count(rd.first) ? 0u : level[rd.first]) + 1);
        level[m] = lv;
    }

    // input look-back (see StagedPlan::input_lookback)
    for (uint32_t m : cuts) {
        const uint64_t lm = L[m];
        for (const StageInstr &in : built[m].instrs) {
            if (in.op == S_INPUT) sp.input_lookback = std::max(sp.input_lookback, lm);
            else if (in.op == S_READ_INPUT) sp.input_lookback = std::max(sp.input_lookback, lm + in.d_lo);
            else if (in.op == S_READ_INPUT_DYN) {
                if (in.d_lo == 0xFFFFFFFFu) sp.input_lookback_unbounded = true;
                else sp.input_lookback = std::max(sp.input_lookback, lm + in.d_lo);
            }
        }
    }
    for (auto &kv : P.bank_of) sp.input_lookback = std::max(sp.input_lookback, L[kv.first]);
    if (!sp.pull_rows.empty()) sp.input_lookback_unbounded = true;

    // rings
    std::unordered_map<uint32_t, uint32_t> ring_of;
    std::vector<uint32_t> ring_nodes(needs_ring.begin(), needs_ring.end());
    std::sort(ring_nodes.begin(), ring_nodes.end());
    for (uint32_t n : ring_nodes) {
        ring_of[n] = sp.n_rings++;
        sp.lmax = std::max(sp.lmax, L[n]);
    }
    if (feedback) {
        // The look-back of a loop has no bound; the rings are never rebuilt from a window.  They are brought up to date by
        // replaying the frames from 0 in order (engine.cpp execute()), so a ring only ever serves its readers' own delays.
        sp.lmax = 0;
        for (auto &kv : built)
            for (auto &rd : kv.second.reads) sp.lmax = std::max(sp.lmax, rd.second);
        sp.input_lookback_unbounded = true;
    }

    // Which ranks need each cut node / voice (partial-block sharding): everything reachable from the roots of a rank's
    // rows without entering a voice.  One bit per rank.
    std::unordered_map<uint32_t, uint64_t> need;
    if (partials) {
        for (uint32_t o = 0; o < world; ++o) {
            uint32_t lo, hi;
            shard_row_range(o, world, n_rows, lo, hi);
            std::unordered_set<uint32_t> seen;
            std::vector<uint32_t> st;
            for (uint32_t row : staged_rows)
                if (row >= lo && row < hi) st.push_back(g.outputs[row]);
            while (!st.empty()) {
                uint32_t n = st.back();
                st.pop_back();
                if (!seen.insert(n).second) continue;
                if (P.bank_of.count(n)) { need[n] |= 1ull << o; continue; }
                if (P.cut.count(n)) need[n] |= 1ull << o;   // (a root that is itself an input or a constant is a cut node too)
                if (P.is_leaf(n)) continue;
                const FlatNode &x = g.nodes[n];
                st.push_back(x.a);
                if (x.op != OP_DELAY || P.dyn_max.count(n)) st.push_back(x.b);
            }
        }
    }
    auto needed = [&](uint32_t n) {
        if (!partials) return true;
        auto it = need.find(n);
        return it != need.end() && ((it->second >> my_rank) & 1ull) != 0;
    };
    // output rows this rank writes, per root
    std::unordered_map<uint32_t, std::vector<uint32_t>> my_rows_of;
    for (auto &kv : rows_of)
        for (uint32_t row : kv.second)
            if (mine(row)) my_rows_of[kv.first].push_back(row);

    FR_PT("E_banks");
    // bank launches: voices that go straight to one output row, and voices that fill rings
    {
        std::unordered_map<std::string, size_t> grp;
        std::vector<uint32_t> bank_nodes;
        for (auto &kv : P.bank_of) bank_nodes.push_back(kv.first);
        std::sort(bank_nodes.begin(), bank_nodes.end());
        uint32_t kbits = 0;
        while ((1u << kbits) < world) ++kbits;
        struct Split { uint32_t key, node; SplitVoice sv; VoiceMatch sub; };
        std::vector<Split> splits;
        struct Add { const VoiceMatch *vm; uint32_t dst; bool ring; };
        std::vector<Add> adds;
        for (uint32_t n : bank_nodes) {
            const VoiceMatch &vm = *P.bank_of[n];
            bool ring = needs_ring.count(n) != 0;
            auto ro = rows_of.find(n);
            if (!ring && ro == rows_of.end()) continue;   // unreachable
            const uint64_t mask = partials ? need[n] : 0;
            if (partials && mask != 0 && (mask & (mask - 1)) == 0) {
                // needed by exactly one rank: cut it at the top log2(world) levels of its Sum2 tree; this rank renders
                // sub-tree number `my_rank` of it.  A balanced template voice is cut by slicing its parameters (leaves are
                // collected left to right); any other voice by matching the sub-roots themselves.
                uint32_t owner = 0;
                while (!((mask >> owner) & 1ull)) ++owner;
                VoiceMatch sub;
                bool ok_split = false;
                if (!vm.general && !vm.jit && vm.log2_p >= kbits + 5) {
                    sub.log2_p = vm.log2_p - kbits;
                    sub.input_slot = vm.input_slot;
                    sub.fast_ok = vm.fast_ok;
                    const size_t per = (size_t)2 << sub.log2_p;
                    sub.params.assign(vm.params.begin() + (ptrdiff_t)(per * my_rank), vm.params.begin() + (ptrdiff_t)(per * (my_rank + 1)));
                    ok_split = true;
                } else if (vm.general || vm.jit) {
                    std::vector<uint32_t> subs{n};
                    ok_split = true;
                    for (uint32_t lv = 0; lv < kbits && ok_split; ++lv) {
                        std::vector<uint32_t> nx;
                        for (uint32_t x : subs) {
                            if (g.nodes[x].op != OP_SUM2) { ok_split = false; break; }
                            nx.push_back(g.nodes[x].a);
                            nx.push_back(g.nodes[x].b);
                        }
                        subs.swap(nx);
                    }
                    for (size_t i = 0; ok_split && i < subs.size(); ++i) ok_split = matcher->match(subs[i]) != nullptr;
                    if (ok_split) sub = *matcher->match(subs[my_rank]);
                }
                if (ok_split) {
                    uint32_t rev = 0;   // exchange order: by the owner's bits, lowest first, so every step's ranges are contiguous
                    for (uint32_t b = 0; b < kbits; ++b) rev |= ((owner >> b) & 1u) << (kbits - 1 - b);
                    Split sp1;
                    sp1.key = rev;
                    sp1.node = n;
                    sp1.sv.owner = owner;
                    sp1.sv.to_ring = ring;
                    sp1.sv.dst = ring ? ring_of[n] : ro->second[0];
                    sp1.sub = std::move(sub);
                    splits.push_back(std::move(sp1));
                    continue;
                }
            }
            if (!needed(n)) continue;
            if (!ring) {
                auto mr = my_rows_of.find(n);
                if (mr == my_rows_of.end()) continue;
                adds.push_back(Add{&vm, mr->second[0], false});
            } else {
                adds.push_back(Add{&vm, ring_of[n], true});
            }
        }
        std::unordered_map<std::string, size_t> totals;
        for (const Add &a : adds) totals[voice_key(*a.vm, a.ring, false)] += a.vm->params.size();
        for (const Add &a : adds) add_voice(sp.banks, grp, *a.vm, a.dst, a.ring, false, &totals);
        std::stable_sort(splits.begin(), splits.end(), [](const Split &a, const Split &b) { return a.key != b.key ? a.key < b.key : a.node < b.node; });
        for (size_t i = 0; i < splits.size(); ++i) {
            sp.split.push_back(splits[i].sv);
            add_voice(sp.banks, grp, splits[i].sub, (uint32_t)i, false, true);
        }
    }

    FR_PT("F_progs");
    // programs, ordered by level
    struct Pending { uint32_t level; uint32_t node; ProgBuild *pb; uint32_t dst_ring; int32_t out_row; };
    std::vector<Pending> pend;
    std::vector<ProgBuild> extra;   // trivial copies ring -> additional output rows
    extra.reserve(n_rows);
    uint32_t max_level = 0;
    for (uint32_t m : cuts) {
        if (!needed(m)) continue;
        auto ro = my_rows_of.find(m);
        int32_t row0 = ro != my_rows_of.end() ? (int32_t)ro->second[0] : -1;
        pend.push_back({level[m], m, &built[m], needs_ring.count(m) ? ring_of[m] : NO_RING, row0});
        max_level = std::max(max_level, level[m]);
    }
    for (auto &kv : my_rows_of) {
        bool bank = P.bank_of.count(kv.first) != 0;
        size_t first_extra = (bank && !needs_ring.count(kv.first)) ? kv.second.size() : (bank ? 0 : 1);
        for (size_t i = first_extra; i < kv.second.size(); ++i) {
            ProgBuild pb;
            StageInstr in{};
            in.op = S_READ; in.buf = kv.first; in.dst = 0;
            pb.instrs.push_back(in);
            pb.result_reg = 0;
            extra.push_back(std::move(pb));
            uint32_t lv = (bank ? 0u : level[kv.first]) + 1;
            pend.push_back({lv, kv.first, &extra.back(), NO_RING, (int32_t)kv.second[i]});
            max_level = std::max(max_level, lv);
        }
    }
    std::stable_sort(pend.begin(), pend.end(), [](const Pending &a, const Pending &b) { return a.level < b.level; });
    sp.level_first.assign((size_t)max_level + 2, 0);
    for (const Pending &pd : pend) {
        StageProg pg{};
        pg.first_instr = (uint32_t)sp.instrs.size();
        pg.n_instr = (uint32_t)pd.pb->instrs.size();
        pg.result_reg = pd.pb->result_reg;
        pg.dst_ring = pd.dst_ring;
        pg.out_row = pd.out_row;
        pg.n_loads = pd.pb->n_loads;
        for (StageInstr in : pd.pb->instrs) {
            if (in.op == S_READ || in.op == S_READ_DYN) in.buf = ring_of.at(in.buf);
            sp.instrs.push_back(in);
        }
        sp.progs.push_back(pg);
        ++sp.level_first[pd.level + 1];
    }
    for (size_t l = 1; l < sp.level_first.size(); ++l) sp.level_first[l] += sp.level_first[l - 1];

    FR_PT("G_fused");
    // Fused steady-state form: one launch.  Sinks = cut nodes no other program uses at the same frame; everything they
    // use at the same frame is computed inline.  Worth it only if the level form needs more than one launch.
    if (max_level >= 2 || feedback) {
        uint64_t min_delay = ~0ull, gcd_delay = 0;
        bool fits = true;
        std::vector<StageInstr> finstrs;
        std::vector<StageProg> fprogs;
        auto emit = [&](const ProgBuild &pb, uint32_t dst_ring, int32_t out_row) {
            StageProg pg{};
            pg.first_instr = (uint32_t)(sp.instrs.size() + finstrs.size());
            pg.n_instr = (uint32_t)pb.instrs.size();
            pg.result_reg = pb.result_reg;
            pg.dst_ring = dst_ring;
            pg.out_row = out_row;
            pg.n_loads = pb.n_loads;
            for (StageInstr in : pb.instrs) {
                if (in.op == S_READ || in.op == S_STORE || in.op == S_READ_DYN) in.buf = ring_of.at(in.buf);
                finstrs.push_back(in);
            }
            fprogs.push_back(pg);
        };
        for (uint32_t m : cuts) {
            if (!needed(m)) continue;
            auto ro = my_rows_of.find(m);
            bool sink = !same_frame_used.count(m);
            // a non-sink with output rows still needs those rows written: compute it as its own (fused) program too
            if (!sink && (feedback || ro == my_rows_of.end())) continue;   // (feedback: its rows are copied from its ring afterwards)
            ProgBuild pb;
            if (!build_program(g, P, m, dense_input, sp.input_slots, pb, true, &needs_ring, &min_delay, &gcd_delay)) { fits = false; break; }
            emit(pb, needs_ring.count(m) ? ring_of[m] : NO_RING, ro != my_rows_of.end() ? (int32_t)ro->second[0] : -1);
            if (ro != my_rows_of.end() && !feedback)
                for (size_t i = 1; i < ro->second.size(); ++i) emit(pb, NO_RING, (int32_t)ro->second[i]);   // extra rows: recompute
        }
        for (auto &kv : my_rows_of) {   // bank roots that live in a ring: copy programs, as in the level form
            if (!P.bank_of.count(kv.first) || !needs_ring.count(kv.first)) continue;
            for (uint32_t row : kv.second) {
                ProgBuild pb;
                StageInstr in{};
                in.op = S_READ; in.buf = kv.first; in.dst = 0;
                pb.instrs.push_back(in);
                emit(pb, NO_RING, (int32_t)row);
            }
        }
        if (feedback) {
            if (!fits) throw Error(FR_ERR_UNSUPPORTED, "a feedback loop's expression exceeds the stage programs' budget");
            if (min_delay == 0) throw Error(FR_ERR_UNSUPPORTED, "a feedback plan reads a program's ring through a signal-dependent Delay");
            // Order between the fused programs: one that reads (delayed) a ring it does not store itself must run after every
            // program that stores it -- a launch per level, each over the whole window.  Programs that depend on each other in a
            // circle (a loop through Delays whose sources are not used at the same frame: x = in + Delay(y, 2), y = g *
            // Delay(x, 3)) are MERGED into one program -- their instruction lists one after the other, every result but the last
            // stored to its ring by an S_STORE -- so that one thread computes all of them frame by frame; what they read of
            // each other is delayed by at least one frame, so the order inside a frame does not matter.
            std::unordered_set<uint32_t> bank_rings;
            for (auto &kv : P.bank_of) { auto it = ring_of.find(kv.first); if (it != ring_of.end()) bank_rings.insert(it->second); }
            const size_t np = fprogs.size();
            const size_t fbase = sp.instrs.size();
            std::vector<std::unordered_set<uint32_t>> stores(np), foreign(np);
            for (size_t i = 0; i < np; ++i) {
                const StageProg &pg = fprogs[i];
                if (pg.dst_ring != NO_RING) stores[i].insert(pg.dst_ring);
                const size_t f0 = pg.first_instr - fbase;
                for (uint32_t k = 0; k < pg.n_instr; ++k) if (finstrs[f0 + k].op == S_STORE) stores[i].insert(finstrs[f0 + k].buf);
                for (uint32_t k = 0; k < pg.n_instr; ++k) {
                    const StageInstr &in = finstrs[f0 + k];
                    if (in.op == S_READ && !bank_rings.count(in.buf) && !stores[i].count(in.buf)) foreign[i].insert(in.buf);
                }
            }
            std::vector<std::vector<uint32_t>> deps(np);
            for (size_t i = 0; i < np; ++i)
                for (size_t q = 0; q < np; ++q) {
                    if (q == i) continue;
                    for (uint32_t ring : foreign[i]) if (stores[q].count(ring)) { deps[i].push_back((uint32_t)q); break; }
                }
            // strongly connected components (Tarjan, iterative); components come out producers first
            std::vector<int> index(np, -1), low(np, 0), comp(np, -1);
            std::vector<bool> on_stack(np, false);
            std::vector<uint32_t> tstack;
            std::vector<std::vector<uint32_t>> comps;
            int counter = 0;
            for (size_t root = 0; root < np; ++root) {
                if (index[root] >= 0) continue;
                std::vector<std::pair<uint32_t, size_t>> work{{(uint32_t)root, 0}};
                while (!work.empty()) {
                    const uint32_t v = work.back().first;
                    size_t &ei = work.back().second;
                    if (ei == 0) { index[v] = low[v] = counter++; tstack.push_back(v); on_stack[v] = true; }
                    if (ei < deps[v].size()) {
                        const uint32_t w = deps[v][ei++];
                        if (index[w] < 0) work.push_back({w, 0});
                        else if (on_stack[w]) low[v] = std::min(low[v], index[w]);
                        continue;
                    }
                    if (low[v] == index[v]) {
                        comps.emplace_back();
                        for (;;) {
                            const uint32_t w = tstack.back();
                            tstack.pop_back();
                            on_stack[w] = false;
                            comp[w] = (int)comps.size() - 1;
                            comps.back().push_back(w);
                            if (w == v) break;
                        }
                        std::sort(comps.back().begin(), comps.back().end());
                    }
                    work.pop_back();
                    if (!work.empty()) low[work.back().first] = std::min(low[work.back().first], low[v]);
                }
            }
            // one program per component; level = longest chain of components below it
            struct Copy { uint32_t ring; int32_t row; };
            std::vector<Copy> copies;
            std::vector<StageInstr> minstrs;
            std::vector<StageProg> mprogs;
            std::vector<uint32_t> clevel(comps.size(), 0);
            bool carry_only = true;
            for (size_t c = 0; c < comps.size(); ++c) {   // (producers first: every dependency's level is final)
                for (uint32_t i : comps[c])
                    for (uint32_t q : deps[i])
                        if (comp[q] != (int)c) clevel[c] = std::max(clevel[c], clevel[(size_t)comp[q]] + 1);
                StageProg mp{};
                mp.first_instr = (uint32_t)(fbase + minstrs.size());
                for (size_t k = 0; k < comps[c].size(); ++k) {
                    const StageProg &pg = fprogs[comps[c][k]];
                    const size_t f0 = pg.first_instr - fbase;
                    minstrs.insert(minstrs.end(), finstrs.begin() + (ptrdiff_t)f0, finstrs.begin() + (ptrdiff_t)(f0 + pg.n_instr));
                    if (k == 0) mp.n_loads = pg.n_loads;
                    if (k + 1 < comps[c].size()) {
                        if (pg.dst_ring == NO_RING) throw Error(FR_ERR_UNSUPPORTED, "internal: a merged feedback program without a ring");
                        StageInstr stx{};
                        stx.op = S_STORE; stx.a = (uint8_t)pg.result_reg; stx.buf = pg.dst_ring;
                        minstrs.push_back(stx);
                        if (pg.out_row >= 0) copies.push_back(Copy{pg.dst_ring, pg.out_row});
                    } else {
                        mp.result_reg = pg.result_reg;
                        mp.dst_ring = pg.dst_ring;
                        mp.out_row = pg.out_row;
                    }
                }
                // Carry (kernels.hpp STAGE_CARRY): the program's own result becomes an explicit store too, every ring it stores
                // gets a slot (the first 8), and a read of one of them exactly `stride` frames back -- what the same thread stored
                // one iteration ago -- is marked to come from the slot.  A one-sample loop then runs without a memory round trip
                // per frame.
                if (mp.dst_ring != NO_RING) {
                    StageInstr stx{};
                    stx.op = S_STORE; stx.a = (uint8_t)mp.result_reg; stx.buf = mp.dst_ring;
                    minstrs.push_back(stx);
                    mp.dst_ring = NO_RING;
                }
                mp.n_instr = (uint32_t)(fbase + minstrs.size()) - mp.first_instr;
                if (mp.n_instr > MAX_PROG_INSTR) throw Error(FR_ERR_UNSUPPORTED, "a feedback loop's expression exceeds the stage programs' budget");
                {
                    const uint64_t stride = min_delay == ~0ull ? 0 : gcd_delay;
                    std::unordered_map<uint32_t, uint32_t> slot_of;   // ring -> carry slot + 1 (0: it has none)
                    StageInstr *ins = minstrs.data() + (mp.first_instr - fbase);
                    for (uint32_t k = 0; k < mp.n_instr; ++k)
                        if (ins[k].op == S_STORE) {
                            auto it = slot_of.find(ins[k].buf);
                            if (it == slot_of.end()) it = slot_of.emplace(ins[k].buf, slot_of.size() < STAGE_CARRY ? (uint32_t)slot_of.size() + 1u : 0u).first;
                            ins[k].imm = it->second;
                        }
                    for (uint32_t k = 0; k < mp.n_instr; ++k) {
                        if (ins[k].op != S_READ) continue;
                        auto it = slot_of.find(ins[k].buf);
                        if (it == slot_of.end()) continue;                      // a bank's ring, or one an earlier level stored
                        if (stride != 0 && ins[k].d_lo == stride && it->second != 0u) ins[k].imm = it->second;
                        else { ins[k].imm = 0xFFu; carry_only = false; }        // further back than one iteration: through memory, in order
                    }
                }
                mprogs.push_back(mp);
            }
            std::vector<size_t> order(mprogs.size());
            for (size_t i = 0; i < order.size(); ++i) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return clevel[x] < clevel[y]; });
            sp.fused_first = (uint32_t)sp.progs.size();
            sp.fused_count = (uint32_t)mprogs.size();
            sp.fused_max_frames = 0;
            sp.fused_stride = min_delay == ~0ull ? 0 : gcd_delay;
            sp.feedback = true;
            sp.fused_carry_only = carry_only;
            sp.instrs.insert(sp.instrs.end(), minstrs.begin(), minstrs.end());
            sp.fused_level_first.clear();
            for (size_t k = 0; k < order.size(); ++k) {
                if (k == 0 || clevel[order[k]] != clevel[order[k - 1]]) sp.fused_level_first.push_back((uint32_t)k);
                sp.progs.push_back(mprogs[order[k]]);
            }
            sp.fused_level_first.push_back((uint32_t)order.size());
            // rows of nodes computed inline by another program (or stored mid-way by a merged one): ring -> row copies, one more
            // launch over the call's frames
            for (uint32_t m : cuts) {
                if (!needed(m)) continue;
                auto ro = my_rows_of.find(m);
                if (ro == my_rows_of.end()) continue;
                for (size_t i = same_frame_used.count(m) ? 0 : 1; i < ro->second.size(); ++i) copies.push_back(Copy{ring_of.at(m), (int32_t)ro->second[i]});
            }
            sp.post_first = (uint32_t)sp.progs.size();
            for (const Copy &cp : copies) {
                StageProg pg{};
                pg.first_instr = (uint32_t)sp.instrs.size();
                pg.n_instr = 1;
                pg.result_reg = 0;
                pg.dst_ring = NO_RING;
                pg.out_row = cp.row;
                pg.n_loads = 0;
                StageInstr in{};
                in.op = S_READ; in.buf = cp.ring; in.dst = 0;
                sp.instrs.push_back(in);
                sp.progs.push_back(pg);
                ++sp.post_count;
            }
        } else if (fits && !fprogs.empty() && min_delay >= 64) {
            sp.fused_first = (uint32_t)sp.progs.size();
            sp.fused_count = (uint32_t)fprogs.size();
            // no delayed read of a program ring at all (min_delay untouched): a steady call of any length is one launch
            sp.fused_max_frames = std::min<uint64_t>(min_delay, FUSED_UNBOUNDED);
            // every delayed read of a program's ring reaches back a multiple of `fused_stride` frames: a thread that walks the
            // frames wi, wi + stride, wi + 2 stride ... of a longer call only ever reads what IT stored (or an earlier call did)
            sp.fused_stride = min_delay == ~0ull ? 0 : gcd_delay;
            // ... provided the ring is one the reading program stores itself (its own chain: x1 = x0 + g * Delay(x0)); a ring that
            // ANOTHER fused program stores would be another thread's, with no order between the two inside one launch
            std::unordered_set<uint32_t> bank_rings;   // (complete before any program runs: reading them is always safe)
            for (auto &kv : P.bank_of) { auto it = ring_of.find(kv.first); if (it != ring_of.end()) bank_rings.insert(it->second); }
            for (const StageProg &pg : fprogs) {
                std::unordered_set<uint32_t> mine;
                if (pg.dst_ring != NO_RING) mine.insert(pg.dst_ring);
                const size_t f0 = pg.first_instr - sp.instrs.size();
                for (uint32_t i = 0; i < pg.n_instr; ++i) if (finstrs[f0 + i].op == S_STORE) mine.insert(finstrs[f0 + i].buf);
                for (uint32_t i = 0; i < pg.n_instr && sp.fused_stride; ++i) {
                    StageInstr &in = finstrs[f0 + i];
                    if (in.op == S_READ && !bank_rings.count(in.buf) && !mine.count(in.buf)) sp.fused_stride = 0;
                    // (a ring the program stores itself: a strided thread must read it AFTER its own earlier stores -- compiled
                    //  programs fetch their other loads one iteration ahead, kernels.hpp STAGE_CARRY)
                    if (in.op == S_READ && mine.count(in.buf)) in.imm = 0xFFu;
                }
            }
            sp.instrs.insert(sp.instrs.end(), finstrs.begin(), finstrs.end());
            sp.progs.insert(sp.progs.end(), fprogs.begin(), fprogs.end());
        }
    }
    FR_PT("H_end");
    std::sort(sp.pull_rows.begin(), sp.pull_rows.end());
    check_tracks(g, sp, track_from);
    return sp;
}

}  // namespace fr
