// match.cpp -- recognises oscillator banks in a lowered graph.
//
// The reference has no oscillator primitive: a partial is a sub-graph of the seven
// primitives, a voice is a Sum2 tree over partials.  This matcher finds output slots whose lowered
// expression is exactly
//
//   voice  := complete balanced Sum2 tree with P = 2^k >= 32 leaves
//   leaf   := Multiply(C(amp), y)
//   y      := Multiply(Multiply(C(-16), u), Sum2(C(0.5), Multiply(C(-1), absu)))
//   absu   := Multiply(C(-1), Minimum(u, Multiply(C(-1), u)))          (= |u|)
//   u      := Sum2(Modulo(Multiply(Input(s), C(w)), C(1.0)), C(-0.5))
//
// (operand order of the commutative Sum2/Multiply is free) and hands their
// parameters to the fused bank kernel.  Anything else is left to the generic evaluators, so a
// mismatch costs speed, never correctness.
#include "match.hpp"

#include "range.hpp"

#include <algorithm>
#include <cmath>
#include <functional>
#include <unordered_map>
#include <unordered_set>

namespace fr {
namespace {

struct Leaf {
    float w, A;
    uint32_t slot;
    bool ok;
};

struct Matcher {
    const FlatGraph &g;
    std::unordered_map<uint32_t, Leaf> leaf_memo;
    explicit Matcher(const FlatGraph &fg) : g(fg) {}

    const FlatNode &n(uint32_t id) const { return g.nodes[id]; }

    // id == op(C(c), other) for a commutative op: returns other.
    bool bin_const(uint32_t id, FlatOp op, float c, uint32_t &other) const {
        const FlatNode &x = n(id);
        if (x.op != (uint32_t)op) return false;
        if (g.is_const(x.a, c)) { other = x.b; return true; }
        if (g.is_const(x.b, c)) { other = x.a; return true; }
        return false;
    }
    // id == op(C(any), other): returns the constant and other.
    bool bin_anyconst(uint32_t id, FlatOp op, float &c, uint32_t &other) const {
        const FlatNode &x = n(id);
        if (x.op != (uint32_t)op) return false;
        if (g.is_const(x.a) && !g.is_const(x.b)) { c = g.const_val(x.a); other = x.b; return true; }
        if (g.is_const(x.b) && !g.is_const(x.a)) { c = g.const_val(x.b); other = x.a; return true; }
        return false;
    }

    bool match_q(uint32_t q, uint32_t u) const {
        uint32_t n1 = 0, absu = 0, m = 0, nu_chk = 0;
        if (!bin_const(q, OP_SUM2, 0.5f, n1)) return false;
        if (!bin_const(n1, OP_MUL, -1.0f, absu)) return false;
        if (!bin_const(absu, OP_MUL, -1.0f, m)) return false;
        const FlatNode &mn = n(m);
        if (mn.op != OP_MIN) return false;
        uint32_t nu = (mn.a == u) ? mn.b : (mn.b == u ? mn.a : 0xFFFFFFFFu);
        if (nu == 0xFFFFFFFFu) return false;
        return bin_const(nu, OP_MUL, -1.0f, nu_chk) && nu_chk == u;
    }

    Leaf match_leaf(uint32_t id) {
        auto it = leaf_memo.find(id);
        if (it != leaf_memo.end()) return it->second;
        Leaf L{0, 0, 0, false};
        float amp = 0, w = 0;
        uint32_t y, p, q, u = 0, phase, x, in;
        do {
            if (!bin_anyconst(id, OP_MUL, amp, y)) break;
            const FlatNode &yn = n(y);
            if (yn.op != OP_MUL) break;
            // y = p * q in either operand order
            bool ok = false;
            for (int swap = 0; swap < 2 && !ok; ++swap) {
                p = swap ? yn.b : yn.a;
                q = swap ? yn.a : yn.b;
                ok = bin_const(p, OP_MUL, -16.0f, u) && match_q(q, u);
            }
            if (!ok) break;
            if (!bin_const(u, OP_SUM2, -0.5f, phase)) break;
            const FlatNode &ph = n(phase);
            if (ph.op != OP_MOD || !g.is_const(ph.b, 1.0f)) break;
            x = ph.a;
            if (!bin_anyconst(x, OP_MUL, w, in)) break;
            if (n(in).op != OP_INPUT) break;
            // the kernel multiplies by A4 = -4*amp (and 4*A4 = -16*amp): both must be exact scalings
            float A16 = -16.0f * amp, A4 = -4.0f * amp;
            if (!std::isfinite(A16) || !std::isfinite(w) || A16 / -16.0f != amp || A4 * 4.0f != A16) break;
            L = Leaf{w, A4, n(in).a, true};
        } while (false);
        leaf_memo.emplace(id, L);
        return L;
    }

    // Collects the leaves of a complete Sum2 tree of the given height rooted at id.  Sub-trees of 64 leaves are remembered
    // by node id: after an edit inside a voice (hash-consing gives every node on the path from the edited leaf to the root
    // a new id and leaves all others alone) 63 of a 4096-leaf voice's 64 sub-trees are copied instead of walked
    // (re-planning after one changed amplitude at config C: 0.80 -> 0.62 ms, tools/replan_bench.cpp).
    struct SubTree { bool ok = false, fast_ok = true; uint32_t slot = 0; std::vector<float> params; };
    static constexpr uint32_t SUB_H = 6;
    std::unordered_map<uint32_t, SubTree> sub_memo;
    size_t sub_live = 0;   // entries the last pruning pass found in use (BankMatcher::retain_used)
    bool collect(uint32_t id, uint32_t height, std::vector<float> &params, uint32_t &slot, bool &first, bool &fast_ok) {
        if (height == SUB_H) {
            auto it = sub_memo.find(id);
            if (it == sub_memo.end()) {
                SubTree st;
                bool f = true;
                const FlatNode &x = n(id);
                st.params.reserve(2u << SUB_H);
                st.ok = x.op == OP_SUM2 && collect(x.a, height - 1, st.params, st.slot, f, st.fast_ok) &&
                        collect(x.b, height - 1, st.params, st.slot, f, st.fast_ok);
                if (!st.ok) { st.params.clear(); st.params.shrink_to_fit(); }
                it = sub_memo.emplace(id, std::move(st)).first;
            }
            const SubTree &st = it->second;
            if (!st.ok) return false;
            if (first) { slot = st.slot; first = false; }
            else if (slot != st.slot) return false;
            params.insert(params.end(), st.params.begin(), st.params.end());
            if (!st.fast_ok) fast_ok = false;
            return true;
        }
        if (height == 0) {
            Leaf L = match_leaf(id);
            if (!L.ok) return false;
            if (first) { slot = L.slot; first = false; }
            else if (slot != L.slot) return false;
            params.push_back(L.w);
            params.push_back(L.A);
            if (!(L.w >= 0.0f && L.w <= 4294967296.0f)) fast_ok = false;
            return true;
        }
        const FlatNode &x = n(id);
        if (x.op != OP_SUM2) return false;
        return collect(x.a, height - 1, params, slot, first, fast_ok) &&
               collect(x.b, height - 1, params, slot, first, fast_ok);
    }
};

}  // namespace

struct BankMatcher::Impl : Matcher {
    using Matcher::Matcher;

    // height of the complete sub-tree of matched leaves rooted at n (0 = a leaf, up to GENERAL_MAX_ITEM_LOG2 = 2048
    // leaves), -1 if n is a Sum2 node that is not such a sub-tree, -2 if n is neither a leaf nor a Sum2 (not a voice)
    std::unordered_map<uint32_t, int> cj_memo;
    int complete_height(uint32_t id, uint64_t &budget) {
        auto it = cj_memo.find(id);
        if (it != cj_memo.end()) return it->second;
        if (budget == 0) return -2;
        --budget;
        int r;
        if (match_leaf(id).ok) {
            r = 0;
        } else if (n(id).op != OP_SUM2) {
            r = -2;
        } else {
            int a = complete_height(n(id).a, budget), b = complete_height(n(id).b, budget);
            if (a == -2 || b == -2) r = -2;
            else if (a >= 0 && a == b && a < (int)GENERAL_MAX_ITEM_LOG2) r = a + 1;
            else r = -1;
        }
        cj_memo.emplace(id, r);
        return r;
    }

    void append_leaves(uint32_t id, VoiceMatch &vm, bool &first, bool &ok) {
        Leaf L = match_leaf(id);
        if (L.ok) {
            if (first) { vm.input_slot = L.slot; first = false; }
            else if (vm.input_slot != L.slot) ok = false;
            vm.params.push_back(L.w);
            vm.params.push_back(L.A);
            if (!(L.w >= 0.0f && L.w <= 4294967296.0f)) vm.fast_ok = false;
            ++vm.n_leaves;
            return;
        }
        append_leaves(n(id).a, vm, first, ok);
        append_leaves(n(id).b, vm, first, ok);
    }

    // post-order emission: maximal complete sub-trees become items, every other Sum2 node a merge after its right operand
    bool emit_general(uint32_t root, VoiceMatch &vm) {
        uint64_t budget = 1u << 22;
        if (complete_height(root, budget) == -2) return false;
        struct Item { uint32_t id; int state; };
        std::vector<Item> st{{root, 0}};
        bool first = true, ok = true;
        uint32_t depth = 0, max_depth = 0;
        while (!st.empty() && ok) {
            Item it = st.back();
            st.pop_back();
            if (it.state == 1) {   // both operands emitted: one merge
                if (vm.groups.empty()) return false;
                vm.groups.back() += 1u << 4;
                --depth;
                continue;
            }
            int cj = cj_memo.at(it.id);
            if (cj >= 0) {
                size_t before = vm.params.size();
                append_leaves(it.id, vm, first, ok);
                if (cj < 3) vm.params.resize(before + 16, 0.0f);   // a small item is padded to one group of 8 {w, A4} pairs
                vm.groups.push_back((uint32_t)cj);
                max_depth = std::max(max_depth, ++depth);
                if (vm.groups.size() > (1u << 20)) return false;
                continue;
            }
            st.push_back({it.id, 1});
            st.push_back({n(it.id).b, 0});
            st.push_back({n(it.id).a, 0});
        }
        return ok && max_depth <= 16 && vm.n_leaves >= 16;
    }

    // ---- shape matching (for jit.hpp): leaves of ANY common expression shape -------------------------------
    uint32_t track_from = 0xFFFFFFFFu;   // input slots from here on are tracks (leafshape.hpp LEAF_TRACK)
    std::unordered_map<uint32_t, uint64_t> shape_memo;   // 0 = not a leaf expression (contains a Delay, too deep)
    static uint64_t hmix(uint64_t h, uint64_t v) {
        h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        h *= 0xBF58476D1CE4E5B9ull;
        return h ^ (h >> 29);
    }
    uint64_t shape_hash(uint32_t id, int depth = 0) {
        auto it = shape_memo.find(id);
        if (it != shape_memo.end()) return it->second;
        uint64_t r;
        const FlatNode &x = n(id);
        if (depth > 48 || x.op == OP_DELAY) r = 0;
        else if (x.op == OP_CONST) r = 0xC0757ull;
        else if (x.op == OP_INPUT) r = x.a >= track_from ? 0x7BAC4ull | 1 : hmix(0x17B07ull, x.a) | 1;   // (tracks: any slot, same shape)
        else {
            uint64_t a = shape_hash(x.a, depth + 1), b = shape_hash(x.b, depth + 1);
            if (!a || !b) r = 0;
            else {
                if ((x.op == OP_SUM2 || x.op == OP_MUL) && a > b) std::swap(a, b);
                r = hmix(hmix(0x0F00ull + x.op, a), b) | 1;
            }
        }
        shape_memo.emplace(id, r);
        return r;
    }

    // Canonical serialisation of one leaf as a TREE: every use of a node is emitted again (hash-consing may have
    // merged sub-expressions or constants in one leaf that stay distinct in another, so the DAG is not canonical;
    // the tree is, and the device compiler's CSE puts the sharing back -- exact, since every op is pure).
    // Operands of the commutative ops are ordered by shape hash.  Fills ops/input_slots when `define`, else checks
    // that this leaf serialises to the same ops; appends the leaf's constants (one column per USE) to `consts`.
    bool serialise_leaf(uint32_t root, LeafShape &shape, std::vector<uint32_t> &consts, bool define) {
        std::vector<LeafShape::Op> ops;
        std::vector<uint32_t> inputs = define ? std::vector<uint32_t>{} : shape.input_slots;
        bool ok = true;
        std::function<uint32_t(uint32_t)> emit = [&](uint32_t id) -> uint32_t {
            if (!ok) return 0;
            if (ops.size() >= 96) { ok = false; return 0; }
            const FlatNode &x = n(id);
            if (x.op == OP_CONST) {
                ops.push_back({OP_CONST, (uint32_t)consts.size(), 0});
                consts.push_back(x.a);
                return (uint32_t)ops.size() - 1;
            }
            if (x.op == OP_INPUT && x.a >= track_from) {   // the row's number is a per-leaf parameter like a constant
                ops.push_back({LEAF_TRACK, (uint32_t)consts.size(), 0});
                consts.push_back(x.a);
                return (uint32_t)ops.size() - 1;
            }
            if (x.op == OP_INPUT) {
                uint32_t idx = 0;
                while (idx < inputs.size() && inputs[idx] != x.a) ++idx;
                if (idx == inputs.size()) {
                    if (!define) { ok = false; return 0; }
                    inputs.push_back(x.a);
                }
                ops.push_back({OP_INPUT, idx, 0});
                return (uint32_t)ops.size() - 1;
            }
            uint32_t a = x.a, b = x.b;
            if ((x.op == OP_SUM2 || x.op == OP_MUL) && shape_hash(a) > shape_hash(b)) std::swap(a, b);
            // (two tracks under one commutative op hash alike: lower slot first, as make() ordered them by id or not)
            if ((x.op == OP_SUM2 || x.op == OP_MUL) && n(a).op == OP_INPUT && n(b).op == OP_INPUT && n(a).a >= track_from && n(b).a >= track_from &&
                n(a).a > n(b).a) std::swap(a, b);
            uint32_t la = emit(a), lb = emit(b);
            ops.push_back({x.op, la, lb});
            return (uint32_t)ops.size() - 1;
        };
        emit(root);
        if (!ok) return false;
        if (define) {
            shape.ops = ops;
            shape.input_slots = inputs;
            shape.n_consts = (uint32_t)consts.size();
            return true;
        }
        if (ops.size() != shape.ops.size() || consts.size() != shape.n_consts) return false;
        for (size_t i = 0; i < ops.size(); ++i)
            if (ops[i].op != shape.ops[i].op || ops[i].a != shape.ops[i].a || ops[i].b != shape.ops[i].b) return false;
        return true;
    }

    bool collect_shape(uint32_t id, uint32_t height, uint64_t leaf_hash, std::vector<uint32_t> &leaves) {
        if (height == 0) {
            if (shape_hash(id) != leaf_hash) return false;
            leaves.push_back(id);
            return true;
        }
        const FlatNode &x = n(id);
        return x.op == OP_SUM2 && collect_shape(x.a, height - 1, leaf_hash, leaves) && collect_shape(x.b, height - 1, leaf_hash, leaves);
    }

    bool match_shape_voice(uint32_t root, uint32_t max_log2_p, VoiceMatch &vm) {
        uint32_t h = 0, cur = root;
        while (n(cur).op == OP_SUM2 && shape_hash(n(cur).a) != 0 && shape_hash(n(cur).a) == shape_hash(n(cur).b) && h < max_log2_p) {
            cur = n(cur).a;
            ++h;
        }
        if (h < 5 || h > 13) return false;                 // one workgroup per voice tile: 32..8192 leaves
        if (n(cur).op == OP_CONST || n(cur).op == OP_INPUT) return false;   // a sum of bare inputs/constants is not worth a kernel
        std::vector<uint32_t> leaves;
        leaves.reserve((size_t)1 << h);
        if (!collect_shape(root, h, shape_hash(cur), leaves)) return false;
        std::vector<uint32_t> first, row;
        if (!serialise_leaf(leaves[0], vm.shape, first, true)) return false;
        if (vm.shape.n_consts > 32 || vm.shape.input_slots.size() > 4 || vm.shape.ops.empty()) return false;
        const uint32_t nc = vm.shape.n_consts;
        std::vector<uint32_t> all((size_t)leaves.size() * nc);
        std::copy(first.begin(), first.end(), all.begin());
        vm.varying.assign(nc, false);
        for (size_t i = 1; i < leaves.size(); ++i) {
            row.clear();
            if (!serialise_leaf(leaves[i], vm.shape, row, false)) return false;
            for (uint32_t c = 0; c < nc; ++c) {
                all[i * nc + c] = row[c];
                if (row[c] != first[c]) vm.varying[c] = true;
            }
        }
        // the tree form repeats a shared sub-expression's constants: columns equal in every leaf share one parameter
        vm.alias.resize(nc);
        for (uint32_t c = 0; c < nc; ++c) {
            vm.alias[c] = c;
            if (!vm.varying[c]) continue;
            for (uint32_t e = 0; e < c && vm.alias[c] == c; ++e) {
                if (!vm.varying[e] || vm.alias[e] != e) continue;
                bool same = true;
                for (size_t i = 0; i < leaves.size() && same; ++i) same = all[i * nc + c] == all[i * nc + e];
                if (same) vm.alias[c] = e;
            }
        }
        vm.k = 0;
        for (uint32_t c = 0; c < nc; ++c) vm.k += (vm.varying[c] && vm.alias[c] == c) ? 1 : 0;
        if (vm.k > 8) return false;
        vm.literal_bits = first;
        const uint32_t K = vm.k ? vm.k : 1;
        vm.params.assign((size_t)leaves.size() * K, 0.0f);
        for (size_t i = 0; i < leaves.size(); ++i) {
            uint32_t j = 0;
            for (uint32_t c = 0; c < nc; ++c)
                if (vm.varying[c] && vm.alias[c] == c) vm.params[i * K + j++] = f32_from_bits(all[i * nc + c]);
        }
        vm.jit = true;
        vm.log2_p = h;
        vm.n_leaves = (uint32_t)leaves.size();
        vm.fast_ok = fract_form_is_exact(vm, all, first);
        vm.tracks = false;
        for (const LeafShape::Op &o : vm.shape.ops) vm.tracks = vm.tracks || o.op == LEAF_TRACK;
        if (vm.tracks) vm.fast_ok = false;   // (a track's values are not range-tested per wave as the shared inputs are)
        vm.max_track_slot = 0;
        for (const LeafShape::Op &o : vm.shape.ops)
            if (o.op == LEAF_TRACK)
                for (size_t i = 0; i < leaves.size(); ++i) vm.max_track_slot = std::max(vm.max_track_slot, all[i * nc + o.a]);
        return true;
    }

    // The generated kernel has a second body in which Modulo(x, 1.0) is one v_fract_f32 (x - floor(x)); that equals the
    // graph's fmod-based value exactly when x is finite and >= +0.  Decide here whether that holds for EVERY such
    // Modulo of the leaf whenever the inputs are in [+0, 2^32] (the kernel tests the inputs per wave): interval
    // arithmetic for finiteness, a sign rule for "never negative, never -0".
    bool fract_form_is_exact(const VoiceMatch &vm, const std::vector<uint32_t> &all, const std::vector<uint32_t> &first) const {
        const uint32_t nc = vm.shape.n_consts;
        const size_t n_leaves = nc ? all.size() / nc : 0;
        struct Val { Range r; bool nonneg; };
        std::vector<Val> col(nc);
        for (uint32_t c = 0; c < nc; ++c) {
            if (!vm.varying[c]) {
                float f = f32_from_bits(first[c]);
                col[c] = Val{Range::exactly(f), !(first[c] >> 31) && f == f};
                continue;
            }
            double lo = HUGE_VAL, hi = -HUGE_VAL;
            bool nan = false, nonneg = true;
            for (size_t i = 0; i < n_leaves; ++i) {
                uint32_t bits = all[i * nc + c];
                float f = f32_from_bits(bits);
                if (f != f) { nan = true; continue; }
                lo = std::min(lo, (double)f);
                hi = std::max(hi, (double)f);
                nonneg = nonneg && !(bits >> 31);
            }
            col[c] = Val{Range{lo, hi, nan}, nonneg && !nan};
        }
        std::vector<Val> val(vm.shape.ops.size());
        bool any = false;
        for (size_t i = 0; i < vm.shape.ops.size(); ++i) {
            const LeafShape::Op &o = vm.shape.ops[i];
            if (o.op == OP_CONST) { val[i] = col[o.a]; continue; }
            if (o.op == OP_INPUT) { val[i] = Val{Range{0.0, 4294967296.0, false}, true}; continue; }
            if (o.op == LEAF_TRACK) { val[i] = Val{Range::unbounded(), false}; continue; }
            const Val &a = val[o.a], &b = val[o.b];
            bool nonneg;
            switch (o.op) {
            case OP_SUM2: case OP_MUL: case OP_MIN: nonneg = a.nonneg && b.nonneg; break;
            case OP_DIV: nonneg = a.nonneg && b.nonneg && b.r.lo > 0.0; break;
            default: nonneg = a.nonneg && b.r.lo > 0.0 && !b.r.nan; break;   // OP_MOD: fmod keeps the dividend's sign
            }
            val[i] = Val{Range::combine(o.op, a.r, b.r, g.sparkle), nonneg};
            const LeafShape::Op &d = vm.shape.ops[o.b];
            if (o.op == OP_MOD && d.op == OP_CONST && !vm.varying[d.a] && first[d.a] == 0x3F800000u) {
                any = true;
                if (!(a.nonneg && !a.r.nan && a.r.hi <= 3.0e38)) return false;
            }
        }
        return any;
    }
};

BankMatcher::BankMatcher(const FlatGraph &g, uint32_t max_log2_p, bool allow_jit, bool allow_template, uint32_t track_from)
    : impl_(new Impl(g)), g_(g), max_log2_p_(max_log2_p), allow_jit_(allow_jit), allow_template_(allow_template) { impl_->track_from = track_from; }
BankMatcher::~BankMatcher() { delete impl_; }

void BankMatcher::begin_plan() { used_.clear(); }

void BankMatcher::retain_used() {
    // remembered 64-leaf sub-trees: every edit inside a voice adds one (the edited one's new id); keep those the voices of
    // this plan are made of once the others outnumber them
    if (impl_->sub_memo.size() > 2 * impl_->sub_live + 4096) {   // (the walk below is not free: only when it can pay)
        std::unordered_set<uint32_t> live_sub;
        for (auto &kv : memo_) {
            if (kv.second < 0 || !used_.count(kv.first)) continue;
            const VoiceMatch &vm = found_[(size_t)kv.second];
            if (vm.general || vm.jit || vm.log2_p < Impl::SUB_H) continue;
            std::vector<uint32_t> level{kv.first};
            for (uint32_t h = vm.log2_p; h > Impl::SUB_H; --h) {
                std::vector<uint32_t> next;
                next.reserve(level.size() * 2);
                for (uint32_t id : level) { next.push_back(g_.nodes[id].a); next.push_back(g_.nodes[id].b); }
                level.swap(next);
            }
            live_sub.insert(level.begin(), level.end());
        }
        if (impl_->sub_memo.size() > 2 * live_sub.size() + 1024)
            for (auto it = impl_->sub_memo.begin(); it != impl_->sub_memo.end();)
                it = live_sub.count(it->first) ? std::next(it) : impl_->sub_memo.erase(it);
        impl_->sub_live = live_sub.size();
    }
    size_t live = 0;
    for (auto &kv : memo_) live += used_.count(kv.first) ? 1 : 0;
    if (memo_.size() <= 2 * live + 64) return;
    std::unordered_map<uint32_t, int64_t> memo;
    std::deque<VoiceMatch> found;
    for (auto &kv : memo_) {
        if (!used_.count(kv.first)) continue;
        if (kv.second < 0) { memo.emplace(kv.first, -1); continue; }
        memo.emplace(kv.first, (int64_t)found.size());
        found.push_back(std::move(found_[(size_t)kv.second]));
    }
    memo_.swap(memo);
    found_.swap(found);
}

bool BankMatcher::try_voice(uint32_t root, VoiceMatch &out) {
    const VoiceMatch *vm = match(root);
    if (!vm) return false;
    out = *vm;
    return true;
}

const VoiceMatch *BankMatcher::match(uint32_t root) {
    used_[root] = true;
    auto it = memo_.find(root);
    if (it != memo_.end()) return it->second < 0 ? nullptr : &found_[(size_t)it->second];
    // height = number of Sum2 nodes on the leftmost path
    uint32_t h = 0, cur = root;
    while (g_.nodes[cur].op == OP_SUM2 && h <= max_log2_p_) { cur = g_.nodes[cur].a; ++h; }
    if (allow_template_ && h >= 5 && h <= max_log2_p_) {
        VoiceMatch vm;
        vm.log2_p = h;
        vm.params.reserve((size_t)2 << h);
        bool first = true;
        vm.fast_ok = true;
        if (impl_->collect(root, h, vm.params, vm.input_slot, first, vm.fast_ok)) {
            memo_.emplace(root, (int64_t)found_.size());
            found_.push_back(std::move(vm));
            return &found_.back();
        }
    }
    if (allow_template_ && g_.nodes[root].op == OP_SUM2) {   // not a balanced power-of-two tree: try the general schedule
        VoiceMatch vm;
        vm.general = true;
        if (impl_->emit_general(root, vm)) {
            memo_.emplace(root, (int64_t)found_.size());
            found_.push_back(std::move(vm));
            return &found_.back();
        }
    }
    if (allow_jit_ && g_.nodes[root].op == OP_SUM2) {   // leaves of some other common shape: hipRTC specialisation
        VoiceMatch vm;
        if (impl_->match_shape_voice(root, max_log2_p_, vm)) {
            memo_.emplace(root, (int64_t)found_.size());
            found_.push_back(std::move(vm));
            return &found_.back();
        }
    }
    memo_.emplace(root, -1);
    return nullptr;
}

}  // namespace fr
