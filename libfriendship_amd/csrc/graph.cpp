// graph.cpp -- mirror maintenance (GraphWatcher side) and lowering to a flat primitive DAG.
#include "graph.hpp"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <sys/resource.h>
#include <unistd.h>

namespace {
// The lowering's helper threads, parked between updates: starting 31 threads costs about a millisecond, which is a third of
// a note-on's whole budget; waking parked ones costs tens of microseconds.  One pool per process, grown on demand, never
// shrunk; run() is serialised (two engines lowering at once take turns -- each update is a few milliseconds).
class LowerPool {
public:
    static LowerPool &get() { static LowerPool p; return p; }
    // fn(t) for t in [0, n): t == 0 on the caller, the rest on pool threads; returns when all have
    void run(unsigned n, const std::function<void(unsigned)> &fn) {
        if (n <= 1) { fn(0); return; }
        std::lock_guard<std::mutex> serial(run_mu_);
        {
            std::unique_lock<std::mutex> lk(mu_);
            if (pid_ != getpid()) {   // after a fork() the parked threads exist in the parent only: forget them (never joined, never freed)
                new std::vector<std::thread>(std::move(workers_));
                workers_.clear();
                pid_ = getpid();
            }
            while (workers_.size() < n - 1) {
                const unsigned id = (unsigned)workers_.size() + 1;
                workers_.emplace_back([this, id] { worker(id); });
            }
            fn_ = &fn;
            n_ = n;
            pending_ = n - 1;
            ++epoch_;
        }
        cv_.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return pending_ == 0; });
        fn_ = nullptr;
    }
    ~LowerPool() {
        if (pid_ != getpid()) { new std::vector<std::thread>(std::move(workers_)); return; }
        { std::lock_guard<std::mutex> lk(mu_); quit_ = true; }
        cv_.notify_all();
        for (std::thread &t : workers_) t.join();
    }
private:
    void worker(unsigned id) {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_.wait(lk, [&] { return quit_ || epoch_ != seen; });
            if (quit_) return;
            seen = epoch_;
            if (id >= n_) continue;   // this round uses fewer threads
            const std::function<void(unsigned)> *fn = fn_;
            lk.unlock();
            (*fn)(id);
            lk.lock();
            if (--pending_ == 0) done_.notify_one();
        }
    }
    std::mutex run_mu_, mu_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> workers_;
    const std::function<void(unsigned)> *fn_ = nullptr;
    unsigned n_ = 0, pending_ = 0;
    uint64_t epoch_ = 0;
    bool quit_ = false;
    pid_t pid_ = getpid();
};
}   // namespace

namespace fr {

float f32_from_bits(uint32_t b) {
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}
uint32_t f32_to_bits(float f) {
    uint32_t b;
    std::memcpy(&b, &f, 4);
    return b;
}

static inline uint64_t mix(uint64_t h, uint64_t v) {
    h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h *= 0xBF58476D1CE4E5B9ull;
    return h ^ (h >> 31);
}

// ---- Mirror ------------------------------------------------------------------------------------

static void check_effect(const fr_effect *e) {
    if (!e) throw Error(FR_ERR_INVALID_ARG, "null effect");
    if (e->kind < 0 || e->kind > FR_EFFECT_GRAPH) throw Error(FR_ERR_INVALID_ARG, "bad effect kind");
    if (e->kind == FR_EFFECT_GRAPH &&
        ((e->n_nodes && (!e->node_handles || !e->node_effects)) || (e->n_edges && !e->edges)))
        throw Error(FR_ERR_INVALID_ARG, "composite effect with null arrays");
}

static void set_slot(std::vector<EdgeRef> &v, uint32_t slot, const EdgeRef &r) {
    if (v.size() <= slot) v.resize((size_t)slot + 1);
    v[slot] = r;
}

static void set_slot(EdgeSlots &v, uint32_t slot, const EdgeRef &r) { v.set(slot, r); }

bool SubGraph::equals(const SubGraph &o) const {
    auto same_refs = [](const auto &a, const auto &b) {
        if (a.size() != b.size()) return false;
        for (size_t i = 0; i < a.size(); ++i)
            if (a[i].present != b[i].present ||
                (a[i].present && (a[i].from != b[i].from || a[i].from_slot != b[i].from_slot)))
                return false;
        return true;
    };
    if (hash != o.hash || handles != o.handles || !same_refs(outputs, o.outputs)) return false;
    for (size_t i = 0; i < nodes.size(); ++i) {
        if (nodes[i].kind != o.nodes[i].kind || nodes[i].sub != o.nodes[i].sub ||
            !same_refs(nodes[i].inbound, o.nodes[i].inbound))
            return false;
    }
    return true;
}

// Builds (or finds) the immutable definition of a composite effect: the renderer-side copy that
// RefRenderer::make_node produces (reference.rs:101-111), made once per distinct definition.
std::shared_ptr<const SubGraph> Mirror::intern(const fr_effect *e, int depth) {
    if (depth > 256) throw Error(FR_ERR_INVALID_ARG, "effect nesting too deep");
    auto g = std::make_shared<SubGraph>();
    uint64_t h = 0x5EEDull;
    g->handles.reserve(e->n_nodes);
    g->nodes.reserve(e->n_nodes);
    for (uint32_t i = 0; i < e->n_nodes; ++i) {
        uint32_t hnd = e->node_handles[i];
        const fr_effect *ce = e->node_effects[i];
        if (hnd == 0) throw Error(FR_ERR_INVALID_ARG, "node handle 0 is reserved for graph I/O");
        check_effect(ce);
        MNode n;
        n.kind = ce->kind;
        if (ce->kind == FR_EFFECT_GRAPH) n.sub = intern(ce, depth + 1);
        auto ins = g->index.emplace(hnd, (uint32_t)g->nodes.size());
        if (!ins.second) {  // HashMap::insert replaces an existing entry (reference.rs:105)
            g->nodes[ins.first->second] = std::move(n);
        } else {
            g->handles.push_back(hnd);
            g->nodes.push_back(std::move(n));
        }
        h = mix(h, hnd);
        h = mix(h, (uint64_t)ce->kind);
        if (ce->kind == FR_EFFECT_GRAPH) h = mix(h, g->nodes[g->index[hnd]].sub->hash);
    }
    for (uint32_t i = 0; i < e->n_edges; ++i) {
        const fr_edge &ed = e->edges[i];
        EdgeRef r{ed.from, ed.from_slot, true};
        if (ed.to == 0) {
            set_slot(g->outputs, ed.to_slot, r);
        } else {
            auto it = g->index.find(ed.to);
            if (it == g->index.end())  // `.unwrap()` of reference.rs:145
                throw Error(FR_ERR_NO_SUCH_NODE, "composite effect: edge into unknown node " + std::to_string(ed.to));
            set_slot(g->nodes[it->second].inbound, ed.to_slot, r);
        }
        h = mix(h, ((uint64_t)ed.from << 32) | ed.to);
        h = mix(h, ((uint64_t)ed.from_slot << 32) | ed.to_slot);
    }
    g->hash = h;
    auto range = interned_.equal_range(h);
    for (auto it = range.first; it != range.second;) {
        if (auto sp = it->second.lock()) {
            if (sp->equals(*g)) return sp;
            ++it;
        } else {
            it = interned_.erase(it);
        }
    }
    interned_.emplace(h, g);
    return g;
}

void Mirror::add_node(uint32_t handle, const fr_effect *e) {  // reference.rs:117-120
    if (handle == 0) throw Error(FR_ERR_INVALID_ARG, "node handle 0 is reserved for graph I/O");
    check_effect(e);
    MNode n;
    n.kind = e->kind;
    if (e->kind == FR_EFFECT_GRAPH) n.sub = intern(e, 0);
    // readers of an F32Constant node are not tracked one by one (its value rides on the edges): replacing it is a
    // from-scratch lowering
    if (const MNode *old = nodes.find(handle))
        if (old->kind == FR_PRIM_F32CONSTANT) journal_overflow = true;
    nodes.set(handle, std::move(n));
    note(nodes.find(handle)->pos | JOURNAL_NODE);
    ++version;
}

void Mirror::del_node(uint32_t handle) {  // reference.rs:121-123
    if (const MNode *old = nodes.find(handle)) {
        if (old->kind == FR_PRIM_F32CONSTANT) journal_overflow = true;
        note(old->pos | JOURNAL_NODE);
    }
    nodes.erase(handle);
    ++version;
}

void Mirror::add_edge(const fr_edge &e) {  // reference.rs:124-126,141-153
    // The reference resizes its slot vector to to_slot + 1 whatever the value (reference.rs:148-150); a conforming
    // host never sends a slot beyond an effect's arity (routegraph.rs:165-208 checks first).  Refuse absurd ones
    // instead of allocating gigabytes on garbage.
    if (e.to_slot >= MAX_TO_SLOT) throw Error(FR_ERR_UNSUPPORTED, "add_edge: to_slot " + std::to_string(e.to_slot) + " beyond the supported " + std::to_string(MAX_TO_SLOT));
    EdgeRef r{e.from, e.from_slot, true};
    if (e.to == 0) {
        set_slot(outputs, e.to_slot, r);
    } else {
        MNode *n = nodes.find(e.to);
        if (!n) throw Error(FR_ERR_NO_SUCH_NODE, "add_edge: destination node " + std::to_string(e.to) + " unknown");
        set_slot(n->inbound, e.to_slot, r);
        note(n->pos);
    }
    ++version;
}

void Mirror::del_edge(const fr_edge &e) {  // reference.rs:127-136
    if (e.to == 0) {
        if (e.to_slot < outputs.size()) outputs[e.to_slot] = EdgeRef{};
    } else {
        MNode *n = nodes.find(e.to);
        if (!n) throw Error(FR_ERR_NO_SUCH_NODE, "Attempt to delete edge, but it was never created!");
        n->inbound.clear(e.to_slot);
        note(n->pos);
    }
    ++version;
}

// ---- FlatGraph -----------------------------------------------------------------------------------

// Same arithmetic as the device code and as reference.rs:221-262; this file is compiled with
// -ffp-contract=off so every operation rounds once.
float host_binop(FlatOp op, float a, float b, bool sparkle) {
    switch (op) {
    case OP_SUM2: return a + b;
    case OP_MUL: return a * b;
    case OP_DIV: return a / b;
    case OP_MOD: {
        float rem = std::fmod(a, b);
        return rem < 0.0f ? rem + b : rem;
    }
    case OP_MIN:
        if (sparkle && a != a) return a;             // select(fcmp ult a, b, a, b), sparkle.rs:495-496: a NaN on the left wins
        return (a < b || b != b) ? a : b;            // Rust >= 1.20 f32::min
    default: return 0.0f;
    }
}

bool FlatGraph::is_const(uint32_t id, float v) const {
    return nodes[id].op == OP_CONST && nodes[id].a == f32_to_bits(v);
}
float FlatGraph::const_val(uint32_t id) const { return f32_from_bits(nodes[id].a); }

uint32_t FlatGraph::push(FlatOp op, uint32_t a, uint32_t b, uint32_t depth) {
    if (nodes.size() >= 0x3FFFFFFFu) throw Error(FR_ERR_UNSUPPORTED, "lowered graph exceeds 2^30 nodes");
    nodes.push_back(FlatNode{op, a, b, depth});
    if (depth > max_depth) max_depth = depth;
    return (uint32_t)nodes.size() - 1;
}

uint32_t FlatGraph::konst(uint32_t bits) {
    uint64_t &slot = cse_[OP_CONST].get(bits);
    if (!slot) slot = (uint64_t)push(OP_CONST, bits, 0, 0) + 1;
    return (uint32_t)(slot - 1);
}

uint32_t FlatGraph::input(uint32_t slot) {
    uint64_t &e = cse_[OP_INPUT].get(slot);
    if (e) return (uint32_t)(e - 1);
    uint32_t id = push(OP_INPUT, slot, 0, 0);
    cse_[OP_INPUT].get(slot) = (uint64_t)id + 1;   // (push() does not touch this map, but re-fetch the slot anyway)
    if (!has_input || slot > max_input_slot) max_input_slot = slot;
    has_input = true;
    return id;
}

uint32_t FlatGraph::make(FlatOp op, uint32_t a, uint32_t b) {
    if (op == OP_DELAY) {
        // reference.rs:197-216 with a constant amount resolved now.
        if (is_const(a, 0.0f)) return a;  // every branch of Delay yields +0.0 when the source is +0.0
        if (is_const(b)) {
            float d = const_val(b);
            if (d >= 18446744073709551616.0f) return konst(0);
            if (sparkle && !(d >= 0.0f)) return konst(0);     // sparkle.rs:531-534: amount `ult 0` (negative, NaN) -> 0.0
            uint64_t di = (d < 0.0f || d != d) ? 0 : (uint64_t)d;
            if (di == 0) return a;
        }
    } else {
        if (is_const(a) && is_const(b)) return konst(f32_to_bits(host_binop(op, const_val(a), const_val(b), sparkle)));
        // a+b and a*b are bitwise commutative up to NaN payload, which is outside the contract.
        if ((op == OP_SUM2 || op == OP_MUL) && a > b) std::swap(a, b);
    }
    uint64_t key = ((uint64_t)op << 60) | ((uint64_t)a << 30) | b;
    uint64_t &e = cse_bin_.get(key);
    if (e) return (uint32_t)(e - 1);
    uint32_t d = 1 + std::max(nodes[a].depth, nodes[b].depth);
    uint32_t id = push(op, a, b, d);
    e = (uint64_t)id + 1;
    return id;
}

uint32_t FlatGraph::fbref() {
    const uint32_t j = (uint32_t)fb_target.size();
    fb_target.push_back(0);
    return push(OP_FBREF, j, 0, 0);   // (never hash-consed: one per cut cycle)
}

// ---- concurrent construction -------------------------------------------------------------------------
bool FlatGraph::par_begin(size_t max_new_nodes, size_t const_budget) {
    if (!nodes.mapped() || nodes.capacity() < nodes.size() + max_new_nodes + const_budget + 64) return false;
    cse_bin_.reserve(cse_bin_.size() + max_new_nodes);
    cse_[OP_CONST].reserve(cse_[OP_CONST].size() + const_budget + 1024);   // (+ the few a thread may add between its check and the others')
    cse_[OP_INPUT].reserve(cse_[OP_INPUT].size() + 4096);
    if (!cse_bin_.concurrent_ready(max_new_nodes) || !cse_[OP_CONST].concurrent_ready(const_budget) || !cse_[OP_INPUT].concurrent_ready(4096)) return false;
    par_next_s_.v = (uint32_t)nodes.size();
    par_input_claimed_ = 0;
    par_input_budget_ = 4096;
    par_const_pool_ = (int64_t)const_budget;
    par_const_grant_ = (uint32_t)std::max<size_t>(16, std::min<size_t>(1024, const_budget / 256));   // (64 threads strand a quarter of the pool at worst)
    return true;
}

bool FlatGraph::par_begin_in_place(size_t max_new_nodes, size_t const_budget) {
    // (an incremental update: growing a hash table means rehashing the whole graph's entries -- tens of milliseconds --
    //  which is what the caller is trying to avoid; the tables normally have a quarter to a half of their slots free)
    if (!nodes.mapped() || nodes.capacity() < nodes.size() + max_new_nodes + const_budget + 64) return false;
    // (a small table is rehashed in no time: give it room)
    if (cse_bin_.size() < (1u << 16)) cse_bin_.reserve(cse_bin_.size() + max_new_nodes);
    if (cse_[OP_CONST].size() < (1u << 16)) cse_[OP_CONST].reserve(cse_[OP_CONST].size() + const_budget + 1024);
    if (!cse_bin_.concurrent_ready(max_new_nodes) || !cse_[OP_CONST].concurrent_ready(const_budget + 1024)) return false;
    par_next_s_.v = (uint32_t)nodes.size();
    par_const_pool_ = (int64_t)const_budget;
    par_const_grant_ = (uint32_t)std::max<size_t>(16, std::min<size_t>(1024, const_budget / 256));
    par_input_claimed_ = 0;
    par_input_budget_ = 0;
    while (cse_[OP_INPUT].concurrent_ready(par_input_budget_ + 128) && par_input_budget_ < 4096) par_input_budget_ += 128;   // what it has room for
    return true;
}

uint32_t FlatGraph::par_push(FlatOp op, uint32_t a, uint32_t b, uint32_t depth, ParCounters &c) {
    const uint32_t above = (op == OP_CONST || op == OP_INPUT) ? 0u : std::max(a, b);   // operands must have smaller ids
    uint32_t id;
    if (c.id_next != c.id_end && c.id_next <= above) {
        // an operand another thread made, from a later block (voices an octave apart share every second partial's
        // sub-expression through hash-consing): this node, and then its users up the tree, take single ids from the counter --
        // above everything handed out so far -- while the thread's block stays in use for what does not depend on them
        id = __atomic_fetch_add(&par_next_s_.v, 1u, __ATOMIC_RELAXED);
        if (id >= 0x3FFFFFFFu) throw Error(FR_ERR_UNSUPPORTED, "lowered graph exceeds 2^30 nodes");
    } else {
        if (c.id_next == c.id_end) {
            c.id_next = __atomic_fetch_add(&par_next_s_.v, PAR_ID_BLOCK, __ATOMIC_RELAXED);
            c.id_end = c.id_next + PAR_ID_BLOCK;
            if (c.id_end >= 0x3FFFFFFFu) throw Error(FR_ERR_UNSUPPORTED, "lowered graph exceeds 2^30 nodes");
        }
        id = c.id_next++;
    }
    nodes.data()[id] = FlatNode{op, a, b, depth};   // (published by whoever hands the id on: release store of the table entry / memo)
    return id;
}

uint32_t FlatGraph::par_konst(uint32_t bits, ParCounters &c) {
    if (c.new_const >= c.const_budget) {   // a further grant from what the constants' table was sized for (threads take unequal shares:
        //                                     whoever starts first scans more of the node table)
        const int64_t grant = par_const_grant_;
        const int64_t before = __atomic_fetch_sub(&par_const_pool_, grant, __ATOMIC_RELAXED);
        if (before < grant) { __atomic_fetch_add(&par_const_pool_, grant, __ATOMIC_RELAXED); throw ParBudget{}; }
        c.const_budget += (size_t)grant;
    }
    return (uint32_t)(cse_[OP_CONST].concurrent_get(bits, [&] { return (uint64_t)par_push(OP_CONST, bits, 0, 0, c) + 1; }, &c.new_const) - 1);
}

uint32_t FlatGraph::par_input(uint32_t slot, ParCounters &c) {
    // (new input slots are rare; one shared count against what the table was found to have room for)
    if (__atomic_load_n(&par_input_claimed_, __ATOMIC_RELAXED) + 64 >= par_input_budget_) throw ParBudget{};
    const uint32_t id = (uint32_t)(cse_[OP_INPUT].concurrent_get(slot, [&] {
        __atomic_fetch_add(&par_input_claimed_, (size_t)1, __ATOMIC_RELAXED);
        return (uint64_t)par_push(OP_INPUT, slot, 0, 0, c) + 1;
    }, &c.new_input) - 1);
    if (!c.has_input || slot > c.max_input_slot) c.max_input_slot = slot;
    c.has_input = true;
    return id;
}

uint32_t FlatGraph::par_make(FlatOp op, uint32_t a, uint32_t b, ParCounters &c) {   // make(), on the concurrent tables
    const FlatNode *nd = nodes.data();
    auto kconst = [&](uint32_t id) { return nd[id].op == OP_CONST; };
    if (op == OP_DELAY) {
        if (kconst(a) && nd[a].a == f32_to_bits(0.0f)) return a;
        if (kconst(b)) {
            const float d = f32_from_bits(nd[b].a);
            if (d >= 18446744073709551616.0f) return par_konst(0, c);
            if (sparkle && !(d >= 0.0f)) return par_konst(0, c);
            const uint64_t di = (d < 0.0f || d != d) ? 0 : (uint64_t)d;
            if (di == 0) return a;
        }
    } else {
        if (kconst(a) && kconst(b)) return par_konst(f32_to_bits(host_binop(op, f32_from_bits(nd[a].a), f32_from_bits(nd[b].a), sparkle)), c);
        if ((op == OP_SUM2 || op == OP_MUL) && a > b) std::swap(a, b);
    }
    const uint64_t key = ((uint64_t)op << 60) | ((uint64_t)a << 30) | b;
    return (uint32_t)(cse_bin_.concurrent_get(key, [&] {
        const uint32_t d = 1 + std::max(nd[a].depth, nd[b].depth);
        if (d > c.max_depth) c.max_depth = d;
        return (uint64_t)par_push(op, a, b, d, c) + 1;
    }, &c.new_bin) - 1);
}

void FlatGraph::par_end(const std::vector<ParCounters> &threads) {
    size_t nc = 0, ni = 0, nb = 0;
    for (const ParCounters &c : threads) {
        nc += c.new_const; ni += c.new_input; nb += c.new_bin;
        if (c.max_depth > max_depth) max_depth = c.max_depth;
        if (c.has_input && (!has_input || c.max_input_slot > max_input_slot)) max_input_slot = c.max_input_slot;
        has_input = has_input || c.has_input;
    }
    cse_[OP_CONST].concurrent_added(nc);
    cse_[OP_INPUT].concurrent_added(ni);
    cse_bin_.concurrent_added(nb);
    nodes.set_size(__atomic_load_n(&par_next_s_.v, __ATOMIC_ACQUIRE));
}

// ---- lowering --------------------------------------------------------------------------------------

namespace {

struct Ctx {
    int parent;              // -1 for the root
    const MNode *inst;       // the composite node (in the parent's graph) this context instantiates
    const SubGraph *g;       // null for the root
};

struct Frame {
    int ctx;
    const MNode *node;
    int next;           // operand being resolved (0, 1) or 2 = ready
    uint32_t vals[2];
};


struct UserCell {
    uint64_t user;
    uint32_t next;   // cell index + 1, 0 = end
};

}  // namespace

struct Lowering::Impl {
    const Mirror *m = nullptr;
    FlatGraph fg;
    std::vector<Ctx> ctxs;
    // keyed by (context << 32 | dense position of the node inside its graph): exact, no pointer hashing
    // Context 0 (the top-level graph, where almost all nodes of a big patch live) is indexed directly by position;
    // the hash maps serve the contexts of composite instances.
    struct Table {
        VArray<uint64_t> top;
        FlatMap64 rest;
        uint64_t *find(uint64_t k) {
            if (!(k >> 32)) return (uint32_t)k < top.size() ? &top[(uint32_t)k] : nullptr;
            return rest.find(k);
        }
        uint64_t &get(uint64_t k) {
            if (!(k >> 32)) {
                const size_t p = (uint32_t)k;
                if (p >= top.size()) top.resize(p + 1, 0);   // (no copy: VArray)
                return top[p];
            }
            return rest.get(k);
        }
        void clear() { top.clear(); rest.clear(); }
    };
    Table child_ctx;   // -> context id + 1 (0: forget, the instance was replaced)
    Table memo;        // -> lowered id + 2; 1 = evaluation in progress; 0 = invalidated / never lowered
    // readers: key -> list of keys whose lowering looked at `key` (as an operand, or -- for a top-level composite
    // instance -- by following one of its inbound edges or outputs)
    Table readers_head;   // -> cell index + 1
    VArray<UserCell> cells;
    std::vector<uint32_t> free_cells;
    bool valid = false;
    uint32_t row_lo = 0, row_hi = 0xFFFFFFFFu;   // output slots lowered (Lowering::update)
    uint64_t generation = 0, relowered = 0;
    bool was_full = true;
    size_t base_nodes = 0, base_ctxs = 0, base_cells = 0;

    static constexpr uint64_t NOBODY = ~0ull;

    void reset(const Mirror &mm) {
        m = &mm;
        fg = FlatGraph();
        fg.sparkle = mm.sparkle;
        ctxs.clear();
        ctxs.push_back(Ctx{-1, nullptr, nullptr});
        child_ctx.clear();
        memo.clear();
        readers_head.clear();
        memo.top.reserve(mm.nodes.capacity_positions());
        readers_head.top.reserve(mm.nodes.capacity_positions());
        cells.clear();
        free_cells.clear();
        fg.reserve_nodes(mm.nodes.size());
        fg.konst(0);   // node 0 is always +0.0
        ++generation;
    }

    void add_reader(uint64_t of, uint64_t reader) {
        if (reader == NOBODY) return;
        uint64_t &head = readers_head.get(of);
        if (head && cells[head - 1].user == reader) return;   // both operands the same node
        uint32_t c;
        if (!free_cells.empty()) { c = free_cells.back(); free_cells.pop_back(); }
        else { c = (uint32_t)cells.size(); cells.push_back(UserCell{}); }
        cells[c] = UserCell{reader, (uint32_t)head};
        head = (uint64_t)c + 1;
    }

    // `k` changed: forget its lowering and that of everything that read it, transitively.  A reader whose entry is
    // already invalid was handled when it became invalid (nothing can have read it since), so the walk stops there.
    void invalidate(uint64_t k0) {
        std::vector<uint64_t> work{k0};
        while (!work.empty()) {
            uint64_t k = work.back();
            work.pop_back();
            if (uint64_t *mv = memo.find(k)) *mv = 0;
            uint64_t *head = readers_head.find(k);
            if (!head) continue;
            uint32_t c = (uint32_t)*head;
            *head = 0;   // the readers re-register when they are lowered again
            while (c) {
                UserCell cell = cells[c - 1];
                free_cells.push_back(c - 1);
                uint64_t *rv = memo.find(cell.user);
                if (rv && *rv >= 2) work.push_back(cell.user);
                c = cell.next;
            }
        }
    }

    const MNode *find(int ctx, uint32_t handle) const {
        const SubGraph *g = ctxs[ctx].g;
        if (!g) return m->nodes.find(handle);
        auto it = g->index.find(handle);
        return it == g->index.end() ? nullptr : &g->nodes[it->second];
    }
    uint64_t key(int ctx, const MNode *n) const {
        const SubGraph *g = ctxs[ctx].g;
        uint32_t pos = g ? (uint32_t)(n - g->nodes.data()) : m->nodes.position(n);
        return ((uint64_t)(uint32_t)ctx << 32) | pos;
    }

    // Follows an edge through graph inputs and composite outputs until it lands on a value that is
    // already known (returns true, id in `out`) or on a primitive node still to be evaluated
    // (returns false, `need` filled).  Mirrors the dispatch at reference.rs:178-195.  `reader` is the node on
    // whose behalf the edge is followed: it is registered with every mutable thing the walk depends on.
    bool resolve(int ctx, EdgeRef ref, uint32_t &out, Frame &need, uint64_t reader) {
        for (uint64_t hops = 0;; ++hops) {
            if (hops > (1u << 20)) throw Error(FR_ERR_CYCLE, "edge chain through composite I/O never reaches a node");
            if (!ref.present) {  // get_maybe_edge_value: missing edge is 0f32 (reference.rs:164-173)
                out = fg.konst(0);
                return true;
            }
            if (ref.from == 0) {  // reading this graph's input (reference.rs:181-183)
                const Ctx &c = ctxs[ctx];
                if (c.parent < 0) {
                    out = fg.input(ref.from_slot);
                    return true;
                }
                // closure of reference.rs:189-193: the instance's inbound[slot] in the parent graph
                if (c.parent == 0) add_reader(key(0, c.inst), reader);   // a top-level instance: its inbound edges can change
                uint32_t slot = ref.from_slot;
                ref = slot < c.inst->inbound.size() ? c.inst->inbound[slot] : EdgeRef{};
                ctx = c.parent;
                continue;
            }
            const MNode *n = find(ctx, ref.from);
            if (!n) throw Error(FR_ERR_NO_SUCH_NODE, "edge reads from unknown node " + std::to_string(ref.from));
            ++fg.n_mirror_nodes_visited;
            if (n->kind == FR_EFFECT_GRAPH) {  // reference.rs:188-194
                if (ctx == 0) add_reader(key(0, n), reader);             // the instance can be replaced or deleted
                uint64_t &cc_slot = child_ctx.get(key(ctx, n));
                if (!cc_slot) {
                    ctxs.push_back(Ctx{ctx, n, n->sub.get()});
                    cc_slot = ctxs.size();   // id + 1
                }
                int cc = (int)(cc_slot - 1);
                uint32_t slot = ref.from_slot;
                ref = slot < n->sub->outputs.size() ? n->sub->outputs[slot] : EdgeRef{};
                ctx = cc;
                continue;
            }
            if (n->kind == FR_PRIM_F32CONSTANT) {  // reference.rs:217-220
                out = fg.konst(ref.from_slot);
                return true;
            }
            if (ref.from_slot != 0)  // assert!(from_slot == 0), reference.rs:199,223,...
                throw Error(FR_ERR_BAD_SLOT, "primitive node " + std::to_string(ref.from) + " read through output slot " +
                                                 std::to_string(ref.from_slot));
            const uint64_t k = key(ctx, n);
            add_reader(k, reader);
            if (uint64_t *mv = memo.find(k)) {
                if (*mv == 1) {   // a dependency cycle: eval() cuts it at a Delay on the way round, if there is one
                    cycle_at = Frame{ctx, n, 0, {0, 0}};
                    cycle_handle = ref.from;
                    need = cycle_at;
                    return false;
                }
                if (*mv >= 2) {
                    out = (uint32_t)(*mv - 2);
                    return true;
                }
            }
            need = Frame{ctx, n, 0, {0, 0}};
            return false;
        }
    }

    static FlatOp op_of(int kind) {
        switch (kind) {
        case FR_PRIM_DELAY: return OP_DELAY;
        case FR_PRIM_SUM2: return OP_SUM2;
        case FR_PRIM_MULTIPLY: return OP_MUL;
        case FR_PRIM_DIVIDE: return OP_DIV;
        case FR_PRIM_MODULO: return OP_MOD;
        default: return OP_MIN;
        }
    }

    // Feedback.  The reference evaluates a graph with a dependency cycle by recursion all the same (its cycle check never
    // fires: routegraph.rs:218-237), and the recursion ends exactly when every trip round the cycle passes a Delay that
    // moves time back: get_edge_value(t) -> Delay -> get_edge_value(t - d) ... -> 0 once t < d (reference.rs:197-216).
    // When the walk below meets a node that is still being lowered, the cycle is CUT at the innermost Delay on the way round
    // whose SOURCE operand is being resolved: that operand becomes an OP_FBREF leaf, the frames above the Delay are
    // abandoned, and the source itself is lowered once the walk has come back down (`pending`), when everything it reaches
    // is in the memo.  No such Delay: a cycle with no delay in it, which the reference would recurse into forever.
    struct PendingTarget { uint32_t j; int ctx; EdgeRef source; };
    std::vector<PendingTarget> pending;
    Frame cycle_at{};            // set by resolve() when it returns false for a node that is in progress
    uint32_t cycle_handle = 0;

    // true: cut (the Delay's frame is now the top of the stack, its source resolved to an OP_FBREF)
    bool cut_cycle(std::vector<Frame> &stack) {
        size_t bottom = stack.size();
        for (size_t i = stack.size(); i-- > 0;)
            if (stack[i].ctx == cycle_at.ctx && stack[i].node == cycle_at.node) { bottom = i; break; }
        if (bottom == stack.size()) return false;   // (in progress but not on this stack: cannot happen on one thread)
        for (size_t i = stack.size(); i-- > bottom;) {
            Frame &d = stack[i];
            if (d.node->kind != FR_PRIM_DELAY || d.next != 0) continue;
            {   // only a Delay that always moves time back can end the recursion: its amount a constant of >= 1 frames, read
                // straight from the constant node (a Delay by 0 frames on the loop is a pass-through: try the next one out)
                const EdgeRef amt = d.node->inbound.size() > 1 ? d.node->inbound[1] : EdgeRef{};
                const MNode *src = (amt.present && amt.from != 0) ? find(d.ctx, amt.from) : nullptr;
                if (!src || src->kind != FR_PRIM_F32CONSTANT || !(f32_from_bits(amt.from_slot) >= 1.0f)) continue;
            }
            for (size_t k = stack.size(); k-- > i + 1;) memo.get(key(stack[k].ctx, stack[k].node)) = 0;
            stack.resize(i + 1);
            Frame &dd = stack.back();
            const uint32_t leaf = fg.fbref();
            pending.push_back(PendingTarget{fg.nodes[leaf].a, dd.ctx, dd.node->inbound.size() == 0 ? EdgeRef{} : dd.node->inbound[0]});
            dd.vals[0] = leaf;
            dd.next = 1;
            return true;
        }
        return false;
    }

    uint32_t eval(int ctx0, EdgeRef root) {
        uint32_t result = 0;
        Frame need;
        cycle_at.node = nullptr;
        if (resolve(ctx0, root, result, need, NOBODY)) return result;
        if (cycle_at.node) throw Error(FR_ERR_CYCLE, "internal: an output row resolves to a node that is being lowered");
        std::vector<Frame> stack;
        stack.push_back(need);
        memo.get(key(need.ctx, need.node)) = 1;
        try {
            while (!stack.empty()) {
                Frame &f = stack.back();
                if (f.next < 2) {
                    const MNode *n = f.node;
                    EdgeRef ref = (size_t)f.next < n->inbound.size() ? n->inbound[f.next] : EdgeRef{};
                    uint32_t id;
                    Frame child;
                    cycle_at.node = nullptr;
                    if (resolve(f.ctx, ref, id, child, key(f.ctx, n))) {
                        f.vals[f.next++] = id;
                    } else if (cycle_at.node) {
                        if (!cut_cycle(stack))
                            throw Error(FR_ERR_CYCLE, "dependency cycle through node " + std::to_string(cycle_handle) +
                                                          " with no Delay of a constant >= 1 frames on it (the reference would recurse forever)");
                    } else {
                        memo.get(key(child.ctx, child.node)) = 1;
                        stack.push_back(child);  // invalidates f; loop re-reads the top
                    }
                    continue;
                }
                if (f.node->kind == FR_PRIM_DELAY && fg.nodes[f.vals[0]].op == OP_FBREF) {
                    // the Delay that cuts a cycle must move time back by at least one frame, every time (reference.rs:200-215:
                    // NaN and negative amounts are 0 frames)
                    const float d = fg.is_const(f.vals[1]) ? fg.const_val(f.vals[1]) : 0.0f;
                    if (!fg.is_const(f.vals[1]) || !(d >= 1.0f))
                        throw Error(FR_ERR_CYCLE, std::string("dependency cycle closed through a Delay whose amount is ") +
                                                      (fg.is_const(f.vals[1]) ? "less than one frame" : "a signal") +
                                                      ": only feedback through a constant Delay of >= 1 frames is evaluable");
                }
                uint32_t id = fg.make(op_of(f.node->kind), f.vals[0], f.vals[1]);
                memo.get(key(f.ctx, f.node)) = (uint64_t)id + 2;
                ++relowered;
                stack.pop_back();
                if (stack.empty()) {
                    result = id;
                } else {
                    Frame &p = stack.back();
                    p.vals[p.next++] = id;
                }
            }
        } catch (...) {   // nothing half-evaluated may stay marked: the next update() must fail the same way
            for (const Frame &f : stack) memo.get(key(f.ctx, f.node)) = 0;
            throw;
        }
        return result;
    }

    // ---- parallel from-scratch lowering ---------------------------------------------------------------------------------
    // Output rows are independent sub-graphs for the most part (voices), and lowering one is a walk over hash tables that
    // misses the cache at every step: 0.25 us per node, 0.94 s for config C's 3.7 M nodes on one thread, inside the first
    // fill_buffer call.  Here the rows are dealt to threads; the shared structures take concurrent use as follows:
    //   * the flat graph: FlatGraph::par_* (ids from an atomic counter, lock-free insert-if-absent in the hash-consing maps);
    //   * memo (top-level table, pre-sized): 0 = untouched, id + 2 = final (release store), PAR_BUSY | thread = being lowered.
    //     A thread claims a node with a compare-and-swap; a thread that meets another thread's claim waits for the final value
    //     -- claims are held only along one root-to-leaf path, so two threads can wait for each other only around a real
    //     cycle, and a wait that lasts is handed to the sequential pass (as is a claim of one's own: a cycle);
    //   * readers lists: cells from an atomic counter, pushed with a compare-and-swap on the list head;
    //   * anything inside a composite instance (contexts, their hash-mapped tables) is NOT shared: a row that reaches a
    //     composite, or fails, is left to the sequential pass after the threads have joined -- which also reports errors in
    //     row order exactly as before.  What the threads leave behind is only ever final values of correctly lowered nodes.
    // Incremental updates (a few nodes per edit) stay sequential.
    static constexpr uint64_t PAR_BUSY = 1ull << 40;
    struct NeedsSequential {};
    struct alignas(128) ParThread {   // (a cache line pair of its own: the counters are written at every node)
        FlatGraph::ParCounters cnt;
        uint64_t relowered = 0, visited = 0, me = 0;
        uint32_t cell_next = 0, cell_end = 0;   // this thread's block of reader cells (one contended add per 4096)
        std::vector<Frame> stack;
    };
    struct alignas(128) ParCellsNext { uint32_t v = 0; char pad[124]; };
    ParCellsNext par_cells_next_s;   // (its own cache line, like FlatGraph's id counter)
    bool allow_parallel = true;   // (ids then depend on thread timing; Lowering::update's `deterministic` turns it off)
    uint64_t par_items = 0;       // sub-trees the last update lowered on threads

    void par_add_reader(ParThread &ts, uint64_t of, uint64_t reader) {
        if (reader == NOBODY) return;
        uint64_t *head = &readers_head.top[(uint32_t)of];
        uint64_t old = __atomic_load_n(head, __ATOMIC_ACQUIRE);
        if (old && cells.data()[old - 1].user == reader) return;   // both operands the same node
        if (ts.cell_next == ts.cell_end) {
            ts.cell_next = __atomic_fetch_add(&par_cells_next_s.v, 4096u, __ATOMIC_RELAXED);
            ts.cell_end = ts.cell_next + 4096u;
            if (ts.cell_end >= cells.capacity() || ts.cell_end < ts.cell_next) throw NeedsSequential{};
        }
        const uint32_t c = ts.cell_next++;
        do cells.data()[c] = UserCell{reader, (uint32_t)old};
        while (!__atomic_compare_exchange_n(head, &old, (uint64_t)c + 1, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE));
    }

    // resolve() for context 0 only; a composite instance ends the parallel attempt at this row
    bool par_resolve(ParThread &ts, EdgeRef ref, uint32_t &out, Frame &need, uint64_t reader) {
        if (!ref.present) { out = fg.par_konst(0, ts.cnt); return true; }
        if (ref.from == 0) { out = fg.par_input(ref.from_slot, ts.cnt); return true; }
        const MNode *n = m->nodes.find(ref.from);
        if (!n) throw Error(FR_ERR_NO_SUCH_NODE, "edge reads from unknown node " + std::to_string(ref.from));
        ++ts.visited;
        if (n->kind == FR_EFFECT_GRAPH) throw NeedsSequential{};
        if (n->kind == FR_PRIM_F32CONSTANT) { out = fg.par_konst(ref.from_slot, ts.cnt); return true; }
        if (ref.from_slot != 0) throw Error(FR_ERR_BAD_SLOT, "primitive node read through a non-zero output slot");
        const uint64_t k = m->nodes.position(n);
        par_add_reader(ts, k, reader);
        uint64_t *mv = &memo.top[(uint32_t)k];
        uint64_t spins = 0;
        for (;;) {
            uint64_t v = __atomic_load_n(mv, __ATOMIC_ACQUIRE);
            if (v >= 2 && v < PAR_BUSY) { out = (uint32_t)(v - 2); return true; }
            if (v == 0) {
                if (__atomic_compare_exchange_n(mv, &v, ts.me, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
                    need = Frame{0, n, 0, {0, 0}};
                    return false;
                }
                continue;
            }
            if (v == ts.me) throw NeedsSequential{};            // a cycle: the sequential pass reports it
            ++fr_trace_memo_spins;
            if (++spins > (1u << 22)) {   // another thread's claim that does not resolve: cycle across threads
                if (std::getenv("FR_LOWER_TRACE")) std::fprintf(stderr, "parallel lowering: thread %llx gave up waiting for node %u (kind %u, memo %llx), read by position %llu\n",
                                                                 (unsigned long long)ts.me, ref.from, (unsigned)n->kind, (unsigned long long)v, (unsigned long long)reader);
                throw NeedsSequential{};
            }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
    }

    uint32_t par_eval(ParThread &ts, EdgeRef root) {
        uint32_t result = 0;
        Frame need;
        if (par_resolve(ts, root, result, need, NOBODY)) return result;
        std::vector<Frame> &stack = ts.stack;
        stack.clear();
        stack.push_back(need);
        try {
            while (!stack.empty()) {
                Frame &f = stack.back();
                if (f.next < 2) {
                    const MNode *n = f.node;
                    EdgeRef ref = (size_t)f.next < n->inbound.size() ? n->inbound[f.next] : EdgeRef{};
                    uint32_t id;
                    Frame child;
                    if (par_resolve(ts, ref, id, child, m->nodes.position(n))) f.vals[f.next++] = id;
                    else stack.push_back(child);   // invalidates f; the loop re-reads the top
                    continue;
                }
                const uint32_t id = fg.par_make(op_of(f.node->kind), f.vals[0], f.vals[1], ts.cnt);
                __atomic_store_n(&memo.top[m->nodes.position(f.node)], (uint64_t)id + 2, __ATOMIC_RELEASE);
                ++ts.relowered;
                stack.pop_back();
                if (stack.empty()) result = id;
                else { Frame &p = stack.back(); p.vals[p.next++] = id; }
            }
        } catch (...) {   // give the claims back: whoever waits for them goes on, the sequential pass starts from a clean slate
            for (const Frame &f : stack) __atomic_store_n(&memo.top[m->nodes.position(f.node)], (uint64_t)0, __ATOMIC_RELEASE);
            stack.clear();
            throw;
        }
        return result;
    }

    // (read at every from-scratch lowering, which is rare: tests switch them inside one process)
    static unsigned par_threads() {
        if (const char *e = std::getenv("FR_LOWER_THREADS")) return (unsigned)std::max(1, std::atoi(e));
        const unsigned hw = std::thread::hardware_concurrency();
        return std::min(32u, std::max(1u, hw));   // (measured on the 256-thread host of an MI355X box: 16 / 32 / 64 threads 92 / 59 / 113 ms at config C)
    }
    static size_t par_min_nodes() {
        const char *e = std::getenv("FR_LOWER_PAR_MIN_NODES");
        return e ? (size_t)std::atoll(e) : (size_t)200000;
    }
    static size_t par_min_edit() {   // journalled edits (a new node and its two edges are three) from which an incremental update goes parallel
        const char *e = std::getenv("FR_LOWER_PAR_MIN_EDIT");
        return e ? (size_t)std::atoll(e) : (size_t)16384;
    }

    // par_eval for a node that is known to be a top-level primitive, not a constant (a sub-tree of the frontier below)
    void par_eval_node(ParThread &ts, const MNode *n) {
        uint64_t *mv = &memo.top[m->nodes.position(n)];
        uint64_t v = 0;
        if (!__atomic_compare_exchange_n(mv, &v, ts.me, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) return;   // lowered meanwhile, or another thread's
        std::vector<Frame> &stack = ts.stack;
        stack.clear();
        stack.push_back(Frame{0, n, 0, {0, 0}});
        try {
            while (!stack.empty()) {
                Frame &f = stack.back();
                if (f.next < 2) {
                    const MNode *nn = f.node;
                    EdgeRef ref = (size_t)f.next < nn->inbound.size() ? nn->inbound[f.next] : EdgeRef{};
                    uint32_t id;
                    Frame child;
                    if (par_resolve(ts, ref, id, child, m->nodes.position(nn))) f.vals[f.next++] = id;
                    else stack.push_back(child);
                    continue;
                }
                const uint32_t id = fg.par_make(op_of(f.node->kind), f.vals[0], f.vals[1], ts.cnt);
                __atomic_store_n(&memo.top[m->nodes.position(f.node)], (uint64_t)id + 2, __ATOMIC_RELEASE);
                ++ts.relowered;
                stack.pop_back();
                if (!stack.empty()) { Frame &p = stack.back(); p.vals[p.next++] = id; }
            }
        } catch (...) {
            for (const Frame &f : stack) __atomic_store_n(&memo.top[m->nodes.position(f.node)], (uint64_t)0, __ATOMIC_RELEASE);
            stack.clear();
            throw;
        }
    }

    // The parallel part of an update: the sub-trees hanging below the rows that still need lowering, on several threads.
    // Work items are the nodes of a FRONTIER: starting from those rows' roots, un-lowered top-level primitives are expanded
    // breadth-first into their operands until there are a few items per thread (64 voices give 128+ sub-trees; ONE new
    // voice -- a note-on -- gives 64 sub-trees of its Sum2 tree too).  What is above the frontier, rows through composites,
    // and anything that fails stays for the sequential pass that follows, which then finds the sub-trees in the memo.
    // `full`: a from-scratch lowering (tables sized here, leaf pre-pass); else an incremental one, taken only if the tables
    // have room for `expect_new` more entries as they are.  Returns false when the parallel form does not apply.
    bool lower_parallel(const Mirror &mm, uint32_t n_slots, bool full, size_t expect_new) {
        const uint32_t lo = std::min(row_lo, n_slots), hi = std::min(row_hi, n_slots);
        const unsigned want_threads = par_threads();
        const size_t positions = mm.nodes.capacity_positions();
        if (want_threads < 2 || hi <= lo || expect_new < (full ? par_min_nodes() : par_min_edit()) || positions >= (1ull << 31)) return false;
        const bool trace = std::getenv("FR_LOWER_TRACE") != nullptr;
        const auto t_begin = std::chrono::steady_clock::now();
        auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
        const size_t const_budget = full ? mm.nodes.size() / 8 + 65536 : expect_new / 2 + 4096;
        if (!cells.mapped() || !memo.top.mapped() || !readers_head.top.mapped()) return false;
        if (full ? !fg.par_begin(mm.nodes.size() + 16, const_budget) : !fg.par_begin_in_place(expect_new + 16, const_budget)) return false;
        memo.top.resize(positions, 0);
        readers_head.top.resize(positions, 0);
        par_cells_next_s.v = (uint32_t)cells.size();
        // ---- the frontier ----
        auto unlowered_primitive = [&](EdgeRef ref) -> const MNode * {
            if (!ref.present || ref.from == 0) return nullptr;
            const MNode *n = mm.nodes.find(ref.from);
            if (!n || n->kind == FR_EFFECT_GRAPH || n->kind == FR_PRIM_F32CONSTANT || ref.from_slot != 0) return nullptr;
            return memo.top[mm.nodes.position(n)] == 0 ? n : nullptr;
        };
        std::vector<const MNode *> frontier;
        {
            FlatMap64 seen;
            for (uint32_t s = lo; s < hi; ++s)
                if (const MNode *n = unlowered_primitive(s < mm.outputs.size() ? mm.outputs[s] : EdgeRef{})) {
                    bool fresh = false;
                    seen.get(mm.nodes.position(n), &fresh);
                    if (fresh) frontier.push_back(n);
                }
            const size_t want = (size_t)want_threads * 4;
            size_t head = 0;   // frontier[0, head) have been expanded: they stay for the sequential pass, their operands joined the frontier
            while (head < frontier.size() && frontier.size() - head < want && head < (1u << 16)) {
                const MNode *n = frontier[head++];
                for (size_t i = 0; i < 2 && i < n->inbound.size(); ++i)
                    if (const MNode *c = unlowered_primitive(n->inbound[i])) {
                        bool fresh = false;
                        seen.get(mm.nodes.position(c), &fresh);
                        if (fresh) frontier.push_back(c);
                    }
            }
            frontier.erase(frontier.begin(), frontier.begin() + (ptrdiff_t)head);
        }
        const unsigned nthreads = (unsigned)std::min<size_t>(want_threads, frontier.size());
        if (nthreads < 2) { std::vector<FlatGraph::ParCounters> none; fg.par_end(none); return false; }
        const double t_setup = since();
        std::vector<ParThread> ts(nthreads);
        std::atomic<bool> stop{false};
        for (unsigned t = 0; t < nthreads; ++t) { ts[t].me = PAR_BUSY | (uint64_t)(t + 1); ts[t].cnt.const_budget = 0; }
        auto run_threads = [&](const std::function<void(unsigned)> &fn) { LowerPool::get().run(nthreads, fn); };
        // Pass 1 (from-scratch only) -- every constant and input the top-level nodes read, made before any other node: leaves
        // have no operands, so their ids may come from any block, and with all of them below every block pass 2 draws, a node
        // made from a constant some OTHER thread made first keeps its place in its thread's block (the waveform's constants and
        // the amplitudes 1/(k+1) are shared by every voice).  A scan of the node table in position order: memory-bound.
        if (full) {
            std::atomic<size_t> next_chunk{0};
            run_threads([&](unsigned t) {
                ParThread &me = ts[t];
                uint32_t const_handle = 0;   // the last handle seen to be an F32Constant node (usually there is exactly one)
                try {
                    for (;;) {
                        const size_t c0 = next_chunk.fetch_add(1 << 14);
                        if (c0 >= positions || stop.load(std::memory_order_relaxed)) return;
                        const size_t c1 = std::min(positions, c0 + (1u << 14));
                        for (size_t p = c0; p < c1; ++p) {
                            const MNode &n = mm.nodes.by_position(p);
                            if (n.kind == FR_EFFECT_GRAPH || n.kind == FR_PRIM_F32CONSTANT) continue;
                            for (size_t i = 0; i < n.inbound.size() && i < 2; ++i) {
                                const EdgeRef ref = n.inbound[i];
                                if (!ref.present) continue;
                                if (ref.from == 0) { fg.par_input(ref.from_slot, me.cnt); continue; }
                                if (ref.from != const_handle) {
                                    const MNode *src = mm.nodes.find(ref.from);
                                    if (!src || src->kind != FR_PRIM_F32CONSTANT) continue;
                                    const_handle = ref.from;
                                }
                                fg.par_konst(ref.from_slot, me.cnt);
                            }
                        }
                    }
                } catch (...) {   // budget, too many input slots: no harm done, the rest goes the sequential way
                    stop.store(true);
                }
            });
            for (ParThread &x : ts) { x.cnt.id_next = x.cnt.id_end = 0; }   // pass 2 starts on fresh blocks, above every leaf
        }
        const double t_leaves = since();
        // Pass 2 -- the sub-trees
        std::atomic<size_t> next_item{0};
        struct ThreadTrace { double start = 0, wall = 0, sys = 0, worst = 0; long faults = 0; uint64_t spins[3] = {0, 0, 0}; };
        std::vector<ThreadTrace> tt(trace ? nthreads : 0);
        run_threads([&](unsigned t) {
            ParThread &me = ts[t];
            rusage r0{};
            if (trace) { getrusage(RUSAGE_THREAD, &r0); tt[t].start = since(); }
            for (;;) {
                if (stop.load(std::memory_order_relaxed)) break;
                const size_t i = next_item.fetch_add(1);
                if (i >= frontier.size()) break;
                const double t_item = trace ? since() : 0.0;
                try {
                    par_eval_node(me, frontier[i]);
                } catch (const FlatGraph::ParBudget &) {   // the constants' table is as full as it was sized for: the rest sequentially
                    if (trace && !stop.load()) std::fprintf(stderr, "  thread %u: a table is as full as it was sized for; the rest sequentially\n", t);
                    stop.store(true);
                } catch (...) {   // composite, error, suspected cycle: the sequential pass lowers this (and reports)
                    if (trace) std::fprintf(stderr, "  thread %u: sub-tree %zu left to the sequential pass\n", t, i);
                }
                if (trace) tt[t].worst = std::max(tt[t].worst, since() - t_item);
            }
            if (trace) {
                rusage r1{};
                getrusage(RUSAGE_THREAD, &r1);
                tt[t].wall = since() - tt[t].start;
                tt[t].sys = (r1.ru_stime.tv_sec - r0.ru_stime.tv_sec) * 1e3 + (r1.ru_stime.tv_usec - r0.ru_stime.tv_usec) * 1e-3;
                tt[t].faults = r1.ru_minflt - r0.ru_minflt;
                tt[t].spins[0] = fr_trace_val_spins; tt[t].spins[1] = fr_trace_memo_spins; tt[t].spins[2] = fr_trace_probe_steps;
                fr_trace_val_spins = fr_trace_memo_spins = fr_trace_probe_steps = 0;
            }
        });
        if (trace)
            for (unsigned t = 0; t < nthreads; ++t)
                if (tt[t].wall > 5.0 || tt[t].start - t_leaves > 5.0)
                    std::fprintf(stderr, "  thread %u: started %.1f ms into the pass, ran %.1f ms (kernel time %.1f ms, %ld page faults), longest sub-tree %.1f ms; spins: value %llu, memo %llu, probe steps %llu\n", t,
                                 tt[t].start - t_leaves, tt[t].wall, tt[t].sys, tt[t].faults, tt[t].worst, (unsigned long long)tt[t].spins[0], (unsigned long long)tt[t].spins[1], (unsigned long long)tt[t].spins[2]);
        std::vector<FlatGraph::ParCounters> cnts;
        for (ParThread &x : ts) { cnts.push_back(x.cnt); relowered += x.relowered; fg.n_mirror_nodes_visited += x.visited; }
        fg.par_end(cnts);
        cells.set_size(par_cells_next_s.v);
        par_items = frontier.size();
        if (trace) std::fprintf(stderr, "parallel lowering (%s), %u threads, %zu sub-trees: tables + frontier %.1f ms, leaves %.1f ms, sub-trees %.1f ms\n",
                                full ? "from scratch" : "incremental", nthreads, frontier.size(), t_setup, t_leaves - t_setup, since() - t_leaves);
        return true;
    }

    // journal == nullptr: from scratch
    const FlatGraph &update(const Mirror &mm, uint32_t n_slots, const std::vector<uint32_t> *journal) {
        // superseded nodes cost memory only (32 B each with their hash-table entry), a rebuild costs a stall: be generous
        const bool garbage = fg.nodes.size() > 8 * base_nodes + (1u << 20) || ctxs.size() > 8 * base_ctxs + (1u << 16) ||
                             cells.size() > 8 * base_cells + (1u << 22);
        const bool full = !valid || m != &mm || !journal || garbage;
        relowered = 0;
        par_items = 0;
        was_full = full;
        if (full) {
            reset(mm);
        } else {
            for (uint32_t entry : *journal) {
                const uint64_t k = entry & ~Mirror::JOURNAL_NODE;   // context 0
                if (entry & Mirror::JOURNAL_NODE)
                    if (uint64_t *cc = child_ctx.find(k)) *cc = 0;
                invalidate(k);
            }
        }
        valid = true;   // from here on the state matches the mirror even if an output fails to lower
        fg.outputs.assign(n_slots, 0);
        // big jobs -- a from-scratch lowering, or an edit that brought thousands of new nodes (a note-on) -- first lower the
        // sub-trees below the rows on several threads; the loop here then finds them in the memo
        if (allow_parallel) lower_parallel(mm, n_slots, full, full ? mm.nodes.size() : (journal ? journal->size() : 0));
        pending.clear();
        try {
            for (uint32_t s = 0; s < n_slots; ++s) {
                if (s < row_lo || s >= row_hi) { fg.outputs[s] = fg.konst(0); continue; }   // another rank's row
                EdgeRef ref = s < mm.outputs.size() ? mm.outputs[s] : EdgeRef{};  // reference.rs:158-161
                fg.outputs[s] = eval(0, ref);
            }
            while (!pending.empty()) {   // what the cut Delays read (may cut further cycles)
                const PendingTarget p = pending.back();
                pending.pop_back();
                fg.fb_target[p.j] = eval(p.ctx, p.source);
            }
        } catch (...) {
            if (!fg.fb_target.empty()) valid = false;   // half-cut cycles are in the memo: start over next time
            throw;
        }
        if (full) { base_nodes = fg.nodes.size(); base_ctxs = ctxs.size(); base_cells = cells.size(); }
        // (Feedback and incremental updates: the Delay that cuts a loop is registered as a reader of its source, so an edit
        //  anywhere on the loop invalidates it and the loop is cut again with a NEW OP_FBREF leaf; a leaf that survives an update
        //  still stands for the same node.  Leaves of loops that no longer exist stay behind as garbage: plan_stages counts only
        //  the ones the rendered rows reach.)
        return fg;
    }
};

Lowering::Lowering() : impl_(new Impl) {}
Lowering::~Lowering() = default;
const FlatGraph &Lowering::update(Mirror &m, uint32_t n_slots, uint32_t row_lo, uint32_t row_hi, bool deterministic) {
    impl_->allow_parallel = !deterministic;
    std::vector<uint32_t> journal;
    journal.swap(m.journal);
    bool usable = m.journal_on && !m.journal_overflow;
    m.journal_on = true;        // the invalidations below are applied before anything can throw
    m.journal_overflow = false;
    if (row_lo != impl_->row_lo || row_hi != impl_->row_hi) {
        impl_->row_lo = row_lo;
        impl_->row_hi = row_hi;
        usable = false;         // other rows than last time: from scratch
    }
    return impl_->update(m, n_slots, usable ? &journal : nullptr);
}
uint64_t Lowering::generation() const { return impl_->generation; }
bool Lowering::last_was_full() const { return impl_->was_full; }
uint64_t Lowering::last_relowered() const { return impl_->relowered; }
uint64_t Lowering::last_parallel_subtrees() const { return impl_->par_items; }

FlatGraph lower(const Mirror &m, uint32_t n_slots) {
    Lowering::Impl one_shot;
    one_shot.update(m, n_slots, nullptr);
    return std::move(one_shot.fg);
}

}  // namespace fr
