// graph.hpp -- renderer-side graph mirror and its lowered (flat) form.
//
// The mirror holds what RefRenderer's `NodeMap` holds (reference src/render/reference.rs:14-44):
// node handle -> {primitive kind | composite sub-graph, inbound edge per slot} plus the output edges.
// Unlike the reference it does not deep-copy a composite's sub-graph per instance
// (reference.rs:98-113): effects are immutable (src/routing/effect.rs:50-57), so identical
// definitions are interned once and shared by every instance.
//
// Lowering turns the nested mirror into a FlatGraph: one hash-consed DAG of primitive ops in
// topological order, composites inlined, F32Constant edges folded to constants.  This is sound
// because the reference evaluator is a pure function value(edge, t) (reference.rs:178-266): two
// structurally identical sub-expressions have identical values at every t.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <new>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/friendship_render.h"

namespace fr {

// Open-addressing hash map, u64 -> u64 (linear probing, power-of-two capacity).  Lowering a multi-million-node
// graph is dominated by hash lookups; this is ~3x faster than the node-based std::unordered_map for that use.
// Growth never stalls a caller: the table of twice the size comes from calloc (pages are touched lazily) and the old
// one is emptied into it a few slots per insertion, lookups consulting both meanwhile -- a doubling at 1.3 M nodes used
// to be a 0.1 s pause inside a fill_buffer call that followed a graph edit.
inline thread_local uint64_t fr_trace_val_spins = 0, fr_trace_memo_spins = 0, fr_trace_probe_steps = 0;
class FlatMap64 {
    struct Slot { uint64_t key1, val; };       // key1 = key + 1, 0 = empty (keys must not be ~0); one cache line per probe
    Slot *slots_ = nullptr;
    size_t cap_ = 0, mask_ = 0, n_ = 0;        // n_: distinct keys, wherever they currently live
    Slot *old_ = nullptr;                      // previous table, being migrated from old_pos_ on
    size_t old_cap_ = 0, old_pos_ = 0;
    static constexpr size_t MIGRATE_PER_INSERT = 16;
    // Groups of 16 consecutive keys stay adjacent (a few cache lines) and the groups are scattered by a full
    // mix: handles and node ids are mostly consecutive, so this keeps lowering cache-friendly without the long runs
    // that make linear probing degenerate under an identity hash.
    static uint64_t hash(uint64_t k) {
        uint64_t g = k >> 4;
        g ^= g >> 33; g *= 0xff51afd7ed558ccdULL; g ^= g >> 33; g *= 0xc4ceb9fe1a85ec53ULL; g ^= g >> 33;
        return (g << 4) | (k & 15u);
    }
    // Big tables come from mmap with transparent huge pages asked for: a from-scratch lowering touches every page of them
    // once, and with 4 KiB pages those first touches -- 74 000 of them at config C, each through the kernel's page-fault path
    // -- were the part of the parallel lowering that did not get faster with more threads.
    static constexpr size_t HUGE_BYTES = (size_t)2 << 20;
    static Slot *alloc(size_t cap);
    static void dealloc(Slot *p, size_t cap);
    static Slot *probe(Slot *t, size_t mask, uint64_t k) {   // the slot holding k, or the empty slot where it would go
        size_t i = hash(k) & mask;
        while (t[i].key1 != 0 && t[i].key1 != k + 1) i = (i + 1) & mask;
        return &t[i];
    }
    void migrate(size_t steps) {
        while (old_ && steps) {
            if (old_pos_ == old_cap_) {
                dealloc(old_, old_cap_);
                old_ = nullptr;
                old_cap_ = old_pos_ = 0;
                return;
            }
            const Slot &o = old_[old_pos_++];
            if (o.key1 == 0) continue;
            --steps;
            Slot *s = probe(slots_, mask_, o.key1 - 1);
            if (s->key1 == 0) { s->key1 = o.key1; s->val = o.val; }   // (else: already moved by a lookup)
        }
    }
    void grow() {
        migrate(~(size_t)0);                   // at most one table in migration (it finished long ago unless reserve() hurried)
        old_ = slots_;
        old_cap_ = cap_;
        old_pos_ = 0;
        cap_ = cap_ ? cap_ * 2 : 64;
        mask_ = cap_ - 1;
        slots_ = alloc(cap_);
        if (!old_cap_) { old_ = nullptr; }
    }
    void copy_from(const FlatMap64 &o) {
        if (!o.cap_) return;
        cap_ = o.cap_;
        mask_ = cap_ - 1;
        slots_ = alloc(cap_);
        auto put = [&](const Slot &e) {
            if (e.key1 == 0) return;
            Slot *s = probe(slots_, mask_, e.key1 - 1);
            if (s->key1 == 0) { *s = e; ++n_; }
        };
        for (size_t i = 0; i < o.cap_; ++i) put(o.slots_[i]);
        for (size_t i = o.old_pos_; i < o.old_cap_; ++i) put(o.old_[i]);
    }
    void steal(FlatMap64 &o) {
        slots_ = o.slots_; cap_ = o.cap_; mask_ = o.mask_; n_ = o.n_; old_ = o.old_; old_cap_ = o.old_cap_; old_pos_ = o.old_pos_;
        o.slots_ = o.old_ = nullptr;
        o.cap_ = o.mask_ = o.n_ = o.old_cap_ = o.old_pos_ = 0;
    }

public:
    FlatMap64() = default;
    FlatMap64(const FlatMap64 &o) { copy_from(o); }
    FlatMap64(FlatMap64 &&o) noexcept { steal(o); }
    FlatMap64 &operator=(const FlatMap64 &o) {
        if (this != &o) { clear(); copy_from(o); }
        return *this;
    }
    FlatMap64 &operator=(FlatMap64 &&o) noexcept {
        if (this != &o) { clear(); steal(o); }
        return *this;
    }
    ~FlatMap64() { clear(); }
    size_t size() const { return n_; }
    void reserve(size_t n) {                   // bulk loads: size once, up front
        while (cap_ * 3 < n * 4 + 4) grow();
        migrate(~(size_t)0);
    }
    const uint64_t *find(uint64_t k) const {
        if (!cap_) return nullptr;
        const Slot *s = probe(slots_, mask_, k);
        if (s->key1) return &s->val;
        if (old_) {
            const Slot *o = probe(old_, old_cap_ - 1, k);
            if (o->key1) return &o->val;
        }
        return nullptr;
    }
    // mutable lookup: an entry still in the old table is moved first, so the pointer stays good until the next grow
    uint64_t *find(uint64_t k) {
        if (!cap_) return nullptr;
        Slot *s = probe(slots_, mask_, k);
        if (s->key1) return &s->val;
        if (old_) {
            const Slot *o = probe(old_, old_cap_ - 1, k);
            if (o->key1) { s->key1 = o->key1; s->val = o->val; return &s->val; }
        }
        return nullptr;
    }
    // value slot for k, inserted as 0 if absent; `inserted` tells which
    uint64_t &get(uint64_t k, bool *inserted = nullptr) {
        if ((n_ + 1) * 4 > cap_ * 3) grow();
        migrate(MIGRATE_PER_INSERT);
        if (inserted) *inserted = false;
        if (uint64_t *v = find(k)) return *v;
        Slot *s = probe(slots_, mask_, k);
        s->key1 = k + 1;
        s->val = 0;
        ++n_;
        if (inserted) *inserted = true;
        return s->val;
    }
    void clear() {
        dealloc(slots_, cap_);
        dealloc(old_, old_cap_);
        slots_ = old_ = nullptr;
        cap_ = mask_ = n_ = old_cap_ = old_pos_ = 0;
    }

    // ---- lock-free insert-if-absent, for the parallel from-scratch lowering (graph.cpp Lowering::Impl::update_parallel) ----
    // The table never grows here: it must have been sized for everything that can arrive (reserve(), which also finishes
    // any migration; concurrent_ready() says so).  make() runs exactly once per new key, in the thread that claimed the slot,
    // and returns a NON-ZERO value; a thread that finds the key claimed but unpublished waits for the value (the claimer is a
    // few instructions from publishing it and takes no lock and no other slot meanwhile).  Afterwards the owner calls
    // concurrent_added(n) with the number of keys the threads inserted, and the map is an ordinary one again.
    bool concurrent_ready(size_t extra) const { return cap_ && !old_ && (n_ + extra) * 4 <= cap_ * 3; }
    template <class F>
    uint64_t concurrent_get(uint64_t k, F &&make, size_t *inserted) {
        size_t i = hash(k) & mask_;
        for (;;) {
            uint64_t cur = __atomic_load_n(&slots_[i].key1, __ATOMIC_ACQUIRE);
            if (cur == 0) {
                uint64_t expect = 0;
                if (__atomic_compare_exchange_n(&slots_[i].key1, &expect, k + 1, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
                    const uint64_t v = make();
                    __atomic_store_n(&slots_[i].val, v, __ATOMIC_RELEASE);
                    ++*inserted;
                    return v;
                }
                cur = expect;
            }
            if (cur == k + 1) {
                uint64_t v;
                while (!(v = __atomic_load_n(&slots_[i].val, __ATOMIC_ACQUIRE))) {
                    ++fr_trace_val_spins;
#if defined(__x86_64__)
                    __builtin_ia32_pause();
#endif
                }
                return v;
            }
            i = (i + 1) & mask_;
            ++fr_trace_probe_steps;
        }
    }
    void concurrent_added(size_t n) { n_ += n; }
};

// Growable array for the engine's multi-million-element tables (lowered nodes, memo tables, reader lists) that never
// moves its elements: the address range for the largest size it may ever reach is reserved once (mmap, no pages
// committed) and the kernel backs pages as they are first touched.  A std::vector doubling one of these tables copies
// tens of megabytes inside whichever fill_buffer call follows the edit that crossed the capacity -- measured 80-110 ms
// for a note-on at config C size (profiles/r02_edit_latency.txt); here growth costs page faults only.
// T must be trivially copyable and valid when all-zero.  Falls back to an ordinary doubling buffer when the
// reservation is refused (strict overcommit).
template <class T>
class VArray {
    T *p_ = nullptr;
    size_t n_ = 0, cap_ = 0;      // cap_: elements the current mapping / buffer can hold
    bool mapped_ = false;
    static constexpr size_t RESERVE_BYTES = (size_t)16 << 30;   // address space, not memory
    void map_once();
    void grow_to(size_t need);
    void release();

public:
    VArray() = default;
    VArray(const VArray &o) { *this = o; }
    VArray(VArray &&o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_), mapped_(o.mapped_) { o.p_ = nullptr; o.n_ = o.cap_ = 0; o.mapped_ = false; }
    VArray &operator=(const VArray &o) {
        if (this != &o) {
            n_ = 0;
            if (o.n_) { grow_to(o.n_); std::memcpy(p_, o.p_, o.n_ * sizeof(T)); n_ = o.n_; }
        }
        return *this;
    }
    VArray &operator=(VArray &&o) noexcept {
        if (this != &o) { release(); p_ = o.p_; n_ = o.n_; cap_ = o.cap_; mapped_ = o.mapped_; o.p_ = nullptr; o.n_ = o.cap_ = 0; o.mapped_ = false; }
        return *this;
    }
    ~VArray() { release(); }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    T *data() { return p_; }
    const T *data() const { return p_; }
    T &operator[](size_t i) { return p_[i]; }
    const T &operator[](size_t i) const { return p_[i]; }
    T &back() { return p_[n_ - 1]; }
    const T &back() const { return p_[n_ - 1]; }
    T *begin() { return p_; }
    T *end() { return p_ + n_; }
    const T *begin() const { return p_; }
    const T *end() const { return p_ + n_; }
    void reserve(size_t) {}                       // nothing to do: growth never copies
    void push_back(const T &v) {
        if (n_ == cap_) grow_to(n_ + 1);
        p_[n_++] = v;
    }
    void resize(size_t n, const T &v = T()) {     // new elements take `v`
        if (n > cap_) grow_to(n);
        for (size_t i = n_; i < n; ++i) p_[i] = v;
        n_ = n;
    }
    void clear();                                 // gives the pages back
    // Concurrent appends (parallel lowering): with the address range mapped the threads write elements at indices they drew
    // from their own atomic counter -- nothing moves, pages appear as they are touched -- and the owner sets the size after.
    bool mapped() { if (!p_ && !mapped_) map_once(); return mapped_; }
    size_t capacity() const { return cap_; }
    void set_size(size_t n) { n_ = n; }
};

struct Error : std::runtime_error {
    fr_status code;
    Error(fr_status c, const std::string &m) : std::runtime_error(m), code(c) {}
};

// An inbound/output edge slot: `Option<Edge>` reduced to the two fields the evaluator reads.
struct EdgeRef {
    uint32_t from = 0;       // 0 = the enclosing graph's input
    uint32_t from_slot = 0;
    bool present = false;
};

// Inbound edges of a node by to_slot.  Primitives have at most two inputs: those live inline, so a node is one
// allocation-free 64-byte record (a patch is millions of them); composite instances with more inputs spill to a vector.
class EdgeSlots {
    EdgeRef inl_[2];
    uint32_t n_ = 0;
    std::unique_ptr<std::vector<EdgeRef>> more_;   // slots 2.. when present

public:
    EdgeSlots() = default;
    EdgeSlots(const EdgeSlots &o) : n_(o.n_) {
        inl_[0] = o.inl_[0]; inl_[1] = o.inl_[1];
        if (o.more_) more_.reset(new std::vector<EdgeRef>(*o.more_));
    }
    EdgeSlots &operator=(const EdgeSlots &o) {
        if (this != &o) { EdgeSlots t(o); *this = std::move(t); }
        return *this;
    }
    EdgeSlots(EdgeSlots &&) noexcept = default;
    EdgeSlots &operator=(EdgeSlots &&) noexcept = default;
    size_t size() const { return n_; }
    const EdgeRef &operator[](size_t i) const { return i < 2 ? inl_[i] : (*more_)[i - 2]; }
    void set(uint32_t slot, const EdgeRef &r) {          // grows with empty slots like Vec::resize
        if (slot >= 2) {
            if (!more_) more_.reset(new std::vector<EdgeRef>());
            if (more_->size() <= slot - 2) more_->resize((size_t)slot - 1);
            (*more_)[slot - 2] = r;
        } else {
            inl_[slot] = r;
        }
        if (n_ <= slot) n_ = slot + 1;
    }
    void clear(uint32_t slot) {                          // Option::take on an existing slot; size unchanged
        if (slot >= n_) return;
        if (slot < 2) inl_[slot] = EdgeRef{};
        else (*more_)[slot - 2] = EdgeRef{};
    }
};

struct SubGraph;

struct MNode {
    int32_t kind = 0;                          // FR_PRIM_* or FR_EFFECT_GRAPH
    uint32_t pos = 0;                          // dense position inside the owning table (top level only)
    std::shared_ptr<const SubGraph> sub;       // composite definition (interned)
    EdgeSlots inbound;                         // by to_slot
};

// Immutable composite definition: nodes in a dense array, looked up by handle through `index`.
struct SubGraph {
    std::vector<uint32_t> handles;
    std::vector<MNode> nodes;
    std::unordered_map<uint32_t, uint32_t> index;  // handle -> position in nodes
    std::vector<EdgeRef> outputs;                  // by to_slot of edges to null
    uint64_t hash = 0;
    bool equals(const SubGraph &o) const;
};

class Mirror {
public:
    void add_node(uint32_t handle, const fr_effect *e);
    void del_node(uint32_t handle);
    void add_edge(const fr_edge &e);
    void del_edge(const fr_edge &e);

    // top-level nodes: handle -> MNode (stable storage + flat index; deleted entries are recycled)
    class NodeTable {
        std::deque<MNode> store_;                // stable addresses, no element moves on growth
        std::vector<uint32_t> free_;
        FlatMap64 index_;                        // handle -> position + 1 (0 = deleted)
    public:
        const MNode *find(uint32_t handle) const {
            const uint64_t *p = index_.find(handle);
            return (p && *p) ? &store_[*p - 1] : nullptr;
        }
        MNode *find(uint32_t handle) { return const_cast<MNode *>(static_cast<const NodeTable *>(this)->find(handle)); }
        const MNode &at(uint32_t handle) const {
            const MNode *n = find(handle);
            if (!n) throw std::out_of_range("no such node");
            return *n;
        }
        void set(uint32_t handle, MNode &&n) {   // HashMap::insert: replaces an existing entry
            uint64_t &pos = index_.get(handle);
            if (!pos) {
                if (!free_.empty()) { pos = free_.back() + 1; free_.pop_back(); }
                else { store_.emplace_back(); pos = store_.size(); }
            }
            n.pos = (uint32_t)(pos - 1);
            store_[pos - 1] = std::move(n);
        }
        void erase(uint32_t handle) {
            uint64_t *p = index_.find(handle);
            if (p && *p) { store_[*p - 1] = MNode{}; free_.push_back((uint32_t)(*p - 1)); *p = 0; }
        }
        size_t size() const { return store_.size() - free_.size(); }
        void reserve(size_t n) { index_.reserve(n); }   // batch inserts: one table allocation instead of repeated regrowth
        // position of a node inside the table (a dense id for per-node side tables)
        uint32_t position(const MNode *n) const { return n->pos; }
        size_t capacity_positions() const { return store_.size(); }
        const MNode &by_position(size_t pos) const { return store_[pos]; }   // (a recycled position holds an empty node)
    };
    NodeTable nodes;
    std::vector<EdgeRef> outputs;
    uint64_t version = 0;                        // bumped by every edit
    bool sparkle = false;                        // FR_SEMANTICS_SPARKLE: Minimum and Delay as SparkleRenderer computes them

    // Edit journal for incremental lowering (class Lowering).  While `journal_on`, every edit appends the dense
    // position of the top-level node it touched (| JOURNAL_NODE when the node itself was added, replaced or deleted).
    // `journal_overflow` means the journal does not describe every edit since it was cleared: lower from scratch.
    static constexpr uint32_t MAX_TO_SLOT = 1u << 20;   // input / output slots per node the engine accepts
    static constexpr uint32_t JOURNAL_NODE = 0x80000000u;
    static constexpr size_t JOURNAL_LIMIT = 1u << 16;   // (at least; see note())
    bool journal_on = false;
    bool journal_overflow = true;
    std::vector<uint32_t> journal;

private:
    void note(uint32_t entry) {   // up to half the graph's size in edits is still cheaper to replay than to start over
        if (!journal_on || journal.size() >= std::max<size_t>(JOURNAL_LIMIT, nodes.size() / 2)) journal_overflow = true;
        else journal.push_back(entry);
    }
    std::shared_ptr<const SubGraph> intern(const fr_effect *e, int depth);
    std::unordered_multimap<uint64_t, std::weak_ptr<const SubGraph>> interned_;
};

// ---- lowered form ----------------------------------------------------------------------------
enum FlatOp : uint32_t {
    OP_CONST = 0,   // a = f32 bits
    OP_INPUT = 1,   // a = external input slot
    OP_DELAY = 2,   // a = source, b = amount (frames)
    OP_SUM2 = 3,
    OP_MUL = 4,
    OP_DIV = 5,
    OP_MOD = 6,
    OP_MIN = 7,
    // Feedback (a dependency cycle closed through a Delay of a constant >= 1 frames, which the reference evaluates by plain
    // recursion: reference.rs:197-216; routegraph.rs:218-237 never refuses the edge).  The lowered graph stays a DAG: the
    // Delay on the cycle reads this leaf instead of its source, and FlatGraph::fb_target[a] names the node it stands for
    // (lowered after the Delay, so with a higher id).  Only ever the source operand of an OP_DELAY.
    OP_FBREF = 8,   // a = index into fb_target
};

struct FlatNode {
    uint32_t op, a, b, depth;   // depth = longest path to a leaf (pull-stack sizing)
};

struct FlatGraph {
    VArray<FlatNode> nodes;            // topological: operands precede users
    std::vector<uint32_t> outputs;     // per rendered slot, a node id
    uint32_t max_depth = 0;
    uint32_t max_input_slot = 0;       // highest OP_INPUT slot referenced (valid if has_input)
    bool sparkle = false;              // FR_SEMANTICS_SPARKLE (constant folding here, range analysis, kernels)
    bool has_input = false;
    uint64_t n_mirror_nodes_visited = 0;
    std::vector<uint32_t> fb_target;   // OP_FBREF a -> the node whose value it is (empty: no feedback in the graph)

    uint32_t fbref();                                    // a new OP_FBREF leaf; its target is set once the node is lowered
    uint32_t src_of(uint32_t delay_source) const {       // the source operand of a Delay, seen through OP_FBREF
        const FlatNode &s = nodes[delay_source];
        return s.op == OP_FBREF ? fb_target[s.a] : delay_source;
    }
    uint32_t konst(uint32_t bits);
    uint32_t input(uint32_t slot);
    uint32_t make(FlatOp op, uint32_t a, uint32_t b);   // hash-consing + constant folding
    bool is_const(uint32_t id) const { return nodes[id].op == OP_CONST; }
    bool is_const(uint32_t id, float v) const;
    float const_val(uint32_t id) const;
    void reserve_nodes(size_t n) { nodes.reserve(n); cse_bin_.reserve(n); }   // one allocation instead of repeated regrowth

    // ---- concurrent construction (the parallel from-scratch lowering; graph.cpp) ----------------------------------------
    // Between par_begin() and par_end() several threads may call the par_* forms of konst / input / make at once: node ids
    // come in per-thread blocks drawn from one atomic counter (one contended add per 4096 nodes instead of per node); a node
    // with an operand another thread made from a later block (shared sub-expressions) takes a single id from the same counter
    // instead -- above every id handed out so far -- so operands still precede users (the planner orders cut nodes by id).
    // Ids of a block that were never used are all-zero nodes (the constant +0.0): garbage like the nodes an incremental update
    // supersedes.  The hash-consing tables take lock-free insert-if-absent.
    struct ParCounters {
        size_t new_const = 0, new_input = 0, new_bin = 0, const_budget = 0;
        uint32_t max_depth = 0, max_input_slot = 0;
        uint32_t id_next = 0, id_end = 0;      // this thread's block of node ids
        bool has_input = false;
    };
    static constexpr uint32_t PAR_ID_BLOCK = 4096;
    struct ParBudget {};   // thrown by par_konst when the constants' table has taken what it was sized for: lower the rest sequentially
    bool par_begin(size_t max_new_nodes, size_t const_budget);   // false: this graph cannot (the node array is not a mapped range)
    bool par_begin_in_place(size_t max_new_nodes, size_t const_budget);   // the same without resizing any table: false if one lacks room
    uint32_t par_konst(uint32_t bits, ParCounters &c);
    uint32_t par_input(uint32_t slot, ParCounters &c);
    uint32_t par_make(FlatOp op, uint32_t a, uint32_t b, ParCounters &c);
    void par_end(const std::vector<ParCounters> &threads);

private:
    // next unassigned node id while a concurrent construction is open (atomic builtins).  On a cache line of its own: the
    // threads add to it, and every other member of this object is something they all READ at every node.
    struct alignas(128) ParNext { uint32_t v = 0; char pad[124]; };
    ParNext par_next_s_;
    size_t par_input_claimed_ = 0, par_input_budget_ = 0;
    uint32_t par_const_grant_ = 1024;                   // new constants a thread may make per draw from the pool below
    alignas(128) int64_t par_const_pool_ = 0;           // what is left of the room the constants' table was sized for
    uint32_t par_push(FlatOp op, uint32_t a, uint32_t b, uint32_t depth, ParCounters &c);
    FlatMap64 cse_[2];   // OP_CONST: bits -> node id + 1; OP_INPUT: slot -> node id + 1
    FlatMap64 cse_bin_;  // (op << 60 | a << 30 | b) -> node id + 1   (ids < 2^30)
    uint32_t push(FlatOp op, uint32_t a, uint32_t b, uint32_t depth);
};

// The lowered graph kept up to date across edits.  The first update() lowers everything reachable from the outputs;
// later ones re-lower only what the journalled edits can have changed: each lowered node remembers which nodes read
// it (and which composite instances' inbound edges were followed to reach it), an edit invalidates the touched node
// and, transitively, its readers, and evaluation of the outputs recomputes exactly those.  The FlatGraph is
// append-only and hash-consed, so an unchanged sub-expression keeps its id across updates (plans cache per id);
// superseded nodes stay behind as garbage until they outnumber the live graph 8:1, then everything is rebuilt.
class Mirror;
struct FlatGraph;
FlatGraph lower(const Mirror &m, uint32_t n_slots);

class Lowering {
public:
    Lowering();
    ~Lowering();
    Lowering(const Lowering &) = delete;
    Lowering &operator=(const Lowering &) = delete;
    // Brings the lowered form of the first n_slots outputs up to date and consumes the mirror's journal.
    // Throws fr::Error like lower(); the state stays consistent and the next update() throws again until the
    // graph is fixed.  The returned reference is stable for the life of this object.
    // [row_lo, row_hi): only these output slots are lowered (a rank of a voice-sharded job needs nothing else); the
    // others read as the constant 0 and are never planned.  A different range than last time lowers from scratch.
    // `deterministic`: node ids must come out the same for the same mirror wherever this runs (the ranks of a partial-block
    // sharded job order their exchange by them): a from-scratch lowering then stays on one thread.
    const FlatGraph &update(Mirror &m, uint32_t n_slots, uint32_t row_lo = 0, uint32_t row_hi = 0xFFFFFFFFu, bool deterministic = false);
    uint64_t generation() const;        // bumped by every from-scratch rebuild: ids of different generations are unrelated
    bool last_was_full() const;
    uint64_t last_relowered() const;    // nodes lowered by the last update()
    uint64_t last_parallel_subtrees() const;   // sub-trees the last update() lowered on threads (0: it ran on the calling thread alone)

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
    friend FlatGraph lower(const Mirror &m, uint32_t n_slots);
};

// Lowers the first n_slots output slots of the mirror.  Throws fr::Error (NO_SUCH_NODE, BAD_SLOT,
// CYCLE) where the reference's evaluation of those slots would panic or never terminate.
FlatGraph lower(const Mirror &m, uint32_t n_slots);

}  // namespace fr
#include <sys/mman.h>
namespace fr {
inline bool lowering_hugepages() {   // FR_LOWER_HUGEPAGES=0: leave the lowering's tables on ordinary pages
    static const bool on = [] { const char *e = std::getenv("FR_LOWER_HUGEPAGES"); return !(e && e[0] == '0'); }();
    return on;
}
inline FlatMap64::Slot *FlatMap64::alloc(size_t cap) {
    const size_t bytes = cap * sizeof(Slot);
    if (bytes >= HUGE_BYTES) {
        void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);   // zero pages, like calloc
        if (m == MAP_FAILED) throw std::bad_alloc();   // (dealloc() tells the two kinds apart by size alone)
        if (lowering_hugepages()) (void)madvise(m, bytes, MADV_HUGEPAGE);
        return (Slot *)m;
    }
    Slot *p = (Slot *)std::calloc(cap, sizeof(Slot));
    if (!p) throw std::bad_alloc();
    return p;
}
inline void FlatMap64::dealloc(Slot *p, size_t cap) {
    if (!p) return;
    if (cap * sizeof(Slot) >= HUGE_BYTES) munmap(p, cap * sizeof(Slot));
    else std::free(p);
}
template <class T>
void VArray<T>::map_once() {
    void *m = mmap(nullptr, RESERVE_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) return;
    if (lowering_hugepages()) (void)madvise(m, RESERVE_BYTES, MADV_HUGEPAGE);   // pages are touched once, in order, by the million: fault them in 2 MiB at a time
    p_ = (T *)m;
    cap_ = RESERVE_BYTES / sizeof(T);
    mapped_ = true;
}
template <class T>
void VArray<T>::grow_to(size_t need) {
    if (!p_ && !mapped_) map_once();
    if (mapped_) {
        if (need > cap_) throw std::bad_alloc();
        return;
    }
    size_t cap = std::max<size_t>(need, std::max<size_t>(cap_ * 2, 1024));   // fallback: a doubling buffer
    T *q = (T *)std::calloc(cap, sizeof(T));
    if (!q) throw std::bad_alloc();
    if (n_) std::memcpy(q, p_, n_ * sizeof(T));
    std::free(p_);
    p_ = q;
    cap_ = cap;
}
template <class T>
void VArray<T>::release() {
    if (mapped_) munmap(p_, RESERVE_BYTES);
    else std::free(p_);
    p_ = nullptr;
    n_ = cap_ = 0;
    mapped_ = false;
}
template <class T>
void VArray<T>::clear() {
    if (mapped_ && n_) madvise(p_, ((n_ * sizeof(T) + 4095) / 4096) * 4096, MADV_DONTNEED);   // pages back, range kept
    n_ = 0;
}

// Exactly-rounded host evaluation of one primitive (same semantics as the device code and as
// reference.rs:197-262); used for constant folding.
float host_binop(FlatOp op, float a, float b, bool sparkle = false);
float f32_from_bits(uint32_t b);
uint32_t f32_to_bits(float f);

}  // namespace fr
