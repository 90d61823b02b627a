// ref_renderer.cpp -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// A C++17 restatement of the reference's `RefRenderer` (src/render/reference.rs, 291 lines), the
// pure-Rust interpreter that defines what `Renderer::fill_buffer` means.  It is deliberately
// *structurally faithful*: pull-based, per-sample recursion, one hash-map probe per visit, no
// memoisation, nested NodeMaps deep-copied per composite instance -- because it doubles as the timed
// "CPU path" baseline (bench.py `cpu_baseline`, kind "port").  It is NOT the Rust binary: no Rust
// toolchain exists in this environment, so nothing of the reference was compiled.
//
// Parity pinning: this restatement is checked against every known-answer test the reference ships
// for the path -- 11 tests / 14 arrays in tests/render_prim.rs, tests/ext_input.rs,
// tests/load_effect.rs -- transcribed as data into tests/golden/reference_kat.json and replayed by
// tests/test_oracle_golden.py.  Behaviour those tests do not pin (documented choices):
//   * Minimum on NaN / +-0 ties: Rust >=1.20 core `f32::min` = (a < b || b.is_nan()) ? a : b.
//   * Delay amount NaN: `NaN as u64` = 0 (saturating-cast Rust; UB on 2017 nightlies).
//   * Delay amount < 0: clamps to 0 (RefRenderer, reference.rs:206-207), NOT SparkleRenderer's
//     "return 0.0" (sparkle.rs:531-534).
// fr_config.semantics = FR_SEMANTICS_SPARKLE switches exactly the two places where the reference's second
// renderer, SparkleRenderer, computes something else (also unpinned by any reference test):
//   * Minimum = select(fcmp ult a, b, a, b) (sparkle.rs:492-498): NaN in either operand returns `a`;
//   * Delay amount `ult 0` (negative or NaN) returns 0.0 from the function (sparkle.rs:525-542).
//
// Dependency cycles: like the reference (whose RouteGraph never refuses the edge, routegraph.rs:218-237) the oracle has no cycle
// check; get_edge_value recurses through a Delay at t - d exactly as reference.rs:197-216 does, so a loop closed through a Delay
// of >= 1 frames evaluates to the feedback filter the reference would compute (cost: t / d stack frames per sample, exponential
// in the number of taps) and a loop without one overflows the stack, as the reference's would.  No reference test holds a cyclic
// graph: parity for feedback is against this restatement only -- "parity unpinned" beyond it.  Tracks
// (fr_set_track_inputs) are ordinary stored inputs here; the declaration is accepted and ignored.
//
// Sharding (fr_set_shard): the oracle renders the rows a rank owns by plain evaluation -- output slots are
// independent (reference.rs:78-82), so that IS the unsharded result for those rows -- and leaves the rest
// untouched; FR_SHARD_GATHER moves rows to rank 0 through the host callback.  It never splits a voice.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
// The product library (libfriendship_hip.so) never links, loads or calls it.
//
// Exports the same C ABI as the product (include/friendship_render.h) plus three `fro_*` helpers.

#include "../include/friendship_render.h"

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

struct Panic : std::runtime_error {
    fr_status code;
    Panic(fr_status c, const std::string &m) : std::runtime_error(m), code(c) {}
};

// routing::Edge (routegraph.rs:38-44); handle 0 == None (nullable_int.rs:65-72).
using Edge = fr_edge;
using MaybeEdge = std::optional<Edge>;

struct NodeMap;

// reference.rs:31-44  (Node, MyNodeData)
struct Node {
    int kind = 0;                       // FR_PRIM_* or FR_EFFECT_GRAPH
    std::unique_ptr<NodeMap> user;      // MyNodeData::UserNode
    std::vector<MaybeEdge> inbound;     // indexed by slot
};

// dyn Fn(u64, u32) -> f32   (reference.rs:165,178)
struct InputGetter {
    virtual float get(uint64_t time, uint32_t slot) const = 0;
    virtual ~InputGetter() = default;
};

// reference.rs:14-18
struct NodeMap {
    std::unordered_map<uint32_t, Node> nodes;
    std::vector<MaybeEdge> output_edges;
    bool sparkle = false;   // FR_SEMANTICS_SPARKLE (set on every nested map by make_node)

    // reference.rs:141-153
    void add_edge(const Edge &e) {
        std::vector<MaybeEdge> *inbound;
        if (e.to == 0) {
            inbound = &output_edges;
        } else {
            auto it = nodes.find(e.to);
            if (it == nodes.end())  // `.unwrap()` on None, reference.rs:145
                throw Panic(FR_ERR_NO_SUCH_NODE, "add_edge: destination node " + std::to_string(e.to) + " unknown");
            inbound = &it->second.inbound;
        }
        size_t slot = e.to_slot;
        if (inbound->size() <= slot) inbound->resize(slot + 1);
        (*inbound)[slot] = e;
    }

    // reference.rs:158-161
    float get_output(uint64_t time, uint32_t slot, const InputGetter &gi) const {
        const MaybeEdge *oe = slot < output_edges.size() ? &output_edges[slot] : nullptr;
        return get_maybe_edge_value(time, oe, gi);
    }

    // reference.rs:164-173
    float get_maybe_edge_value(uint64_t time, const MaybeEdge *me, const InputGetter &gi) const {
        if (me && me->has_value()) return get_edge_value(time, **me, gi);
        return 0.0f;
    }

    static const MaybeEdge *inb(const Node &n, size_t slot) {
        return slot < n.inbound.size() ? &n.inbound[slot] : nullptr;
    }

    // reference.rs:178-266
    float get_edge_value(uint64_t time, const Edge &edge, const InputGetter &gi) const;
};

// Closure of reference.rs:189-193: the sub-graph's input slot2 is the parent node's inbound[slot2].
struct SubInputGetter final : InputGetter {
    const NodeMap *parent;
    const Node *node;
    const InputGetter *outer;
    float get(uint64_t time2, uint32_t slot2) const override {
        return parent->get_maybe_edge_value(time2, NodeMap::inb(*node, slot2), *outer);
    }
};

// Rust >= 1.20 core::num f32::min: `(if self < other || other.is_nan() { self } else { other }) * 1.0`
static inline float rust_min(float a, float b) {
    return ((a < b || b != b) ? a : b) * 1.0f;
}

float NodeMap::get_edge_value(uint64_t time, const Edge &edge, const InputGetter &gi) const {
    const uint32_t from_slot = edge.from_slot;
    if (edge.from == 0) return gi.get(time, from_slot);  // reading from an input, :181-183
    auto it = nodes.find(edge.from);                      // `&self.nodes[&from]`, :186
    if (it == nodes.end())
        throw Panic(FR_ERR_NO_SUCH_NODE, "edge reads from unknown node " + std::to_string(edge.from));
    const Node &node = it->second;
    auto need_slot0 = [&]() {
        if (from_slot != 0) throw Panic(FR_ERR_BAD_SLOT, "primitive output slot must be 0");
    };
    switch (node.kind) {
    case FR_EFFECT_GRAPH: {  // :188-194
        SubInputGetter sub;
        sub.parent = this;
        sub.node = &node;
        sub.outer = &gi;
        return node.user->get_output(time, from_slot, sub);
    }
    case FR_PRIM_DELAY: {  // :197-216
        need_slot0();
        float delay_frames = get_maybe_edge_value(time, inb(node, 1), gi);
        if (delay_frames >= 18446744073709551616.0f) return 0.0f;  // >= 2^64, :202-205
        if (sparkle && !(delay_frames >= 0.0f)) return 0.0f;       // sparkle.rs:531-534: `ult 0` (negative or NaN) -> ret 0
        uint64_t delay_int;
        if (delay_frames < 0.0f) delay_int = 0;                      // :206-207
        else if (delay_frames != delay_frames) delay_int = 0;        // NaN as u64 == 0
        else delay_int = (uint64_t)delay_frames;                     // flooring, :210
        if (delay_int > time) return 0.0f;                           // checked_sub -> None, :213
        return get_maybe_edge_value(time - delay_int, inb(node, 0), gi);
    }
    case FR_PRIM_F32CONSTANT: {  // :217-220
        float f;
        std::memcpy(&f, &from_slot, 4);
        return f;
    }
    case FR_PRIM_MULTIPLY: {  // :221-227
        need_slot0();
        float l = get_maybe_edge_value(time, inb(node, 0), gi);
        float r = get_maybe_edge_value(time, inb(node, 1), gi);
        return l * r;
    }
    case FR_PRIM_SUM2: {  // :228-234
        need_slot0();
        float l = get_maybe_edge_value(time, inb(node, 0), gi);
        float r = get_maybe_edge_value(time, inb(node, 1), gi);
        return l + r;
    }
    case FR_PRIM_DIVIDE: {  // :235-241
        need_slot0();
        float l = get_maybe_edge_value(time, inb(node, 0), gi);
        float r = get_maybe_edge_value(time, inb(node, 1), gi);
        return l / r;
    }
    case FR_PRIM_MINIMUM: {  // :242-248
        need_slot0();
        float l = get_maybe_edge_value(time, inb(node, 0), gi);
        float r = get_maybe_edge_value(time, inb(node, 1), gi);
        if (sparkle) return (l < r || l != l || r != r) ? l : r;   // sparkle.rs:495-496: select(fcmp ult l, r, l, r)
        return rust_min(l, r);
    }
    case FR_PRIM_MODULO: {  // :249-262
        need_slot0();
        float dividend = get_maybe_edge_value(time, inb(node, 0), gi);
        float divisor = get_maybe_edge_value(time, inb(node, 1), gi);
        float rem = std::fmod(dividend, divisor);  // Rust `%` on f32 == fmodf
        if (rem < 0.0f) return rem + divisor;
        return rem;
    }
    default:
        throw Panic(FR_ERR_INVALID_ARG, "corrupt node kind");
    }
}

// One stored input vector, `inputs[slot]` of reference.rs:25.  len() == zero_prefix + data.size();
// the zero prefix stands for the `resize(idx, 0f32)` of a seek / late creation (:56,:63) without
// allocating idx floats.
struct InputVec {
    uint64_t zero_prefix = 0;
    std::vector<float> data;
    uint64_t len() const { return zero_prefix + data.size(); }
    float at(uint64_t t) const { return t < zero_prefix ? 0.0f : data[t - zero_prefix]; }
};

}  // namespace

struct fr_renderer {
    NodeMap nodes;            // reference.rs:22
    // reference.rs:25 `inputs: Vec<Vec<f32>>`.  The reference grows this to n_slots*n_times vectors
    // (:60-65, a quirk: buff.len() is the element count).  Vectors that never received a row are
    // all-zero and unread, so they are kept implicit: `n_vecs` counts them, `implicit_len[i]` of
    // the segment list gives their length, and only rows that were ever fed are materialised.
    std::map<uint32_t, InputVec> fed;
    uint64_t n_vecs = 0;
    struct Seg { uint64_t first, last, len; };  // implicit vectors [first,last) have length len
    std::vector<Seg> segs;
    uint64_t head = 0;        // reference.rs:28
    unsigned threads = 1;
    std::string last_error;
    bool sparkle = false;
    // fr_set_shard
    uint32_t rank = 0, world = 1, shard_flags = 0;
    int shard_mode = FR_SHARD_NONE;
    fr_comm comm{};
    bool has_comm = false;
    void my_rows(uint32_t n_slots, uint32_t &lo, uint32_t &hi) const { rows_of(rank, n_slots, lo, hi); }
    void rows_of(uint32_t rk, uint32_t n_slots, uint32_t &lo, uint32_t &hi) const {
        lo = 0;
        hi = n_slots;
        if (world <= 1 || shard_mode == FR_SHARD_NONE) return;
        const uint32_t q = n_slots / world, r = n_slots % world;
        lo = rk * q + (rk < r ? rk : r);
        hi = lo + q + (rk < r ? 1u : 0u);
    }

    uint64_t implicit_len(uint64_t slot) const {
        for (const Seg &s : segs) if (slot >= s.first && slot < s.last) return s.len;
        return 0;
    }

    struct TopGetter final : InputGetter {  // reference.rs:91-95
        const fr_renderer *self;
        float get(uint64_t time2, uint32_t slot2) const override {
            if (slot2 >= self->n_vecs) return 0.0f;
            auto it = self->fed.find(slot2);
            if (it == self->fed.end()) return 0.0f;  // implicit vectors hold only zeros
            return time2 < it->second.len() ? it->second.at(time2) : 0.0f;
        }
    };

    float get_sample(uint64_t time, uint32_t slot) const {  // reference.rs:90-96
        TopGetter g;
        g.self = this;
        return nodes.get_output(time, slot, g);
    }

    // reference.rs:98-113
    static void make_node(Node &dst, const fr_effect *e, int depth, bool sparkle = false) {
        if (!e) throw Panic(FR_ERR_INVALID_ARG, "null effect");
        if (depth > 256) throw Panic(FR_ERR_INVALID_ARG, "effect nesting too deep");
        if (e->kind < 0 || e->kind > FR_EFFECT_GRAPH) throw Panic(FR_ERR_INVALID_ARG, "bad effect kind");
        dst.kind = e->kind;
        if (e->kind != FR_EFFECT_GRAPH) return;
        dst.user = std::make_unique<NodeMap>();
        dst.user->sparkle = sparkle;
        if ((e->n_nodes && (!e->node_handles || !e->node_effects)) || (e->n_edges && !e->edges))
            throw Panic(FR_ERR_INVALID_ARG, "composite effect with null arrays");
        for (uint32_t i = 0; i < e->n_nodes; ++i) {
            if (e->node_handles[i] == 0) throw Panic(FR_ERR_INVALID_ARG, "node handle 0 is reserved");
            Node n;
            make_node(n, e->node_effects[i], depth + 1, sparkle);
            dst.user->nodes[e->node_handles[i]] = std::move(n);
        }
        for (uint32_t i = 0; i < e->n_edges; ++i) dst.user->add_edge(e->edges[i]);
    }

    // reference.rs:47-75, input-store phase
    void store_inputs(uint32_t n_slots, uint64_t n_times, uint64_t idx, const float *in_data,
                      const uint64_t *offs, uint32_t n_rows) {
        if (idx != head) {  // seek, :52-58
            for (auto &kv : fed) {
                kv.second.data.clear();
                kv.second.zero_prefix = idx;
            }
            segs.clear();
            if (n_vecs) segs.push_back({0, n_vecs, idx});
        }
        uint64_t want = (uint64_t)n_slots * n_times;  // buff.len(), :60
        if (n_vecs < want) {
            segs.push_back({n_vecs, want, idx});
            n_vecs = want;
        }
        for (uint32_t r = 0; r < n_rows && r < n_vecs; ++r) {  // zip stops at the shorter, :68
            auto it = fed.find(r);
            if (it == fed.end()) {
                InputVec v;
                v.zero_prefix = implicit_len(r);
                it = fed.emplace(r, std::move(v)).first;
            }
            InputVec &v = it->second;
            if (v.len() != idx)  // assert_eq!, :69
                throw Panic(FR_ERR_INPUT_HISTORY, "input slot " + std::to_string(r) + " holds " +
                                                      std::to_string(v.len()) + " samples, expected idx=" +
                                                      std::to_string(idx));
            uint64_t rl = offs[r + 1] - offs[r];
            if (rl > n_times)  // assert!, :71 (checked before mutating so a failed call leaves state intact)
                throw Panic(FR_ERR_INPUT_TOO_LONG, "input row " + std::to_string(r) + " longer than the range rendered");
            v.data.insert(v.data.end(), in_data + offs[r], in_data + offs[r] + rl);
            float pad = v.len() ? v.at(v.len() - 1) : 0.0f;  // :72
            v.data.resize(v.data.size() + (n_times - rl), pad);  // :73
        }
    }

    void render(float *out, uint32_t n_slots, uint64_t n_times, uint64_t idx) {  // :77-84
        uint32_t row_lo, row_hi;
        my_rows(n_slots, row_lo, row_hi);
        auto rows = [&](uint32_t s0, uint32_t s1) {
            for (uint32_t slot = s0; slot < s1; ++slot)
                for (uint64_t time = idx; time < idx + n_times; ++time)
                    out[(size_t)slot * n_times + (time - idx)] = get_sample(time, slot);
        };
        unsigned nt = threads < 1 ? 1 : threads;
        if (nt > n_slots) nt = n_slots ? n_slots : 1;
        if (nt <= 1 || row_lo != 0 || row_hi != n_slots) {
            rows(row_lo, row_hi);
        } else {
            // Output slots are independent (the loop of reference.rs:78 carries no state).  Used only
            // for the "all host cores" baseline figure; default is the reference's single thread.
            std::vector<std::thread> pool;
            std::atomic<uint32_t> next{0};
            std::vector<std::string> errs(nt);
            std::vector<fr_status> codes(nt, FR_OK);
            for (unsigned t = 0; t < nt; ++t)
                pool.emplace_back([&, t] {
                    try {
                        for (;;) {
                            uint32_t s = next.fetch_add(1);
                            if (s >= n_slots) break;
                            rows(s, s + 1);
                        }
                    } catch (const Panic &p) {
                        codes[t] = p.code;
                        errs[t] = p.what();
                    }
                });
            for (auto &th : pool) th.join();
            for (unsigned t = 0; t < nt; ++t)
                if (codes[t] != FR_OK) throw Panic(codes[t], errs[t]);
        }
    }
};

namespace {
template <class F>
fr_status guarded(fr_renderer *r, F &&f) {
    if (!r) return FR_ERR_INVALID_ARG;
    try {
        f();
        r->last_error.clear();
        return FR_OK;
    } catch (const Panic &p) {
        r->last_error = p.what();
        return p.code;
    } catch (const std::bad_alloc &) {
        r->last_error = "out of memory";
        return FR_ERR_OUT_OF_MEMORY;
    } catch (const std::exception &e) {
        r->last_error = e.what();
        return FR_ERR_INVALID_ARG;
    }
}
}  // namespace

extern "C" {

fr_status fr_renderer_create(const fr_config *cfg, fr_renderer **out) {
    if (!out) return FR_ERR_INVALID_ARG;
    if (cfg && cfg->abi_version != FR_ABI_VERSION) return FR_ERR_INVALID_ARG;
    if (cfg && cfg->semantics != FR_SEMANTICS_REFERENCE && cfg->semantics != FR_SEMANTICS_SPARKLE) return FR_ERR_INVALID_ARG;
    *out = new (std::nothrow) fr_renderer();
    if (!*out) return FR_ERR_OUT_OF_MEMORY;
    (*out)->sparkle = cfg && cfg->semantics == FR_SEMANTICS_SPARKLE;   // (history_frames: the oracle keeps everything)
    (*out)->nodes.sparkle = (*out)->sparkle;
    return FR_OK;
}

void fr_renderer_destroy(fr_renderer *r) { delete r; }

fr_status fr_on_add_node(fr_renderer *r, uint32_t handle, const fr_effect *effect) {  // :117-120
    return guarded(r, [&] {
        if (handle == 0) throw Panic(FR_ERR_INVALID_ARG, "node handle 0 is reserved for graph I/O");
        Node n;
        fr_renderer::make_node(n, effect, 0, r->sparkle);
        r->nodes.nodes[handle] = std::move(n);  // HashMap::insert replaces
    });
}

fr_status fr_on_del_node(fr_renderer *r, uint32_t handle) {  // :121-123
    return guarded(r, [&] { r->nodes.nodes.erase(handle); });
}

fr_status fr_on_add_edge(fr_renderer *r, const fr_edge *edge) {  // :124-126
    return guarded(r, [&] {
        if (!edge) throw Panic(FR_ERR_INVALID_ARG, "null edge");
        r->nodes.add_edge(*edge);
    });
}

fr_status fr_on_del_edge(fr_renderer *r, const fr_edge *edge) {  // :127-136
    return guarded(r, [&] {
        if (!edge) throw Panic(FR_ERR_INVALID_ARG, "null edge");
        std::vector<MaybeEdge> *inbound;
        if (edge->to == 0) {
            inbound = &r->nodes.output_edges;
        } else {
            auto it = r->nodes.nodes.find(edge->to);
            if (it == r->nodes.nodes.end())  // `.expect("Attempt to delete edge, but it was never created!")`
                throw Panic(FR_ERR_NO_SUCH_NODE, "Attempt to delete edge, but it was never created!");
            inbound = &it->second.inbound;
        }
        if (edge->to_slot < inbound->size()) (*inbound)[edge->to_slot].reset();
    });
}

fr_status fr_on_add_nodes(fr_renderer *r, const uint32_t *handles, const fr_effect *const *effects, size_t n) {
    if (n && (!handles || !effects)) return FR_ERR_INVALID_ARG;
    for (size_t i = 0; i < n; ++i) {
        fr_status s = fr_on_add_node(r, handles[i], effects[i]);
        if (s != FR_OK) return s;
    }
    return FR_OK;
}

fr_status fr_on_add_edges(fr_renderer *r, const fr_edge *edges, size_t n) {
    if (n && !edges) return FR_ERR_INVALID_ARG;
    for (size_t i = 0; i < n; ++i) {
        fr_status s = fr_on_add_edge(r, &edges[i]);
        if (s != FR_OK) return s;
    }
    return FR_OK;
}

fr_status fr_fill_buffer(fr_renderer *r, float *out, uint32_t n_slots, uint64_t n_times, uint64_t idx,
                         const float *in_data, const uint64_t *in_row_offsets, uint32_t n_in_rows) {
    return guarded(r, [&] {
        if ((!out && n_slots != 0 && n_times != 0) || (n_in_rows && !in_row_offsets))
            throw Panic(FR_ERR_INVALID_ARG, "null buffer");
        if (n_in_rows && in_row_offsets[n_in_rows] > in_row_offsets[0] && !in_data)
            throw Panic(FR_ERR_INVALID_ARG, "null input data");
        r->store_inputs(n_slots, n_times, idx, in_data, in_row_offsets, n_in_rows);
        r->render(out, n_slots, n_times, idx);
        r->head = idx + n_times;  // :84
        if (r->world > 1 && r->shard_mode != FR_SHARD_NONE && (r->shard_flags & FR_SHARD_GATHER)) {
            if (!r->has_comm) throw Panic(FR_ERR_COMM, "FR_SHARD_GATHER needs a host transport");
            auto xfer = [&](uint32_t peer, const float *snd, size_t ns, float *rcv, size_t nr) {
                if (r->comm.sendrecv(r->comm.ctx, peer, snd, ns * sizeof(float), rcv, nr * sizeof(float)) != 0)
                    throw Panic(FR_ERR_COMM, "transport callback failed");
            };
            uint32_t lo, hi;
            if (r->rank == 0) {
                for (uint32_t p = 1; p < r->world; ++p) {
                    r->rows_of(p, n_slots, lo, hi);
                    xfer(p, nullptr, 0, out + (size_t)lo * n_times, (size_t)(hi - lo) * n_times);
                }
            } else {
                r->my_rows(n_slots, lo, hi);
                xfer(0, out + (size_t)lo * n_times, (size_t)(hi - lo) * n_times, nullptr, 0);
            }
        }
    });
}

fr_status fr_host_register(fr_renderer *r, void *, size_t) { return r ? FR_OK : FR_ERR_INVALID_ARG; }   // (nothing to pin on the CPU)
fr_status fr_host_unregister(fr_renderer *r, void *) { return r ? FR_OK : FR_ERR_INVALID_ARG; }
fr_status fr_comm_selftest(int32_t, uint64_t) { return FR_ERR_UNSUPPORTED; }
fr_status fr_stream_begin(fr_renderer *, uint32_t) { return FR_ERR_UNSUPPORTED; }   // (a resident GPU launch has no CPU meaning)
fr_status fr_stream_block(fr_renderer *, float *, uint64_t, uint64_t, const float *, uint64_t) { return FR_ERR_UNSUPPORTED; }
fr_status fr_stream_end(fr_renderer *) { return FR_OK; }
fr_status fr_comm_unique_id(uint8_t *) { return FR_ERR_UNSUPPORTED; }   // (RCCL is the product's transport)

fr_status fr_set_shard(fr_renderer *r, const fr_shard *sh) {
    return guarded(r, [&] {
        r->rank = 0; r->world = 1; r->shard_mode = FR_SHARD_NONE; r->shard_flags = 0; r->has_comm = false;
        if (!sh || sh->world <= 1 || sh->mode == FR_SHARD_NONE) return;
        if ((sh->mode != FR_SHARD_VOICES && sh->mode != FR_SHARD_PARTIALS) || sh->world > 64 || sh->rank >= sh->world)
            throw Panic(FR_ERR_INVALID_ARG, "bad shard description");
        if (sh->mode == FR_SHARD_PARTIALS && (sh->world & (sh->world - 1))) throw Panic(FR_ERR_INVALID_ARG, "world must be a power of two");
        if (sh->rccl_id) throw Panic(FR_ERR_UNSUPPORTED, "the CPU oracle has no RCCL transport");
        r->rank = sh->rank; r->world = sh->world; r->shard_mode = sh->mode; r->shard_flags = sh->flags;
        if (sh->comm) { r->comm = *sh->comm; r->has_comm = true; }
    });
}

fr_status fr_shard_rows(const fr_renderer *r, uint32_t n_slots, uint32_t *lo, uint32_t *hi) {
    if (!r || !lo || !hi) return FR_ERR_INVALID_ARG;
    r->my_rows(n_slots, *lo, *hi);
    return FR_OK;
}

// Tracks (friendship_render.h): to the reference they are input rows like any other -- stored, readable by anything -- so
// the oracle takes the declaration and ignores it; the dense call is fill_buffer on the reference's own Array2 shape.
fr_status fr_set_track_inputs(fr_renderer *r, uint32_t) { return r ? FR_OK : FR_ERR_INVALID_ARG; }
fr_status fr_fill_buffer_dense(fr_renderer *r, float *out, uint32_t n_slots, uint64_t n_times, uint64_t idx, const float *in, uint32_t n_in_rows) {
    std::vector<uint64_t> offs((size_t)n_in_rows + 1);
    for (uint32_t i = 0; i <= n_in_rows; ++i) offs[i] = (uint64_t)i * n_times;
    return fr_fill_buffer(r, out, n_slots, n_times, idx, in, offs.data(), n_in_rows);
}
fr_status fr_fill_buffer_device_dense(fr_renderer *r, float *, uint32_t, uint64_t, uint64_t, const float *, uint32_t, void *) {
    if (r) r->last_error = "the CPU oracle has no device path";
    return FR_ERR_UNSUPPORTED;
}

fr_status fr_fill_buffer_device(fr_renderer *r, float *, uint32_t, uint64_t, uint64_t, const float *,
                                const uint64_t *, uint32_t, void *) {
    if (r) r->last_error = "the CPU oracle has no device path";
    return FR_ERR_UNSUPPORTED;
}

const char *fr_last_error(const fr_renderer *r) { return r ? r->last_error.c_str() : "null renderer"; }

const char *fr_status_string(fr_status s) {
    switch (s) {
    case FR_OK: return "ok";
    case FR_ERR_INVALID_ARG: return "invalid argument";
    case FR_ERR_INPUT_TOO_LONG: return "input row extends past the rendered range";
    case FR_ERR_INPUT_HISTORY: return "input row does not continue the slot's stored history";
    case FR_ERR_NO_SUCH_NODE: return "no such node";
    case FR_ERR_BAD_SLOT: return "primitive read through a non-zero output slot";
    case FR_ERR_CYCLE: return "dependency cycle";
    case FR_ERR_DEVICE: return "device error";
    case FR_ERR_NO_DEVICE: return "no usable device";
    case FR_ERR_OUT_OF_MEMORY: return "out of memory";
    case FR_ERR_UNSUPPORTED: return "unsupported";
    case FR_ERR_COMM: return "communication error";
    default: return "unknown status";
    }
}

const char *fr_backend_name(void) { return "cpu-oracle"; }
uint32_t fr_abi_version(void) { return FR_ABI_VERSION; }
const char *fr_plan_json(fr_renderer *) { return "{}"; }
fr_status fr_set_timing(fr_renderer *r, int32_t) { return r ? FR_OK : FR_ERR_INVALID_ARG; }
fr_status fr_get_timing(fr_renderer *r, const char *, double *ms, uint64_t *launches) {
    if (!r) return FR_ERR_INVALID_ARG;
    if (ms) *ms = 0.0;
    if (launches) *launches = 0;
    return FR_OK;
}
fr_status fr_reset_timing(fr_renderer *r) { return r ? FR_OK : FR_ERR_INVALID_ARG; }

// ---- oracle-only helpers (not in the product ABI) --------------------------------------------

// Threads used by fill_buffer's slot loop (1 = the reference's single thread).
fr_status fro_set_threads(fr_renderer *r, uint32_t n) {
    if (!r) return FR_ERR_INVALID_ARG;
    r->threads = n ? n : 1;
    return FR_OK;
}

// Random-access evaluation: out[i] = get_sample(times[i], slots[i]) against the stored input
// history (reference.rs:90-96).  The evaluator is a pure function of (graph, history, t), so this is
// how full-size configurations are spot-checked without rendering every frame on the CPU.
fr_status fro_eval_samples(fr_renderer *r, const uint32_t *slots, const uint64_t *times, size_t n, float *out) {
    return guarded(r, [&] {
        if (n && (!slots || !times || !out)) throw Panic(FR_ERR_INVALID_ARG, "null array");
        for (size_t i = 0; i < n; ++i) out[i] = r->get_sample(times[i], slots[i]);
    });
}

// Number of evaluator visits is not tracked (it would perturb the timed baseline).
uint64_t fro_head(const fr_renderer *r) { return r ? r->head : 0; }

}  // extern "C"
