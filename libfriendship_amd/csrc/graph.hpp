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

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/friendship_render.h"

namespace fr {

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

struct SubGraph;

struct MNode {
    int32_t kind = 0;                          // FR_PRIM_* or FR_EFFECT_GRAPH
    std::shared_ptr<const SubGraph> sub;       // composite definition (interned)
    std::vector<EdgeRef> inbound;              // by to_slot
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

    std::unordered_map<uint32_t, MNode> nodes;   // top level, mutable
    std::vector<EdgeRef> outputs;
    uint64_t version = 0;                        // bumped by every edit

private:
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
};

struct FlatNode {
    uint32_t op, a, b, depth;   // depth = longest path to a leaf (pull-stack sizing)
};

struct FlatGraph {
    std::vector<FlatNode> nodes;       // topological: operands precede users
    std::vector<uint32_t> outputs;     // per rendered slot, a node id
    uint32_t max_depth = 0;
    uint32_t max_input_slot = 0;       // highest OP_INPUT slot referenced (valid if has_input)
    bool has_input = false;
    uint64_t n_mirror_nodes_visited = 0;

    uint32_t konst(uint32_t bits);
    uint32_t input(uint32_t slot);
    uint32_t make(FlatOp op, uint32_t a, uint32_t b);   // hash-consing + constant folding
    bool is_const(uint32_t id) const { return nodes[id].op == OP_CONST; }
    bool is_const(uint32_t id, float v) const;
    float const_val(uint32_t id) const;

private:
    std::unordered_map<uint64_t, uint32_t> cse_[8];
    uint32_t push(FlatOp op, uint32_t a, uint32_t b, uint32_t depth);
};

// Lowers the first n_slots output slots of the mirror.  Throws fr::Error (NO_SUCH_NODE, BAD_SLOT,
// CYCLE) where the reference's evaluation of those slots would panic or never terminate.
FlatGraph lower(const Mirror &m, uint32_t n_slots);

// Exactly-rounded host evaluation of one primitive (same semantics as the device code and as
// reference.rs:197-262); used for constant folding.
float host_binop(FlatOp op, float a, float b);
float f32_from_bits(uint32_t b);
uint32_t f32_to_bits(float f);

}  // namespace fr
