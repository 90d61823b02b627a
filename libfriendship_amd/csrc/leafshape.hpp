// leafshape.hpp -- the expression shape shared by the leaves of a shape-matched voice (host-only data).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace fr {

// The leaf of a shape-matched voice as the code generator needs it.
// A leaf op beside the FlatOps: an input row that differs from leaf to leaf (a control-rate TRACK: per-partial frequency /
// amplitude envelopes arriving as input rows, reference.rs:66-74,181-183).  a = the constants column that holds the slot
// NUMBER (as bits), so which row a leaf reads travels with its other per-leaf parameters.  Only slots declared as tracks
// (fr_set_track_inputs) are matched this way: they are read from the call's own dense input matrix, never stored.
constexpr uint32_t LEAF_TRACK = 9;

struct LeafShape {
    struct Op { uint32_t op, a, b; };   // op: FlatOp (OP_CONST: a = column, OP_INPUT: a = index into input_slots) or LEAF_TRACK
    std::vector<Op> ops;                // post-order; operands are indices into ops; the last op is the leaf value
    std::vector<uint32_t> input_slots;  // external input slots read at t
    uint32_t n_consts = 0;              // constants per leaf, in traversal order (columns)
    std::string key() const {           // canonical text of the shape (no constant values)
        std::string s = "in" + std::to_string(input_slots.size()) + ":";
        for (auto &o : ops) s += std::to_string(o.op) + "," + std::to_string(o.a) + "," + std::to_string(o.b) + ";";
        return s;
    }
};

}  // namespace fr
