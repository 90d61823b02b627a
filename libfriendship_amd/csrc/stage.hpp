// stage.hpp -- plan of the staged (materialised) evaluation: fused banks + per-cut-node register programs.
#pragma once

#include <algorithm>
#include <cstdint>
#include <memory>
#include <vector>

#include "graph.hpp"
#include "kernels.hpp"
#include "match.hpp"

namespace fr {

// Voices rendered by one bank launch.
struct BankLaunch {
    uint32_t log2_p = 0;
    uint32_t input_slot = 0;
    bool fast_ok = true;
    bool to_ring = false;            // rows are ring indices (window with look-back) instead of output rows
    bool to_ws = false;              // rows index the exchange workspace (partial-block sharding: this rank's sub-trees)
    std::vector<uint32_t> rows;      // destination row per voice
    std::vector<float> params;       // balanced: [voices][P]{w, -4*amp}; general: [groups][8]{w, -4*amp}
    bool general = false;            // voices are arbitrary Sum2 trees evaluated by schedule (match.hpp VoiceMatch)
    std::vector<uint32_t> groups;    // general: group words of all voices, concatenated
    std::vector<uint32_t> group_off; // general: [voices + 1][2] {first item, first parameter group of 8 pairs} of each voice
    uint32_t max_leaves = 0;
    // jit == true: leaves of an arbitrary common shape, kernel specialised with hipRTC (jit.hpp); params = [voices][P][k]
    bool jit = false;
    LeafShape shape;
    std::vector<bool> varying;
    std::vector<uint32_t> literal_bits;
    std::vector<uint32_t> alias;
    uint32_t k = 0;
    bool tracks = false;             // jit leaves read per-leaf track rows of the call's dense input matrix (leafshape.hpp LEAF_TRACK)
    uint32_t max_track_slot = 0;
};

// How the job is split over `world` renderers, one per GPU (friendship_render.h fr_shard).  Output rows are owned in
// contiguous blocks; every rank plans from the same graph with the same code, so what must agree between ranks -- the
// list of split voices and its order, the look-back window -- agrees by construction.
struct ShardSpec {
    uint32_t rank = 0, world = 1;
    int mode = FR_SHARD_NONE;
};
inline void shard_row_range(uint32_t rank, uint32_t world, uint32_t n_rows, uint32_t &lo, uint32_t &hi) {
    const uint32_t q = n_rows / world, r = n_rows % world;
    lo = rank * q + std::min(rank, r);
    hi = lo + q + (rank < r ? 1u : 0u);
}
inline uint32_t shard_row_owner(uint32_t row, uint32_t world, uint32_t n_rows) {
    const uint32_t q = n_rows / world, r = n_rows % world;
    const uint32_t big = r * (q + 1);   // rows held by the ranks that own q + 1 rows
    return row < big ? row / (q + 1) : r + (q ? (row - big) / q : 0);
}

// A bank voice cut at the top log2(world) levels of its Sum2 tree (FR_SHARD_PARTIALS): row i of the exchange
// workspace.  Every rank renders its own sub-tree of it; after the exchange the owner holds the voice's value and
// stores it where the unsharded plan would have: ring `dst` (to_ring) or output row `dst`.
struct SplitVoice {
    uint32_t owner = 0;
    bool to_ring = false;
    uint32_t dst = 0;
};

struct StagedPlan {
    std::vector<BankLaunch> banks;
    std::vector<SplitVoice> split;         // exchange workspace rows, in exchange order (same on every rank)
    std::vector<StageInstr> instrs;
    std::vector<StageProg> progs;          // ordered by level
    std::vector<uint32_t> level_first;     // level l = progs[level_first[l] .. level_first[l+1])
    // fused steady-state form (optional): progs[fused_first .. fused_first + fused_count) do the work of all levels in
    // one launch; valid when the rings are current and the call is at most fused_max_frames long
    uint32_t fused_first = 0, fused_count = 0;
    uint64_t fused_max_frames = 0;
    // gcd of the delays of the fused form's reads of program rings (0: there are none).  A call longer than
    // fused_max_frames is still ONE launch when its threads stride by this many frames (engine.cpp execute()).
    uint64_t fused_stride = 0;
    // Feedback plans (graph.hpp OP_FBREF): the fused form is the only valid one and always runs strided; its programs are
    // ordered in levels (progs[fused_first + fused_level_first[l] .. fused_first + fused_level_first[l + 1]), one launch each),
    // and progs[post_first .. post_first + post_count) copy rings to output rows after them.  Rings are brought up to date by
    // replaying the frames from 0 (there is no look-back window that bounds a loop).
    bool feedback = false;
    uint32_t feedback_loops = 0;     // cut loops the rendered rows reach
    std::vector<uint32_t> fused_level_first;
    uint32_t post_first = 0, post_count = 0;
    bool fused_carry_only = false;   // every fused program reads the rings it stores through the carry only (kernels.hpp STAGE_CARRY)
    uint32_t n_rings = 0;
    uint64_t lmax = 0;                     // deepest look-back any ring must serve
    // How far back in the INPUT history the staged part can read when it computes frame t (ring look-backs + the delays
    // of inputs on top of them: constant amounts and proven bounds); `input_lookback_unbounded`: some read has no bound
    // (an input delayed by an unbounded signal, or rows left to the pull interpreter).  fr_config.history_frames.
    uint64_t input_lookback = 0;
    bool input_lookback_unbounded = false;
    std::vector<uint32_t> input_slots;     // dense input index used by programs -> external slot
    std::vector<uint32_t> pull_rows;       // output rows left to the pull interpreter
    bool uses_rings() const { return n_rings != 0; }
};

// allow_banks: recognise fused oscillator banks; allow_programs: stage everything else that qualifies.
// `reuse`: a matcher built over this same FlatGraph object with the same limits, kept by the caller across plans
// (incremental re-planning after a graph edit: unchanged voices are answered from its memo).
class BankMatcher;
// `track_from`: input slots >= it are control-rate tracks (fr_set_track_inputs): visible only to the call that supplies them,
// readable only by the leaves of shape-matched voices (anything else that reads one makes the plan FR_ERR_UNSUPPORTED).
StagedPlan plan_stages(const FlatGraph &g, bool allow_banks, bool allow_programs, uint32_t max_log2_p, bool allow_jit = false,
                       bool allow_template = true, BankMatcher *reuse = nullptr, const ShardSpec *shard = nullptr, uint32_t track_from = 0xFFFFFFFFu);

}  // namespace fr
