// stage.hpp -- plan of the staged (materialised) evaluation: fused banks + per-cut-node register programs.
#pragma once

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
};

struct StagedPlan {
    std::vector<BankLaunch> banks;
    std::vector<StageInstr> instrs;
    std::vector<StageProg> progs;          // ordered by level
    std::vector<uint32_t> level_first;     // level l = progs[level_first[l] .. level_first[l+1])
    // fused steady-state form (optional): progs[fused_first .. fused_first + fused_count) do the work of all levels in
    // one launch; valid when the rings are current and the call is at most fused_max_frames long
    uint32_t fused_first = 0, fused_count = 0;
    uint64_t fused_max_frames = 0;
    uint32_t n_rings = 0;
    uint64_t lmax = 0;                     // deepest look-back any ring must serve
    std::vector<uint32_t> input_slots;     // dense input index used by programs -> external slot
    std::vector<uint32_t> pull_rows;       // output rows left to the pull interpreter
    bool uses_rings() const { return n_rings != 0; }
};

// allow_banks: recognise fused oscillator banks; allow_programs: stage everything else that qualifies.
// `reuse`: a matcher built over this same FlatGraph object with the same limits, kept by the caller across plans
// (incremental re-planning after a graph edit: unchanged voices are answered from its memo).
class BankMatcher;
StagedPlan plan_stages(const FlatGraph &g, bool allow_banks, bool allow_programs, uint32_t max_log2_p, bool allow_jit = false,
                       bool allow_template = true, BankMatcher *reuse = nullptr);

}  // namespace fr
