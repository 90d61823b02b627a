// match.hpp -- oscillator-bank recognition over a lowered graph (see match.cpp).
#pragma once

#include <cstdint>
#include <vector>

#include "graph.hpp"

namespace fr {

// Voices (output rows) that share a partial count and a time-carrying input slot: one fused launch.
struct BankGroup {
    uint32_t log2_p = 0;            // partials per voice = 1 << log2_p
    uint32_t input_slot = 0;        // external input slot read as `t`
    bool fast_ok = true;            // every w in [0, 2^32]
    std::vector<uint32_t> rows;     // output row of each voice
    std::vector<float> params;      // [rows][P]{w, -4*amp}
};

struct MatchResult {
    std::vector<BankGroup> banks;
    std::vector<uint32_t> other_rows;   // output rows no fused kernel covers
};

MatchResult match_banks(const FlatGraph &g, uint32_t max_log2_p);

}  // namespace fr
