// match.hpp -- oscillator-bank recognition over a lowered graph (see match.cpp).
#pragma once

#include <cstdint>
#include <deque>
#include <unordered_map>
#include <vector>

#include "graph.hpp"
#include "leafshape.hpp"

namespace fr {

// One recognised voice: a complete balanced Sum2 tree over 2^log2_p partial leaves.
// One recognised voice: a Sum2 tree over partial leaves.
//   balanced (general == false): complete tree over 2^log2_p leaves; params = [P]{w, -4*amp}.
//   general  (general == true):  any Sum2 tree (odd carries, unbalanced, non-power-of-two leaf counts), cut into
//     items = maximal complete sub-trees of 2^j <= 2048 consecutive leaves, evaluated in post-order with a stack:
//     groups[i] = j | merges_after << 4; params = the items' {w, -4*amp} pairs in order, an item of fewer than 8
//     leaves padded with zeros to 8 pairs (so every item starts on a group-of-8 boundary).
constexpr uint32_t GENERAL_MAX_ITEM_LOG2 = 11;   // kernels.hip bank_wave_sum: at most 2^8 groups of 8 leaves

struct VoiceMatch {
    uint32_t log2_p = 0;
    uint32_t input_slot = 0;        // external input slot read as `t`
    bool fast_ok = true;            // every w in [0, 2^32]
    std::vector<float> params;
    bool general = false;
    uint32_t n_leaves = 0;
    std::vector<uint32_t> groups;
    // shape-matched voice (jit == true): balanced tree over 2^log2_p leaves of one arbitrary expression shape;
    // params = [P][k] (the varying constant columns), input_slot unused (shape.input_slots instead)
    bool jit = false;
    LeafShape shape;
    std::vector<bool> varying;
    std::vector<uint32_t> literal_bits;
    std::vector<uint32_t> alias;      // alias[c] = first column whose values equal column c's in every leaf (c itself if none)
    uint32_t k = 0;
    bool tracks = false;              // some leaf input is a per-leaf track row (shape ops LEAF_TRACK)
    uint32_t max_track_slot = 0;      // highest input slot any leaf reads as a track
};

class BankMatcher {
public:
    // track_from: input slots >= it are control-rate tracks (leafshape.hpp LEAF_TRACK; fr_set_track_inputs)
    BankMatcher(const FlatGraph &g, uint32_t max_log2_p, bool allow_jit = false, bool allow_template = true, uint32_t track_from = 0xFFFFFFFFu);
    ~BankMatcher();
    BankMatcher(const BankMatcher &) = delete;
    BankMatcher &operator=(const BankMatcher &) = delete;
    // Is the expression rooted at `root` a voice?  Results (including failures) are memoised per node.
    bool try_voice(uint32_t root, VoiceMatch &out);
    // The same without the copy: the memoised match, or null.  The pointer stays valid until retain_used() (matches live in
    // a deque: later matches do not move earlier ones), so a plan may hold it while it is being made.
    const VoiceMatch *match(uint32_t root);
    // A matcher kept across plans of the same (append-only, hash-consed) FlatGraph answers repeated roots from its
    // memo.  begin_plan() .. retain_used() bracket one plan: entries no plan has asked for since are dropped once
    // they outnumber the live ones.
    void begin_plan();
    void retain_used();
    size_t cached_voices() const { return found_.size(); }

private:
    struct Impl;
    Impl *impl_;
    const FlatGraph &g_;
    uint32_t max_log2_p_;
    bool allow_jit_;
    bool allow_template_;   // false (FR_BANK_TEMPLATE=0, A/B runs only): skip the hand-matched partial template
    std::unordered_map<uint32_t, int64_t> memo_;   // root -> index into found_, or -1
    std::deque<VoiceMatch> found_;
    std::unordered_map<uint32_t, bool> used_;      // roots asked for since begin_plan()
};

}  // namespace fr
