// kernels.hpp -- launch interface between the host engine and the gfx950 kernels.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace fr {

// One external input slot's stored history (reference.rs:25 `inputs[slot]`), device resident.
// value(t) = t < base ? 0 : t < len ? data[t - base] : 0.   `base` is the zero prefix left by a seek.
struct DevInput {
    const float *data;
    uint64_t base;
    uint64_t len;
};

// Lowered node as the device sees it: {op, a, b, depth} (graph.hpp FlatNode), 16 bytes.
struct DevNode {
    uint32_t op, a, b, depth;
};

struct PullArgs {
    const DevNode *nodes;
    const uint32_t *outputs;   // [n_slots] root node per output slot
    const DevInput *inputs;    // [n_inputs]
    uint32_t n_inputs;
    float *out;                // [n_slots, n_times] row-major
    uint32_t n_slots;
    uint64_t n_times;
    uint64_t idx;
    uint64_t first;            // first linear (slot*n_times + t) element of this launch
    uint64_t count;            // elements in this launch
    uint32_t *st_node;         // pull stack, [depth][count]
    uint64_t *st_time;
    float *st_val;
};
hipError_t launch_pull(const PullArgs &a, hipStream_t s);

// Fused oscillator bank (see kernels.hip): out[v][t] = balanced Sum2 tree over P partials of
// amp * parabolic_sine(Modulo(time[t] * w, 1)).
struct BankArgs {
    const float2 *params;      // [n_voices][P] {w, -4*amp}
    const float *time;         // the time-carrying input row, already offset to frame idx (may be null)
    uint64_t time_valid;       // frames of `time` that are stored; beyond -> 0
    float *out;                // [n_voices rows of out], row stride n_times
    const uint32_t *rows;      // [n_voices] output row of each voice
    uint32_t n_voices;
    uint32_t log2_p;           // P = 1 << log2_p
    uint64_t n_times;
    uint32_t fast_ok;          // every w in [0, 2^32]: the non-negative fast path may be used
    uint32_t chunk_log2;       // partials per workgroup = 1 << chunk_log2 (5..13, <= log2_p); from bank_shape
    uint32_t frames_per_lane;  // 1, 2 or 4; from bank_shape
    uint32_t leaf_variant;     // 0 = product-form leaves; 1 = FMA-form leaves + zero-sign repair (same bits, faster)
    float *hist_dst;           // if non-null: the kernel also copies time[0..time_valid) here (input-history append)
    float *ws;                 // [P >> chunk_log2][n_voices][n_times] partial sums; unused when one chunk
};
void bank_shape(uint32_t log2_p, uint32_t n_voices, uint64_t n_times, uint32_t &chunk_log2, uint32_t &frames_per_lane);
uint64_t bank_blocks(const BankArgs &a);
hipError_t launch_bank(const BankArgs &a, hipStream_t s);

// Fills dst[0..n) with *src_last (or 0 when src_last is null): last-value padding of a short input
// row (reference.rs:72-73) for the device-resident entry point.
hipError_t launch_pad(float *dst, uint64_t n, const float *src_last, hipStream_t s);

}  // namespace fr
