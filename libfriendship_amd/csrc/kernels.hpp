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
    uint32_t sparkle;          // FR_SEMANTICS_SPARKLE: Minimum / Delay as SparkleRenderer computes them (kernels.hip prim_min)
    uint32_t *st_node;         // pull stack, [depth][count]
    uint64_t *st_time;
    float *st_val;
};
hipError_t launch_pull(const PullArgs &a, hipStream_t s);

// Fused oscillator bank (see kernels.hip): out[v][t] = balanced Sum2 tree over P partials of
// amp * parabolic_sine(Modulo(time[t] * w, 1)).
struct BankArgs {
    const float2 *params;      // [n_voices][P] {w, -4*amp}
    // Window frame ti (0 <= ti < n_times) reads the time-carrying input as
    //   ti < time_skip ? 0 : (ti - time_skip < time_valid ? time[ti - time_skip] : 0)
    const float *time;         // stored history of the time slot, starting at window frame time_skip (may be null)
    uint64_t time_skip;        // leading window frames that precede the stored history (zero after a seek)
    uint64_t time_valid;       // stored frames available from `time`; beyond -> 0
    // Voice v writes window frame ti to out[rows[v] * out_stride + (ring_mask ? (ring_t0 + ti) & ring_mask : ti)]
    float *out;
    const uint32_t *rows;      // [n_voices] destination row of each voice
    uint64_t out_stride;       // floats between destination rows
    uint64_t ring_mask;        // 0: rows are linear windows; else rows are rings of ring_mask + 1 floats
    uint64_t ring_t0;          // absolute frame of window frame 0 (ring addressing)
    uint32_t n_voices;
    uint32_t log2_p;           // P = 1 << log2_p
    uint64_t n_times;          // window length in frames
    uint32_t fast_ok;          // every w in [0, 2^32]: the non-negative fast path may be used
    uint32_t chunk_log2;       // partials per workgroup = 1 << chunk_log2 (5..13, <= log2_p); from bank_shape
    uint32_t frames_per_lane;  // 1, 2 or 4; from bank_shape
    uint32_t waves_per_group;  // 4 or 8; from bank_shape
    uint32_t small_call;       // from bank_shape.  1: lanes-over-partials kernel (calls of <= 2 frames).  2: the short-call
                               // kernel: too few (voice, tile) pairs to fill the chip, so every voice is split into chunks over
                               // several workgroups of 4..16 waves, parameters staged through LDS, chunk sums combined by the
                               // last workgroup to arrive (`tickets`) -- one launch
    uint32_t voices_per_wave;  // > 0: many small voices (<= 256 partials): a wave sums WHOLE voices, this many in a row, for one
                               // tile of frames -- no LDS, no barriers, the time values stay in registers (bank_multi_kernel)
    uint32_t leaf_variant;     // 0 = product-form leaves; 1 = FMA-form leaves + zero-sign repair (same bits, faster)
    float *hist_dst;           // if non-null: the kernel also copies time[0..time_valid) here (input-history append)
    float *ws;                 // [P >> chunk_log2][n_voices][n_times] partial sums; unused when one chunk
    // Row-completion flags for the host entry point (fr_fill_buffer streaming its output, engine.cpp): when host_flags is
    // set (time-major kernel, one chunk per voice only), the workgroup that finishes the LAST tile of voice v -- counted
    // in row_done[v], zero between launches -- stores flag_value to host_flags[rows[v]] (mapped host memory) after every
    // output store of the voice has been acknowledged: the host may then read that row while other voices still compute.
    uint32_t *row_done;
    uint32_t *host_flags;
    uint32_t flag_value;
    uint32_t *tickets;         // small_call == 2 with chunks: [n_voices][tiles] arrival counters, BANK_TICKET_STRIDE words
                               // apart, all zero between launches
    // general voices (launch_gbank): groups[i] = log2(item leaves, <= 11) | merges_after << 4; params = the items'
    // {w, -4*amp} pairs in order, items of < 8 leaves padded to 8 pairs; voice v owns items
    // [group_off[2v], group_off[2v+2]) and its parameters start at pair 8 * group_off[2v+1]
    const uint32_t *groups;
    const uint32_t *group_off;
};
constexpr uint32_t BANK_TICKET_STRIDE = 32;

// ---- block streaming (fr_stream_*, engine.cpp): ONE resident launch renders block after block -------------------------
// What a real-time host pays per 64-frame block through fr_fill_buffer is the launch path (~12 us from enqueue to first
// wave on this stack, profiles/r02_host_short_blocks.txt), not the 1.3 us of arithmetic.  Here the kernel is launched once
// and stays: the host rings a doorbell in mapped pinned memory, workgroup 0 sees it, copies the block's input row into
// device memory and releases the other workgroups; each renders its (voice, chunk) exactly as bank_short_kernel does, the
// chunk sums meet through the same tickets, finished rows go straight to mapped host memory and the workgroup that
// completes the last voice writes the block's sequence number back.  No workgroup ever waits for another one's result
// (tickets: the last arriver does the work), and every polling loop is bounded on the wall clock: without a doorbell for
// BANK_STREAM_IDLE_MS (the one number, quoted in friendship_render.h as FR_STREAM_IDLE_MS) the kernel ends itself, so a
// host that dies leaves no spinning GPU behind.
struct BankStreamCtl {        // mapped pinned host memory
    // host -> device.  The doorbell IS the block's input row: word i = row[i] (low half) | tag (high half), tag = the block's
    // sequence number << 8 | frames in the block (1..64; 0xFFFFFFFF = stop), written with one 8-byte store each.  Lane i of
    // workgroup 0 polls word i: a word whose tag is new carries its value with it (8-byte loads do not tear), so one trip
    // across PCIe brings both the news and the data (three dependent trips -- doorbell, length, row -- cost 5 us more).
    unsigned long long row[64];
    uint32_t done;            // device -> host: tag of the last finished block
    uint32_t alive;           // device -> host: 1 while the kernel is resident, 0 once it has ended
    uint32_t pad1[14];
};
struct BankStreamDev {        // device memory
    uint32_t seq;             // the doorbell, republished by workgroup 0 (0xFFFFFFFF = stop)
    uint32_t n_times;
    uint32_t voices_done;
    uint32_t pad[13];
    float row[64];
};
constexpr uint32_t BANK_STREAM_STOP = 0xFFFFFFFFu;
constexpr uint32_t BANK_STREAM_WGS = 256;      // most workgroups a stream may use (the caller also checks the device's CU count: all must be resident)
constexpr uint32_t BANK_STREAM_IDLE_MS = 2000; // the resident launch ends itself after this long without a block
// `a`: as for the short-call kernel (small_call == 2, 16 waves, chunk_log2 chosen so that voices * chunks <= the CUs of the device),
// out = device pointer of the mapped [n_voices][64] host result (out_stride = 64), ws / tickets allocated for 64 frames.
// Launches exactly n_voices * chunks workgroups.  idle_ms: 0 = BANK_STREAM_IDLE_MS.
hipError_t launch_bank_stream(const BankArgs &a, BankStreamCtl *ctl_dev, BankStreamDev *dev, uint32_t idle_ms, hipStream_t s);
// `many_pairs_whole`: do not cut a job of more than 320 (voice, tile) pairs into chunks (the short-call kernel's tickets and
// workspace are shared between calls, so such a launch cannot overlap with the next call on another stream).
void bank_shape(uint32_t log2_p, uint32_t n_voices, uint64_t n_times, uint32_t &chunk_log2, uint32_t &frames_per_lane,
                uint32_t &waves_per_group, uint32_t &small_call, uint32_t &voices_per_wave, bool many_pairs_whole = false);
uint64_t bank_blocks(const BankArgs &a);
bool bank_publishes_rows(const BankArgs &a);
hipError_t launch_bank(const BankArgs &a, hipStream_t s);
hipError_t launch_gbank(const BankArgs &a, hipStream_t s);   // voices that are arbitrary Sum2 trees (schedule form)

// ---- staged evaluator -------------------------------------------------------------------------------
// A cut node (a value some Delay reads back in time, or an output) is computed per frame by a small
// register program over: constants, inputs at t, and ring reads of other cut nodes at t - d.
enum StageOp : uint8_t {
    S_CONST = 0,       // dst = bits(imm)
    S_INPUT = 1,       // dst = input[imm] at t
    S_READ = 2,        // dst = t >= d ? ring[buf][(t - d) & mask] : 0          d = {lo, hi}
    S_READ_INPUT = 3,  // dst = t >= d ? input[imm] at t - d : 0
    S_STEP = 4,        // dst = t >= d ? bits(imm) : 0                           Delay of a constant
    S_SUM2 = 5, S_MUL = 6, S_DIV = 7, S_MOD = 8, S_MIN = 9,   // dst = a op b
    S_STORE = 10,      // ring[buf][t & mask] = a        (fused form: an inlined cut node still feeds its ring)
    // Delay with a signal amount (register a, evaluated at t): frames = amount -> u64 as reference.rs:200-211
    // (>= 2^64 -> output 0; negative/NaN -> 0 frames; else floor), then as the constant forms.  d_lo = proven bound.
    S_READ_DYN = 11,        // dst = ring[buf] at t - frames
    S_READ_INPUT_DYN = 12,  // dst = input[imm] at t - frames
    S_STEP_DYN = 13,        // dst = t >= frames ? bits(imm) : 0
};
struct StageInstr {    // 16 bytes
    uint8_t op, dst, a, b;
    uint32_t imm;
    uint32_t buf;      // ring index (S_READ)
    uint32_t d_lo;     // delay frames, low 32 bits (delays >= 2^32 frames are not staged)
};
struct StageProg {
    uint32_t first_instr, n_instr;
    uint32_t result_reg;
    uint32_t dst_ring;     // ring to store into, or 0xFFFFFFFF
    int32_t out_row;       // output row to store into (frames >= idx), or -1
    uint32_t n_loads;      // the first n_loads instructions are loads/constants with no register operands: the kernel
                           // issues them back to back so their memory latencies overlap (<= STAGE_MAX_HOISTED)
    uint32_t pad[2];
};
constexpr int STAGE_REGS = 48;
constexpr uint32_t STAGE_INLINE_INPUTS = 8;
constexpr uint32_t STAGE_MAX_HOISTED = 24;
struct StageArgs {
    const StageInstr *instrs;
    const StageProg *progs;    // programs of this level
    uint32_t n_progs;
    float *rings;              // [n_rings][ring_mask + 1]
    uint64_t ring_mask;
    const DevInput *inputs;    // [n_inputs] in device memory, used when n_inputs > STAGE_INLINE_INPUTS
    DevInput inline_inputs[8]; // the usual case: the table travels in the kernel arguments (no copy, no sync)
    uint32_t n_inputs;
    float *out;                // [n_slots, n_times] of the call
    uint64_t n_times;
    uint64_t idx;              // first frame of the call
    uint64_t w0;               // first frame of the window computed now (<= idx)
    uint64_t w_len;            // window length
    uint64_t stride;           // 0: thread wi computes frame w0 + wi; else the frames w0 + wi + k * stride inside the window, in order
                               // (StagedPlan::fused_stride: a long steady call of the fused form in ONE launch)
    uint32_t sparkle;          // FR_SEMANTICS_SPARKLE
    uint32_t carry_only;       // strided launches: every read of a ring that this launch stores comes from the carry (below), so
                               // a stride's stores need not be in memory before the next stride starts
    uint32_t use_carry;        // the programs carry annotations (feedback plans): the launch gets the carry's LDS
};
// Carry (feedback plans, stage.cpp): a strided thread that reads back, `stride` frames later, what it stored to a ring in its
// previous iteration need not go through memory for it -- a dependent L2 round trip per iteration, all there is to a one-sample
// loop.  S_STORE.imm = carry slot + 1 keeps the stored value (double-buffered by iteration parity, so that the order of stores
// and reads inside an iteration does not matter); S_READ.imm = carry slot + 1 (only where d_lo == the launch's stride and the
// ring is one the program stores) takes it from there, except in the thread's first iteration of the launch.  0 = no carry.
constexpr uint32_t STAGE_CARRY = 8;
hipError_t launch_stage(const StageArgs &a, hipStream_t s);

// One step of the partial-block exchange (friendship_render.h FR_SHARD_PARTIALS): row i, window frame t:
//   v = lo[i][t] + hi[i][t]         the Sum2 node one level up: left sub-tree + right sub-tree, one f32 add
// stored to dst_ws[i][t] (steps before the last; may alias lo or hi), or -- the last step, dst_ws == null -- where the
// unsharded plan puts the voice: dst[i] bit 31 set: ring dst[i] & 0x7FFFFFFF at absolute frame ring_t0 + t; else
// output row dst[i], frames t >= out_skip only (the look-back part of a window is not output).
struct ShardCombineArgs {
    const float *lo;
    const float *hi;
    float *dst_ws;
    const uint32_t *dst;
    float *out;
    uint64_t out_stride;
    uint64_t out_skip;
    float *rings;
    uint64_t ring_mask;
    uint64_t ring_t0;
    uint32_t n_rows;
    uint64_t len;
};
hipError_t launch_shard_combine(const ShardCombineArgs &a, hipStream_t s);

// A voice rendered as 2^log2_c consecutive pieces of its leaves (a short call of few voices would not fill the chip
// otherwise): ws holds the pieces' sums as rows [(voice << log2_c) | piece][n_times]; this adds them up the way the voice's
// Sum2 tree does above the pieces -- adjacent pairs, level by level -- and stores the voice at out[rows[voice]].  log2_c <= 6.
struct ChunkCombineArgs {
    const float *ws;
    float *out;
    const uint32_t *rows;
    uint64_t out_stride;
    uint64_t n_times;
    uint32_t n_voices;
    uint32_t log2_c;
};
hipError_t launch_chunk_combine(const ChunkCombineArgs &a, hipStream_t s);

// Fills dst[0..n) with *src_last (or 0 when src_last is null): last-value padding of a short input
// row (reference.rs:72-73) for the device-resident entry point.
hipError_t launch_pad(float *dst, uint64_t n, const float *src_last, hipStream_t s);

}  // namespace fr
