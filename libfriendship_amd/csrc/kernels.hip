// kernels.hip -- gfx950 (CDNA4) kernels of the render engine.  Written for 64-lane wavefronts,
// 256 CUs x 4 SIMD-32; compiled with -ffp-contract=off: every f32 operation of the reference
// evaluator (reference src/render/reference.rs:197-262) rounds exactly once and so must we.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "kernels.hpp"

#pragma clang fp contract(off)

namespace fr {

// Diagnostic build only (tools/fewvoice_diag.hip defines FR_DIAG_STAMPS): per-wave timestamps (s_memrealtime at 100 MHz,
// s_memtime on the shader clock) and placement (HW_ID, XCC_ID), written to a buffer the tool hangs on g_diag.  In the
// product no stamp executes and none of this exists.
#ifdef FR_DIAG_STAMPS
__device__ unsigned long long *g_diag = nullptr;     // [workgroup][16 waves][8]: start, compute done, end, HW_ID | XCC_ID << 32, then the
                                                     // same three moments on the shader clock in slots 4..6
#define FR_DIAG_MARK(slot)                                                                                                 \
    do {                                                                                                                   \
        if (g_diag && (threadIdx.x & 63u) == 0u) {                                                                         \
            unsigned long long v_ = __builtin_amdgcn_s_memrealtime();                                                      \
            if ((slot) == 3) v_ = (unsigned long long)__builtin_amdgcn_s_getreg(0xF804) | ((unsigned long long)(__builtin_amdgcn_s_getreg(0xF814) & 15u) << 32); \
            g_diag[((size_t)blockIdx.x * 16u + (threadIdx.x >> 6)) * 8u + (slot)] = v_;                                    \
            if ((slot) != 3) g_diag[((size_t)blockIdx.x * 16u + (threadIdx.x >> 6)) * 8u + 4u + (slot)] = __builtin_amdgcn_s_memtime(); \
        }                                                                                                                  \
    } while (0)
#else
#define FR_DIAG_MARK(slot) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------
// Shared primitive arithmetic (bit-exact restatement of reference.rs:197-262)
// ---------------------------------------------------------------------------------------------------
enum : uint32_t { OP_CONST = 0, OP_INPUT = 1, OP_DELAY = 2, OP_SUM2 = 3, OP_MUL = 4, OP_DIV = 5, OP_MOD = 6, OP_MIN = 7 };

__device__ __forceinline__ float prim_mod(float a, float b) {
    // `dividend % divisor` is fmodf (exact, sign of the dividend); `if rem < 0 { rem + divisor }`
    float rem = fmodf(a, b);
    return rem < 0.0f ? rem + b : rem;
}

__device__ __forceinline__ float prim_min(float a, float b, bool sparkle = false) {
    // RefRenderer: Rust >= 1.20 core f32::min: (a < b || b.is_nan()) ? a : b.
    // SparkleRenderer (sparkle.rs:495-496): select(fcmp ult a, b, a, b) -- differs only for a NaN `a` beside a number.
    // The value selected for `b` passes through an empty asm: where the compiler can prove an operand is not NaN (a literal,
    // or a 0.0 it has threaded in from an out-of-range read) the AMDGPU backend rewrites `x < c ? x : c` as v_min_f32, which
    // answers -0 for the tie (-0, +0) where this rule answers `b` (found by tools/stress_parity.py in generated code with a
    // literal zero operand; profiles/r02_stress_parity.txt).  Arms that are not the compare's operands cannot be matched.
    if (sparkle && a != a) return a;
    const bool take_a = a < b || b != b;
    float other = b;
    asm("" : "+v"(other));
    return take_a ? a : other;
}

__device__ __forceinline__ float prim_binop(uint32_t op, float a, float b, bool sparkle = false) {
    switch (op) {
    case OP_SUM2: return a + b;
    case OP_MUL: return a * b;
    case OP_DIV: return a / b;   // hipcc default: correctly rounded f32 divide
    case OP_MOD: return prim_mod(a, b);
    default: return prim_min(a, b, sparkle);
    }
}

// Delay's amount -> frames (reference.rs:200-211).  Returns false when the output is 0: the amount is >= 2^64, or --
// SparkleRenderer only, sparkle.rs:531-534 -- it is negative or NaN (`ult 0`), where RefRenderer delays by 0 frames.
__device__ __forceinline__ bool delay_frames(float d, uint64_t &frames, bool sparkle = false) {
    if (d >= 18446744073709551616.0f) return false;
    if (sparkle && !(d >= 0.0f)) return false;
    frames = (d < 0.0f || d != d) ? 0ull : (uint64_t)d;   // clamp negatives, NaN -> 0, floor
    return true;
}

__device__ __forceinline__ float read_input(const DevInput *in, uint32_t n_in, uint32_t slot, uint64_t t) {
    if (slot >= n_in) return 0.0f;
    DevInput s = in[slot];
    if (t < s.base || t >= s.len) return 0.0f;   // zero prefix after a seek; beyond stored -> 0 (reference.rs:92-94)
    return s.data[t - s.base];
}

// ---------------------------------------------------------------------------------------------------
// Pull interpreter: one thread per (output row, t), the reference's recursion with an explicit stack.
// Universal (any DAG of the 7 primitives, signal-dependent delays) and bit-exact; cost grows with the
// number of root-to-leaf paths exactly like the reference, so the planner uses it only for graphs
// it cannot stage.  Stack frames live in global memory, [level][thread] so a wave's accesses coalesce.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pull_kernel(PullArgs a) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.count) return;
    const uint64_t lin = a.first + e;
    const uint32_t slot = (uint32_t)(lin / a.n_times);
    const uint64_t ti = lin - (uint64_t)slot * a.n_times;

    uint32_t sp = 0;
    uint32_t cur = a.outputs[slot];
    uint64_t cur_t = a.idx + ti;
    float ret = 0.0f;
    bool calling = true;
    for (;;) {
        if (calling) {
            DevNode n = a.nodes[cur];
            if (n.op == OP_CONST) {
                ret = __uint_as_float(n.a);
                calling = false;
            } else if (n.op == OP_INPUT) {
                ret = read_input(a.inputs, a.n_inputs, n.a, cur_t);
                calling = false;
            } else {
                // frame {node, stage 0, t}; first operand: Delay evaluates its amount first (reference.rs:200)
                uint64_t o = (uint64_t)sp * a.count + e;
                a.st_node[o] = cur;
                a.st_time[o] = cur_t;
                ++sp;
                cur = (n.op == OP_DELAY) ? n.b : n.a;
            }
        } else {
            if (sp == 0) break;
            uint64_t o = (uint64_t)(sp - 1) * a.count + e;
            uint32_t ns = a.st_node[o];
            uint32_t node = ns & 0x3FFFFFFFu;
            DevNode n = a.nodes[node];
            uint64_t t = a.st_time[o];
            if ((ns >> 30) == 0) {
                if (n.op == OP_DELAY) {
                    uint64_t frames;
                    --sp;   // tail call: the source's value is the Delay's value
                    if (!delay_frames(ret, frames, a.sparkle != 0u) || frames > t) {
                        ret = 0.0f;   // >= 2^64 or t - frames underflows (reference.rs:202-205,213)
                    } else {
                        cur = n.a;
                        cur_t = t - frames;
                        calling = true;
                    }
                } else {
                    a.st_val[o] = ret;
                    a.st_node[o] = node | (1u << 30);
                    cur = n.b;
                    cur_t = t;
                    calling = true;
                }
            } else {
                ret = prim_binop(n.op, a.st_val[o], ret, a.sparkle != 0u);
                --sp;
            }
        }
    }
    a.out[lin] = ret;
}

hipError_t launch_pull(const PullArgs &a, hipStream_t s) {
    if (a.count == 0) return hipSuccess;
    uint64_t blocks = (a.count + 255) / 256;
    hipLaunchKernelGGL(pull_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Fused oscillator bank.
//
// For each voice v (an output row) and frame t:
//     out[v][t] = TREE_{k<P}  amp_k * parab(Modulo(time[t] * w_k, 1))
// where TREE is the balanced binary Sum2 tree over consecutive leaves and parab is the parabolic sine
//     u = phase + (-0.5);  y = (-16*u) * (0.5 + -(-(min(u, -u))))          (11 primitive nodes per partial)
// Algebra that is exact in f32 (proved in DESIGN.md, checked bit-for-bit by tests/test_bank_parity.py):
//     -(-min(u,-u)) == |u| up to the sign of zero, which 0.5 + (-x) discards;
//     (-16*u)*q == -16*RN(u*q) because |u*q| is 0 or >= 2^-50 (no subnormal, no overflow);
//     amp*(-16*z) == RN((-16*amp)*z) when -16*amp is exact (host checks) -> A = -16*amp.
// So a leaf costs 6 VALU ops (mul, fract, sub, sub|abs|, mul, mul) -- and 5 with two FMAs, see bank_leaf --
// and the tree 1 add: 6 VALU ops per partial-frame on the common path.
//
// Mapping (time-major lanes): a lane owns F frames, a wave owns 64*F consecutive frames and a
// contiguous quarter of the voice's partials, a 256-thread workgroup (4 waves) owns one
// (voice, time-tile).  Partial parameters {w, A} are wave-uniform, so they arrive through the scalar
// cache into SGPRs (s_load_dwordx16 = 8 partials) and feed the VALU as scalar operands: no LDS
// traffic, no cross-lane reduction in the hot loop.  The Sum2 tree is evaluated in its own
// association: 8 leaves in registers, then a binary-counter carry chain of named registers across
// groups (wave-uniform branches), then the 4 waves' subtrees combine through LDS.
// ---------------------------------------------------------------------------------------------------

// EXACT = false is the 5-op leaf: with u2 = 2u = RN(2r - 1) (scaling by 2 commutes with rounding) and
// q = 0.5 - |u| always exact (r is a multiple of 2^-25 wherever |u| < 0.25, Sterbenz elsewhere),
//     u*q = RN(u*(0.5 - |u|)) = (1/4) * RN(u2 - u2*|u2|) = (1/4) * fma(-|u2|, u2, u2)
// so leaf = A4 * fma(-|u2|, u2, u2) with A4 = -4*amp: mul, fract, fma, fma, mul.  It differs from the
// graph's value only in the SIGN OF A ZERO leaf (r = 0: the fma's exact cancellation is +0, the
// graph's product u*(+0) is -0).  A zero's sign cannot change a non-zero sum, so a wave's subtree sum
// is exact whenever it is non-zero; when it is zero the wave recomputes its subtree with EXACT = true,
// the 6-op product form, which is bit-identical to the graph including zero signs
// (both claims brute-forced over 1.5e8 (t, w, amp) triples; parity-tested on device).
template <bool FAST, bool EXACT>
__device__ __forceinline__ float bank_leaf(float t, float w, float A4) {
    float x = t * w;
    float r;
    if (FAST) {
        r = __builtin_amdgcn_fractf(x);   // x >= 0 finite: x - floor(x) == fmodf(x, 1) exactly
    } else {
        r = x - truncf(x);                // == fmodf(x, 1) for every finite x; inf -> NaN like fmodf
        r = r < 0.0f ? r + 1.0f : r;      // `if rem < 0 { rem + divisor }`
    }
    if (EXACT) {
        float u = r - 0.5f;
        float q = 0.5f - fabsf(u);
        float z = u * q;
        return (4.0f * A4) * z;           // -16*amp, exact scaling
    }
    float u2 = __builtin_fmaf(r, 2.0f, -1.0f);
    float z4 = __builtin_fmaf(-fabsf(u2), u2, u2);
    return A4 * z4;
}

typedef float __attribute__((address_space(4))) const *const_f32_ptr;   // constant addrspace -> SMEM loads

struct ParamGroup {   // 8 partials' {w, A}: one s_load_dwordx16, lives in SGPRs
    float w[8], A[8];
};

__device__ __forceinline__ void load_group(ParamGroup &pg, const_f32_ptr p, uint32_t g) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        pg.w[j] = p[(size_t)g * 16 + 2 * j];
        pg.A[j] = p[(size_t)g * 16 + 2 * j + 1];
    }
}

// A wave sums at most 2^8 groups = 2048 partials (chunk <= 8192 per workgroup): 9 carry levels.
// The levels are separate named arrays, not one 2-D array: hipcc otherwise merges the per-level
// stores into one dynamically indexed store and the whole table drops to scratch memory.
template <int I, int N, class Fn>
__device__ __forceinline__ void static_for(Fn &&fn) {   // compile-time indices: nothing left to index dynamically
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        static_for<I + 1, N>(fn);
    }
}

template <int F>
using Lvl = float (&)[F];
#define FR_LEVELS_DECL Lvl<F> s0, Lvl<F> s1, Lvl<F> s2, Lvl<F> s3, Lvl<F> s4, Lvl<F> s5, Lvl<F> s6, Lvl<F> s7, Lvl<F> s8
#define FR_LEVELS_PASS s0, s1, s2, s3, s4, s5, s6, s7, s8

template <int F, bool FAST, bool EXACT>
__device__ __forceinline__ void bank_group(const ParamGroup &pg, uint32_t g,
                                           const float (&t)[F], FR_LEVELS_DECL) {
    float v[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        float l0 = bank_leaf<FAST, EXACT>(t[f], pg.w[0], pg.A[0]);
        float l1 = bank_leaf<FAST, EXACT>(t[f], pg.w[1], pg.A[1]);
        float l2 = bank_leaf<FAST, EXACT>(t[f], pg.w[2], pg.A[2]);
        float l3 = bank_leaf<FAST, EXACT>(t[f], pg.w[3], pg.A[3]);
        float l4 = bank_leaf<FAST, EXACT>(t[f], pg.w[4], pg.A[4]);
        float l5 = bank_leaf<FAST, EXACT>(t[f], pg.w[5], pg.A[5]);
        float l6 = bank_leaf<FAST, EXACT>(t[f], pg.w[6], pg.A[6]);
        float l7 = bank_leaf<FAST, EXACT>(t[f], pg.w[7], pg.A[7]);
        v[f] = ((l0 + l1) + (l2 + l3)) + ((l4 + l5) + (l6 + l7));
    }
    // binary-counter carry: level k holds the finished left sibling of height k (wave-uniform branches);
    // the chain always stops at level `levels` at the latest because g < ngroups = 2^levels
#define FR_CARRY(K, SK)                                                    \
    if (((g >> K) & 1u) == 0u) {   /* g < 2^levels: bit `levels` is 0 */    \
        static_for<0, F>([&](auto f) { SK[f] = v[f]; });                   \
        return;                                                            \
    }                                                                      \
    static_for<0, F>([&](auto f) { v[f] = SK[f] + v[f]; });
    FR_CARRY(0u, s0) FR_CARRY(1u, s1) FR_CARRY(2u, s2) FR_CARRY(3u, s3) FR_CARRY(4u, s4)
    FR_CARRY(5u, s5) FR_CARRY(6u, s6) FR_CARRY(7u, s7)
#undef FR_CARRY
#pragma unroll
    for (int f = 0; f < F; ++f) s8[f] = v[f];
}

// (Fetching a PAIR of parameter groups ahead instead of one -- for launches that leave a SIMD only a few waves to hide a
//  scalar load behind -- was built and measured SLOWER everywhere: 64 x 4096, T = 64 7.5 -> 8.4 us, 256 10.6 -> 13.3,
//  512 17.8 -> 20.1, 1024 29.3 -> 33.0; the 32 extra SGPRs take these kernels to the register limit.
//  profiles/r02_short_calls.txt.)
// `first`: group 0 already requested by the caller (short calls ask for it before they wait for their time row).
template <int F, bool FAST, bool EXACT>
__device__ __forceinline__ void bank_wave_sum(const float *params, uint32_t ngroups, uint32_t levels,
                                              const float (&t)[F], float (&res)[F], const ParamGroup *first = nullptr) {
    float s0[F], s1[F], s2[F], s3[F], s4[F], s5[F], s6[F], s7[F], s8[F];
#pragma unroll
    for (int f = 0; f < F; ++f)
        s0[f] = s1[f] = s2[f] = s3[f] = s4[f] = s5[f] = s6[f] = s7[f] = s8[f] = 0.0f;

    const_f32_ptr p = (const_f32_ptr)params;
    // two parameter groups in flight: the scalar load of group g+1 is issued before group g's math
    ParamGroup pa, pb;
    if (first) pa = *first;
    else load_group(pa, p, 0);
    // (scalar loads return out of order, so the only usable wait is lgkmcnt(0): wait for the group
    // about to be consumed FIRST, then issue the next load so it flies under this group's ~120 VALU ops)
    for (uint32_t g = 0; g < ngroups; g += 2) {
        const bool has_b = g + 1 < ngroups;
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): pa has landed
        if (has_b) load_group(pb, p, g + 1);
        bank_group<F, FAST, EXACT>(pa, g, t, FR_LEVELS_PASS);
        if (has_b) {
            __builtin_amdgcn_s_waitcnt(0xC07F);   // pb has landed
            if (g + 2 < ngroups) load_group(pa, p, g + 2);
            bank_group<F, FAST, EXACT>(pb, g + 1, t, FR_LEVELS_PASS);
        }
    }
    // after the last group (all ones) the carry chain stopped at level `levels`
#pragma unroll
    for (int f = 0; f < F; ++f) {
        float r = s8[f];
        r = levels == 7u ? s7[f] : r;
        r = levels == 6u ? s6[f] : r;
        r = levels == 5u ? s5[f] : r;
        r = levels == 4u ? s4[f] : r;
        r = levels == 3u ? s3[f] : r;
        r = levels == 2u ? s2[f] : r;
        r = levels == 1u ? s1[f] : r;
        r = levels == 0u ? s0[f] : r;
        res[f] = r;
    }
}

// The sign of a zero sum.  An RN sum tree yields -0 iff every leaf is -0 (x + (-x) and (+0) + (-0) are +0), so
// when a workgroup's result holds a zero only that AND over its leaves is needed, never a recomputation.
// The AND is order-free, so here lanes run over PARTIALS (coalesced float2 loads), unlike the hot loop.
// Returns, for this thread's share of partials [0, n), whether all leaves at time t are exactly -0.0 in the
// graph's arithmetic (product-form leaf, general fract: bit-identical to the graph for every input).
__device__ __forceinline__ bool leaves_all_negzero(const float2 *params, uint32_t n, float t, uint32_t tid, uint32_t nthreads) {
    bool ok = true;
    for (uint32_t k = tid; k < n && ok; k += nthreads) {
        float2 p = params[k];
        ok = __float_as_uint(bank_leaf<false, true>(t, p.x, p.y)) == 0x80000000u;
    }
    return ok;
}

// The same question for all 64 frames of a tile at once, for tiles whose results are mostly zeros (a silent voice:
// every amplitude 0; partials beyond 2^23 cycles): lanes over frames and parameters through the scalar cache like
// the hot loop, each wave over its own share of the partials.  Returns per lane whether every leaf of the share is
// exactly -0.0 (AND and OR of the bit patterns both equal to the sign bit).  Not inlined: its registers must not
// disturb the hot loop's allocation (an inlined second loop made the kernel 1.6x slower, profiles/r01_bank_variants.txt).
template <bool FAST>
__device__ __attribute__((noinline)) bool wave_leaves_all_negzero(const float *params, uint32_t ngroups, float t, unsigned long long lanes) {
    // `lanes`: the frames of the tile whose answer is wanted (those whose sum came out zero).  A lane's answer is
    // settled as soon as it meets one leaf that is not -0, so the scan stops once that has happened in every wanted
    // lane -- for a silent voice (amplitudes 0: leaves are +-0 with t-dependent signs; or every t*w beyond 2^23: all
    // leaves +0) that is after the first pair of groups, and the voice costs one pass like a sounding one, not two.
    const_f32_ptr p = (const_f32_ptr)params;
    const bool wanted = (lanes >> (threadIdx.x & 63u)) & 1ull;
    uint32_t all_and = 0xFFFFFFFFu, all_or = 0u;
    ParamGroup pa, pb;
    load_group(pa, p, 0);
    auto fold = [&](const ParamGroup &pg) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint32_t b = __float_as_uint(bank_leaf<FAST, true>(t, pg.w[j], pg.A[j]));
            all_and &= b;
            all_or |= b;
        }
    };
    for (uint32_t g = 0; g < ngroups; g += 2) {
        const bool has_b = g + 1 < ngroups;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        if (has_b) load_group(pb, p, g + 1);
        fold(pa);
        if (has_b) {
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (g + 2 < ngroups) load_group(pa, p, g + 2);
            fold(pb);
        }
        if (__ballot(wanted && all_and == 0x80000000u && all_or == 0x80000000u) == 0ull) {
            __builtin_amdgcn_s_waitcnt(0xC07F);   // (a prefetched group may still be landing in pa)
            return false;
        }
    }
    return all_and == 0x80000000u && all_or == 0x80000000u;
}

__device__ __forceinline__ float bank_time(const BankArgs &a, uint64_t ti) {
    return (ti >= a.time_skip && ti - a.time_skip < a.time_valid) ? a.time[ti - a.time_skip] : 0.0f;
}
__device__ __forceinline__ uint64_t bank_out_index(const BankArgs &a, uint64_t ti) {
    return a.ring_mask ? ((a.ring_t0 + ti) & a.ring_mask) : ti;
}

// MODE 0: every leaf in the product form (always exact).
// MODE 1: FMA-form leaves (5 ops + 1 add per partial-frame); where the workgroup's combined result is a
//         zero, its sign is settled by leaves_all_negzero (rare: t*w integral for every partial, e.g. t = 0).
// MODE 2: FMA-form leaves without the sign repair -- diagnostic only (tools/bank_bench.hip).
// (Recomputing flagged tiles with the product-form loop inside the same kernel was tried first: with a second
//  copy of the hot loop inlined, register allocation degrades and the kernel runs 1.6x slower;
//  profiles/r01_bank_variants.txt.)
// NW = waves per workgroup (4 or 8): a wave is the unit of work that cannot be split further, so more, smaller
// waves shorten the kernel's tail at a given call size (measured: profiles/r01_bank_variants.txt).
// FLAGS: publish row-completion flags to the host (BankArgs::host_flags) -- a separate instantiation, so that the few
// scalar registers it needs never touch the occupancy of the ordinary launches.
template <int F, int MODE, int NW, bool FLAGS = false>
__global__ void __launch_bounds__(64 * NW) bank_kernel(BankArgs a, uint32_t tiles, uint32_t nblocks) {
    // XCD-aware order: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a
    // contiguous range of (voice, chunk, tile) work: its L2 then sees 1/8 of the parameter streams.
    uint32_t b = blockIdx.x;
    uint32_t lid = (nblocks % 8u == 0u) ? (b % 8u) * (nblocks / 8u) + b / 8u : b;
    const uint32_t nchunks = 1u << (a.log2_p - a.chunk_log2);
    const uint32_t vc = lid / tiles;                  // voice * nchunks + chunk
    const uint32_t tile = lid - vc * tiles;
    const uint32_t voice = vc >> (a.log2_p - a.chunk_log2);
    const uint32_t chunk = vc & (nchunks - 1u);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    FR_DIAG_MARK(0);
    FR_DIAG_MARK(3);
    const uint64_t t0 = (uint64_t)tile * (64u * F);
    float t[F];
    bool nonneg = true;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        uint64_t ti = t0 + (uint32_t)f * 64u + lane;
        t[f] = bank_time(a, ti);
        nonneg = nonneg && (t[f] >= 0.0f) && (t[f] <= 4294967296.0f);
        // the call's time row doubles as this slot's input history (reference.rs:70): one workgroup
        // column appends it, which saves a separate copy kernel on the stream (only with time_skip == 0)
        if (a.hist_dst && vc == 0u && wave == 0u && ti < a.time_valid) a.hist_dst[ti] = t[f];
    }
    const bool fast = a.fast_ok && __all(nonneg);   // then every x = t*w is in [0, 2^64]: finite, >= 0

    const uint32_t Pc = 1u << a.chunk_log2;           // partials per workgroup, 8*NW <= Pc <= 2048*NW
    const uint32_t Pw = Pc / NW;                      // partials per wave
    const float2 *cparams = a.params + ((size_t)voice << a.log2_p) + (size_t)chunk * Pc;
    const float *params = (const float *)(cparams + (size_t)wave * Pw);
    const uint32_t ngroups = Pw >> 3;
    const uint32_t levels = a.chunk_log2 - 3u - (NW == 8 ? 3u : NW == 4 ? 2u : NW == 2 ? 1u : 0u);   // log2(ngroups) <= 8

    float res[F];
    constexpr bool EXACT = (MODE == 0);
    if (fast) bank_wave_sum<F, true, EXACT>(params, ngroups, levels, t, res);
    else bank_wave_sum<F, false, EXACT>(params, ngroups, levels, t, res);
    FR_DIAG_MARK(1);

    __shared__ float sm[NW][F][64];
    __shared__ unsigned long long zmask[F];
#pragma unroll
    for (int f = 0; f < F; ++f) sm[wave][f][lane] = res[f];
    __syncthreads();
    FR_DIAG_MARK(2);
    // one chunk: straight to the voice's output row; else to the workspace [chunk][voice][t]
    const bool direct = nchunks == 1u;
    float *orow = direct ? a.out + (size_t)a.rows[voice] * a.out_stride
                         : a.ws + ((size_t)chunk * a.n_voices + voice) * a.n_times;
    if (wave == 0) {
#pragma unroll
        for (int f = 0; f < F; ++f) {
            uint64_t ti = t0 + (uint32_t)f * 64u + lane;
            float r = sm[0][f][lane];
            if (NW >= 2) r = r + sm[1 % NW][f][lane];
            if (NW >= 4) r = r + (sm[2 % NW][f][lane] + sm[3 % NW][f][lane]);
            if (NW == 8) r = r + ((sm[4 % NW][f][lane] + sm[5 % NW][f][lane]) + (sm[6 % NW][f][lane] + sm[7 % NW][f][lane]));
            bool live = ti < a.n_times;
            // Streaming store: nothing re-reads it here, and when the row lives in page-locked HOST memory (fr_host_register)
            // the wave is released ~9 us/launch sooner.  FLAGS: the row is read by the host BEFORE the launch ends, as soon as
            // its flag arrives, so the store must be a system-scope one whose acknowledgement means "visible to the host"
            // (a streaming store is acknowledged earlier: the flag overtook the data, tools/host_stream_soak.py).
            if (live) {
                float *dst = &orow[direct ? bank_out_index(a, ti) : ti];
                if (FLAGS) __hip_atomic_store(dst, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                else __builtin_nontemporal_store(r, dst);
            }
            if (MODE == 1) {
                unsigned long long m = __ballot(live && r == 0.0f);
                if (lane == 0) zmask[f] = m;
            }
        }
    }
    if (MODE == 1) {
        __syncthreads();
#pragma unroll
        for (int f = 0; f < F; ++f) {
            unsigned long long m = zmask[f];          // workgroup-uniform
            if (__builtin_popcountll(m) > 4) {        // many zeros in this tile: settle all 64 frames in one pass
                bool mine = fast ? wave_leaves_all_negzero<true>(params, ngroups, t[f], m)
                                 : wave_leaves_all_negzero<false>(params, ngroups, t[f], m);
                sm[wave][f][lane] = mine ? 1.0f : 0.0f;   // (the sums in sm were consumed before the barrier above)
                __syncthreads();
                if (wave == 0 && ((m >> lane) & 1ull)) {
                    bool all = true;
#pragma unroll
                    for (int w = 0; w < NW; ++w) all = all && sm[w][f][lane] != 0.0f;
                    uint64_t ti = t0 + (uint32_t)f * 64u + lane;
                    float *dst = &orow[direct ? bank_out_index(a, ti) : ti];
                    if (FLAGS) __hip_atomic_store(dst, all ? -0.0f : 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    else *dst = all ? -0.0f : 0.0f;
                }
                __syncthreads();
                continue;
            }
            while (m) {
                uint32_t l = (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                uint64_t ti = t0 + (uint32_t)f * 64u + l;
                float tz = bank_time(a, ti);
                int all = __syncthreads_and(leaves_all_negzero(cparams, Pc, tz, threadIdx.x, 64u * NW) ? 1 : 0);
                if (threadIdx.x == 0) {
                    float *dst = &orow[direct ? bank_out_index(a, ti) : ti];
                    if (FLAGS) __hip_atomic_store(dst, all ? -0.0f : 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    else *dst = all ? -0.0f : 0.0f;
                }
            }
        }
    }
    // Row-completion flag (the host entry point copies finished rows while the launch is still running).  Every output
    // store of this workgroup was issued by wave 0; once they are acknowledged one lane counts the tile in, and the
    // workgroup that brings a voice's count to `tiles` publishes the row's flag to the host (system-scope release).
    if (FLAGS && a.host_flags && direct && wave == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0u) {
            const uint32_t old = __hip_atomic_fetch_add(a.row_done + voice, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == tiles - 1u) {
                __hip_atomic_store(a.row_done + voice, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
                __hip_atomic_store(a.host_flags + a.rows[voice], a.flag_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// Many small voices (<= 256 partials each): one workgroup per (batch of voices, tile of frames); every wave sums WHOLE
// voices, `voices_per_wave` of them one after the other, with the same inner loop.  Nothing is shared between waves:
// no LDS, no barriers; the tile's time values are loaded once and stay in registers for all the wave's voices.  (With
// a quarter of a 32-partial voice per wave the fixed cost of a workgroup -- time loads, one exposed parameter load,
// LDS hand-over, barrier -- outweighs its 8 leaves: 2.1 T partial-frames/s.)
template <int F, int MODE>
__global__ void __launch_bounds__(256) bank_multi_kernel(BankArgs a, uint32_t tiles, uint32_t nblocks) {
    uint32_t b = blockIdx.x;
    uint32_t lid = (nblocks % 8u == 0u) ? (b % 8u) * (nblocks / 8u) + b / 8u : b;
    const uint32_t vb = lid / tiles, tile = lid - vb * tiles;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t t0 = (uint64_t)tile * (64u * F);
    float t[F];
    bool nonneg = true;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        uint64_t ti = t0 + (uint32_t)f * 64u + lane;
        t[f] = bank_time(a, ti);
        nonneg = nonneg && (t[f] >= 0.0f) && (t[f] <= 4294967296.0f);
        if (a.hist_dst && vb == 0u && wave == 0u && ti < a.time_valid) a.hist_dst[ti] = t[f];
    }
    const bool fast = a.fast_ok && __all(nonneg);
    const uint32_t ngroups = 1u << (a.log2_p - 3u), levels = a.log2_p - 3u;
    const uint32_t v0 = (vb * 4u + wave) * a.voices_per_wave;
    constexpr bool EXACT = (MODE == 0);
    for (uint32_t j = 0; j < a.voices_per_wave; ++j) {
        const uint32_t voice = v0 + j;
        if (voice >= a.n_voices) break;                      // wave-uniform
        const float2 *vparams = a.params + ((size_t)voice << a.log2_p);
        float res[F];
        if (fast) bank_wave_sum<F, true, EXACT>((const float *)vparams, ngroups, levels, t, res);
        else bank_wave_sum<F, false, EXACT>((const float *)vparams, ngroups, levels, t, res);
        float *orow = a.out + (size_t)a.rows[voice] * a.out_stride;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const uint64_t ti = t0 + (uint32_t)f * 64u + lane;
            const bool live = ti < a.n_times;
            if (live) orow[bank_out_index(a, ti)] = res[f];
            if (MODE == 1) {   // sign of exact zeros, as in bank_kernel, within the wave
                unsigned long long zm = __ballot(live && res[f] == 0.0f);
                if (zm == 0ull) continue;
                if (__builtin_popcountll(zm) > 4) {
                    const bool all = fast ? wave_leaves_all_negzero<true>((const float *)vparams, ngroups, t[f], zm)
                                          : wave_leaves_all_negzero<false>((const float *)vparams, ngroups, t[f], zm);
                    if ((zm >> lane) & 1ull) orow[bank_out_index(a, ti)] = all ? -0.0f : 0.0f;
                    continue;
                }
                while (zm) {
                    const uint32_t l = (uint32_t)__builtin_ctzll(zm);
                    zm &= zm - 1;
                    const uint64_t tz_i = t0 + (uint32_t)f * 64u + l;
                    const float tz = bank_time(a, tz_i);
                    const bool all = __all(leaves_all_negzero(vparams, 1u << a.log2_p, tz, lane, 64u));
                    if (lane == 0u) orow[bank_out_index(a, tz_i)] = all ? -0.0f : 0.0f;
                }
            }
        }
    }
}

// Short calls (fewer frames than half a wave has lanes): time-major lanes would idle, so here lanes run over
// PARTIALS -- the layout the north star sketches: coalesced float2 loads of each partial's {w, A4} (held in
// registers for the call's frames), a wavefront-shuffle butterfly for the per-voice mix, LDS for the 4 waves.
// The butterfly IS the graph's balanced tree: at step m lane i adds the sub-tree sum of lanes i^m, and f32 add is
// bitwise commutative, so after 6 steps every lane holds the tree-ordered sum of its wave's 64 consecutive
// partials.  A workgroup covers 256 consecutive partials of one voice; chunks combine with bank_combine_kernel.
constexpr uint32_t SMALL_MAX_FRAMES = 32;   // capacity of the kernel; bank_shape uses it for T <= 2 only (measured)

__global__ void __launch_bounds__(256) bank_small_kernel(BankArgs a) {
    const uint32_t chunk = blockIdx.x, voice = blockIdx.y;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t nchunks = 1u << (a.log2_p - 8u);
    const float2 prm = a.params[((size_t)voice << a.log2_p) + (size_t)chunk * 256u + threadIdx.x];
    __shared__ float sm[4][SMALL_MAX_FRAMES];
    __shared__ uint32_t zmask;
    for (uint32_t ti = 0; ti < (uint32_t)a.n_times; ++ti) {
        const float t = bank_time(a, ti);                                  // wave-uniform
        const bool fast = a.fast_ok && t >= 0.0f && t <= 4294967296.0f;
        float v = fast ? bank_leaf<true, false>(t, prm.x, prm.y) : bank_leaf<false, false>(t, prm.x, prm.y);
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) v = v + __shfl_xor(v, m, 64);
        if (lane == 0) sm[wave][ti] = v;
    }
    if (threadIdx.x == 0) zmask = 0u;
    __syncthreads();
    const bool direct = nchunks == 1u;
    float *orow = direct ? a.out + (size_t)a.rows[voice] * a.out_stride
                         : a.ws + ((size_t)chunk * a.n_voices + voice) * a.n_times;
    if (threadIdx.x < a.n_times) {
        const uint32_t ti = threadIdx.x;
        float r = (sm[0][ti] + sm[1][ti]) + (sm[2][ti] + sm[3][ti]);
        orow[direct ? bank_out_index(a, ti) : ti] = r;
        if (r == 0.0f) atomicOr(&zmask, 1u << ti);
    }
    __syncthreads();
    uint32_t zm = zmask;                                                   // workgroup-uniform
    while (zm) {   // the sign of a zero: -0 iff each of this workgroup's 256 leaves is -0 in the graph's arithmetic
        uint32_t ti = (uint32_t)__builtin_ctz(zm);
        zm &= zm - 1;
        float tz = bank_time(a, ti);
        bool neg = __float_as_uint(bank_leaf<false, true>(tz, prm.x, prm.y)) == 0x80000000u;
        int all = __syncthreads_and(neg ? 1 : 0);
        if (threadIdx.x == 0) orow[direct ? bank_out_index(a, ti) : ti] = all ? -0.0f : 0.0f;
    }
}

// Upper levels of the voice's Sum2 tree when a voice was split over several workgroups:
// out[v][t] = TREE_c ws[c][v][t], same binary-counter association, one thread per (v, t).
__global__ void __launch_bounds__(256) bank_combine_kernel(BankArgs a) {
    const uint64_t total = (uint64_t)a.n_voices * a.n_times;
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const uint32_t voice = (uint32_t)(e / a.n_times);
    const uint64_t ti = e - (uint64_t)voice * a.n_times;
    const uint32_t levels = a.log2_p - a.chunk_log2;
    const uint32_t nchunks = 1u << levels;
    constexpr int MAXL = 20;
    float s[MAXL];
#pragma unroll
    for (int k = 0; k < MAXL; ++k) s[k] = 0.0f;
    float result = 0.0f;
    for (uint32_t c = 0; c < nchunks; ++c) {
        float v = a.ws[e + (uint64_t)c * total];
#pragma unroll
        for (int k = 0; k < MAXL; ++k) {
            if ((uint32_t)k == levels) { result = v; break; }
            if (((c >> k) & 1u) == 0u) { s[k] = v; break; }
            v = s[k] + v;
        }
    }
    a.out[(size_t)a.rows[voice] * a.out_stride + bank_out_index(a, ti)] = result;
}


// ---------------------------------------------------------------------------------------------------
// Short calls.  With few (voice, tile) pairs -- a 64-frame block of config C is 64 of them -- the time-major kernel
// above leaves most of the chip idle and each wave walks its 512 partials as a chain of scalar loads: one
// s_load_dwordx16 per 8 partials, ~150 ns each with nothing else on the SIMD to hide it (measured: 11 us per call for
// any call of 8..256 frames, tools/short_call_probe.py).  Here instead
//   * a voice is cut into `nchunks` chunks, one workgroup of NW waves each, so that hundreds of workgroups exist and a
//     wave's chain of parameter loads is a handful of groups long, with several waves per SIMD to overlap them;
//     (staging the chunk's parameters in LDS and reading them back as broadcast ds_read_b128 was built first: a
//     wave-uniform value broadcast to 64 lanes costs the LDS the full 1 KiB of return bandwidth per instruction, and the
//     kernel ran 2-6x SLOWER than the scalar-load form, profiles/r02_short_calls.txt);
//   * lanes still run over frames and a wave still sums its partials in the graph's own association (same bank_group
//     carry chain), the NW wave sums meet in LDS, and the chunk sums meet in HBM: every workgroup stores its 64 chunk
//     sums write-through (sc1), waits for them, and takes a ticket (agent-scope atomic add); the workgroup whose add
//     comes last reads all chunks back (sc1 loads) and adds them in tree order.  One launch, no grid barrier
//     (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 stores -> vmcnt(0) -> one lane's atomic add -> the last
//     adder loads sc1).
// ---------------------------------------------------------------------------------------------------
// one ticket per 128-byte line: neighbouring (voice, tile) pairs would otherwise serialise their atomics on one L2 line
constexpr uint32_t TICKET_STRIDE = BANK_TICKET_STRIDE;

template <int NW>
__global__ void __launch_bounds__(64 * NW) bank_short_kernel(BankArgs a, uint32_t tiles) {
    __shared__ float sm[NW][64];
    __shared__ unsigned long long zshared;
    const uint32_t clog = a.log2_p - a.chunk_log2, nchunks = 1u << clog;
    // (an XCD-aware order like bank_kernel's -- a contiguous (voice, tile, chunk) range per XCD -- measured the same within
    //  noise: 8 x 4096 x 4800 21.7 -> 21.3 us, 16 x 4096 36.8 -> 36.9; profiles/r03_fewvoices.txt)
    const uint32_t chunk = blockIdx.x & (nchunks - 1u);   // the chunks of one (voice, tile) are neighbours in the grid
    const uint32_t vt = blockIdx.x >> clog;
    const uint32_t voice = vt / tiles, tile = vt - voice * tiles;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t ti = (uint64_t)tile * 64u + lane;
    const bool live = ti < a.n_times;
    const uint32_t Pc = 1u << a.chunk_log2;
    const uint32_t Pw = Pc / NW, ngroups = Pw >> 3;
    uint32_t levels = 0;
    while ((1u << levels) < ngroups) ++levels;
    const float *mine = (const float *)(a.params + ((size_t)voice << a.log2_p) + (size_t)chunk * Pc + (size_t)wave * Pw);
    FR_DIAG_MARK(0);
    FR_DIAG_MARK(3);
    ParamGroup first;                                      // requested BEFORE the time row: the two trips to memory overlap
    load_group(first, (const_f32_ptr)mine, 0);
    const float t = bank_time(a, ti);
    if (a.hist_dst && voice == 0u && chunk == 0u && wave == 0u && ti < a.time_valid) a.hist_dst[ti] = t;   // (every tile of voice 0)
    const bool fast = a.fast_ok && __all(t >= 0.0f && t <= 4294967296.0f);   // the same 64 frames in every wave
    const float tt[1] = {t};
    float r_wave[1];
    if (fast) bank_wave_sum<1, true, false>(mine, ngroups, levels, tt, r_wave, &first);
    else bank_wave_sum<1, false, false>(mine, ngroups, levels, tt, r_wave, &first);
    FR_DIAG_MARK(1);
    sm[wave][lane] = r_wave[0];
    __syncthreads();
    FR_DIAG_MARK(2);
    float r = 0.0f;
    if (wave == 0u) {   // the NW wave sums in tree order: adjacent pairs, level by level
        float s[NW];
        static_for<0, NW>([&](auto w) { s[w] = sm[w][lane]; });
        static_for<0, NW / 2>([&](auto i) { s[i] = s[2 * i] + s[2 * i + 1]; });
        if constexpr (NW >= 4) static_for<0, NW / 4>([&](auto i) { s[i] = s[2 * i] + s[2 * i + 1]; });
        if constexpr (NW >= 8) static_for<0, NW / 8>([&](auto i) { s[i] = s[2 * i] + s[2 * i + 1]; });
        if constexpr (NW >= 16) static_for<0, NW / 16>([&](auto i) { s[i] = s[2 * i] + s[2 * i + 1]; });
        r = s[0];
        const unsigned long long z = __ballot(live && r == 0.0f);
        if (lane == 0u) zshared = z;
    }
    __syncthreads();
    const unsigned long long zm = zshared;                // workgroup-uniform
    if (zm != 0ull) {   // the sign of a zero chunk sum: -0 iff every leaf of the chunk is -0 in the graph's arithmetic
        const bool ok = fast ? wave_leaves_all_negzero<true>(mine, ngroups, t, zm) : wave_leaves_all_negzero<false>(mine, ngroups, t, zm);
        sm[wave][lane] = ok ? 1.0f : 0.0f;
        __syncthreads();
        if (wave == 0u && ((zm >> lane) & 1ull)) {
            bool all = true;
            static_for<0, NW>([&](auto w) { all = all && sm[w][lane] != 0.0f; });
            r = all ? -0.0f : 0.0f;
        }
    }
    if (wave != 0u) return;
    float *orow = a.out + (size_t)a.rows[voice] * a.out_stride;
    if (nchunks == 1u) {
        if (live) orow[bank_out_index(a, ti)] = r;
        return;
    }
    // publish this chunk's sums, take a ticket; the last workgroup of the (voice, tile) to arrive adds the chunks up
    const size_t vstride = (size_t)a.n_voices * a.n_times;
    float *slot = a.ws + (size_t)voice * a.n_times;
    if (live) __hip_atomic_store(slot + (size_t)chunk * vstride + ti, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t old = 0u;
    if (lane == 0u) old = __hip_atomic_fetch_add(a.tickets + (size_t)vt * TICKET_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != nchunks - 1u) return;
    if (live) {
        // binary-counter carry over the chunks (nchunks = 2^clog <= 256): level k holds the finished left sibling of
        // height k; named registers, so nothing is indexed dynamically (no scratch)
        float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f, c4 = 0.0f, c5 = 0.0f, c6 = 0.0f, c7 = 0.0f, c8 = 0.0f;
        for (uint32_t c = 0; c < nchunks; ++c) {
            float v = c == chunk ? r : __hip_atomic_load(slot + (size_t)c * vstride + ti, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            do {
                if (!(c & 1u)) { c0 = v; break; } v = c0 + v;
                if (!(c & 2u)) { c1 = v; break; } v = c1 + v;
                if (!(c & 4u)) { c2 = v; break; } v = c2 + v;
                if (!(c & 8u)) { c3 = v; break; } v = c3 + v;
                if (!(c & 16u)) { c4 = v; break; } v = c4 + v;
                if (!(c & 32u)) { c5 = v; break; } v = c5 + v;
                if (!(c & 64u)) { c6 = v; break; } v = c6 + v;
                if (!(c & 128u)) { c7 = v; break; } v = c7 + v;
                c8 = v;
            } while (0);
        }
        float result = c8;   // after the last chunk (all ones) the chain stopped at level clog
        result = clog == 7u ? c7 : result; result = clog == 6u ? c6 : result; result = clog == 5u ? c5 : result;
        result = clog == 4u ? c4 : result; result = clog == 3u ? c3 : result; result = clog == 2u ? c2 : result;
        result = clog == 1u ? c1 : result;
        orow[bank_out_index(a, ti)] = result;
    }
    if (lane == 0u) __hip_atomic_store(a.tickets + (size_t)vt * TICKET_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
}

template <int NW>
static hipError_t launch_bank_short(const BankArgs &a, hipStream_t s) {
    const uint64_t tiles = (a.n_times + 63) / 64;
    const uint64_t nb = (tiles * a.n_voices) << (a.log2_p - a.chunk_log2);
    if (nb == 0) return hipSuccess;
    if (nb > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL((bank_short_kernel<NW>), dim3((uint32_t)nb), dim3(64 * NW), 0, s, a, (uint32_t)tiles);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Block streaming: the short-call kernel as ONE resident launch (kernels.hpp, BankStreamCtl).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) bank_stream_kernel(BankArgs a, BankStreamCtl *ctl, BankStreamDev *dev, uint32_t idle_ms) {
    constexpr int NW = 16;
    __shared__ float sm[NW][64];
    __shared__ unsigned long long zshared;
    __shared__ uint32_t s_seq, s_T;
    const uint32_t clog = a.log2_p - a.chunk_log2, nchunks = 1u << clog;
    const uint32_t chunk = blockIdx.x & (nchunks - 1u);
    const uint32_t voice = blockIdx.x >> clog;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t Pc = 1u << a.chunk_log2;
    const uint32_t Pw = Pc / NW, ngroups = Pw >> 3;
    uint32_t levels = 0;
    while ((1u << levels) < ngroups) ++levels;
    const bool working = voice < a.n_voices;                 // (the grid is exactly n_voices * nchunks workgroups: always true)
    // Every polling loop below is bounded on the 100 MHz wall clock (s_memrealtime): with no block for `idle_ms` the launch
    // ends itself (friendship_render.h: FR_STREAM_IDLE_MS), whatever a look across PCIe happens to cost.
    const unsigned long long idle_ticks = (unsigned long long)idle_ms * 100000ull;
    const float *mine = (const float *)(a.params + ((size_t)(working ? voice : 0u) << a.log2_p) + (size_t)chunk * Pc + (size_t)wave * Pw);
    const size_t vstride = (size_t)a.n_voices * 64u;
    uint32_t seen = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&ctl->alive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    for (;;) {
        // ---- wait for the next block (bounded) ----
        if (blockIdx.x == 0) {
            if (wave == 0u) {
                uint32_t tag = seen;
                float v = 0.0f;
                bool fresh = false;
                const unsigned long long wait_from = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    const unsigned long long word = __hip_atomic_load(&ctl->row[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    tag = (uint32_t)(word >> 32);
                    v = __uint_as_float((uint32_t)word);
                    fresh = __all(tag != seen) && (uint32_t)__builtin_amdgcn_readfirstlane(tag) == tag;   // every lane holds the same new tag
                    fresh = __all(fresh);
                    if (fresh || __builtin_amdgcn_s_memrealtime() - wait_from > idle_ticks) break;
                }
                const uint32_t seq = fresh ? __builtin_amdgcn_readfirstlane(tag) : BANK_STREAM_STOP;   // nobody rang: end
                uint32_t T = 0;
                if (seq != BANK_STREAM_STOP) {
                    T = seq & 0xFFu;
                    T = T > 64u ? 64u : T;
                    __hip_atomic_store(&dev->row[lane], lane < T ? v : 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (lane == 0u) {
                    __hip_atomic_store(&dev->n_times, T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(&dev->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (row and n_times are acknowledged)
                    s_seq = seq;
                    s_T = T;
                }
            }
        } else if (threadIdx.x == 0) {
            uint32_t seq = seen;
            const unsigned long long wait_from = __builtin_amdgcn_s_memrealtime();
            for (;;) {                                                       // (outlasts workgroup 0's bound, which ends in a STOP for everybody)
                seq = __hip_atomic_load(&dev->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (seq != seen || __builtin_amdgcn_s_memrealtime() - wait_from > 2ull * idle_ticks + 10000000ull) break;
                __builtin_amdgcn_s_sleep(4);
            }
            if (seq == seen) seq = BANK_STREAM_STOP;
            s_seq = seq;
            s_T = __hip_atomic_load(&dev->n_times, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const uint32_t seq = s_seq, T = s_T;
        if (seq == BANK_STREAM_STOP) break;
        seen = seq;
        if (working) {
            // ---- one (voice, chunk) of one tile, as bank_short_kernel ----
            const bool live = lane < T;
            ParamGroup first;
            load_group(first, (const_f32_ptr)mine, 0);
            const float t = live ? __hip_atomic_load(&dev->row[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
            const bool fast = a.fast_ok && __all(t >= 0.0f && t <= 4294967296.0f);
            const float tt[1] = {t};
            float r_wave[1];
            if (fast) bank_wave_sum<1, true, false>(mine, ngroups, levels, tt, r_wave, &first);
            else bank_wave_sum<1, false, false>(mine, ngroups, levels, tt, r_wave, &first);
            sm[wave][lane] = r_wave[0];
            __syncthreads();
            float r = 0.0f;
            if (wave == 0u) {
                float s[NW];
                static_for<0, NW>([&](auto w) { s[w] = sm[w][lane]; });
                static_for<0, NW / 2>([&](auto i) { s[i] = s[2 * i] + s[2 * i + 1]; });
                static_for<0, NW / 4>([&](auto i) { s[i] = s[2 * i] + s[2 * i + 1]; });
                static_for<0, NW / 8>([&](auto i) { s[i] = s[2 * i] + s[2 * i + 1]; });
                static_for<0, NW / 16>([&](auto i) { s[i] = s[2 * i] + s[2 * i + 1]; });
                r = s[0];
                const unsigned long long z = __ballot(live && r == 0.0f);
                if (lane == 0u) zshared = z;
            }
            __syncthreads();
            const unsigned long long zm = zshared;
            if (zm != 0ull) {
                const bool ok = fast ? wave_leaves_all_negzero<true>(mine, ngroups, t, zm) : wave_leaves_all_negzero<false>(mine, ngroups, t, zm);
                sm[wave][lane] = ok ? 1.0f : 0.0f;
                __syncthreads();
                if (wave == 0u && ((zm >> lane) & 1ull)) {
                    bool all = true;
                    static_for<0, NW>([&](auto w) { all = all && sm[w][lane] != 0.0f; });
                    r = all ? -0.0f : 0.0f;
                }
            }
            if (wave == 0u) {
                float *orow = a.out + (size_t)a.rows[voice] * 64u;
                bool finished_voice = nchunks == 1u;
                float result = r;
                if (nchunks > 1u) {
                    float *slot = a.ws + (size_t)voice * 64u;
                    __hip_atomic_store(slot + (size_t)chunk * vstride + lane, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    uint32_t old = 0u;
                    if (lane == 0u) old = __hip_atomic_fetch_add(a.tickets + (size_t)voice * TICKET_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    old = __builtin_amdgcn_readfirstlane(old);
                    if (old == nchunks - 1u) {
                        finished_voice = true;
                        float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f, c4 = 0.0f, c5 = 0.0f, c6 = 0.0f, c7 = 0.0f, c8 = 0.0f;
                        for (uint32_t c = 0; c < nchunks; ++c) {
                            float v = c == chunk ? r : __hip_atomic_load(slot + (size_t)c * vstride + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            do {
                                if (!(c & 1u)) { c0 = v; break; } v = c0 + v;
                                if (!(c & 2u)) { c1 = v; break; } v = c1 + v;
                                if (!(c & 4u)) { c2 = v; break; } v = c2 + v;
                                if (!(c & 8u)) { c3 = v; break; } v = c3 + v;
                                if (!(c & 16u)) { c4 = v; break; } v = c4 + v;
                                if (!(c & 32u)) { c5 = v; break; } v = c5 + v;
                                if (!(c & 64u)) { c6 = v; break; } v = c6 + v;
                                if (!(c & 128u)) { c7 = v; break; } v = c7 + v;
                                c8 = v;
                            } while (0);
                        }
                        result = c8;
                        result = clog == 7u ? c7 : result; result = clog == 6u ? c6 : result; result = clog == 5u ? c5 : result;
                        result = clog == 4u ? c4 : result; result = clog == 3u ? c3 : result; result = clog == 2u ? c2 : result;
                        result = clog == 1u ? c1 : result;
                        if (lane == 0u) __hip_atomic_store(a.tickets + (size_t)voice * TICKET_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (finished_voice) {
                    // the row goes to the host: system-scope stores, acknowledged before the voice is counted in
                    if (live) __hip_atomic_store(&orow[lane], result, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0u) {
                        const uint32_t n = __hip_atomic_fetch_add(&dev->voices_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (n == a.n_voices - 1u) {
                            __hip_atomic_store(&dev->voices_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(&ctl->done, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (every row's stores were acknowledged before its voice was counted)
                        }
                    }
                }
            }
        }
        __syncthreads();   // LDS is reused by the next block
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&ctl->alive, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t launch_bank_stream(const BankArgs &a, BankStreamCtl *ctl_dev, BankStreamDev *dev, uint32_t idle_ms, hipStream_t s) {
    if (a.chunk_log2 < 7 || a.chunk_log2 > 13 || a.chunk_log2 > a.log2_p || a.log2_p - a.chunk_log2 > 8) return hipErrorInvalidValue;
    if (((uint64_t)a.n_voices << (a.log2_p - a.chunk_log2)) > BANK_STREAM_WGS || a.n_voices == 0) return hipErrorInvalidValue;
    if ((1u << a.chunk_log2) / 16u < 8u) return hipErrorInvalidValue;          // a wave needs a whole group of 8 partials
    if (a.chunk_log2 != a.log2_p && (!a.ws || !a.tickets)) return hipErrorInvalidValue;
    if (a.leaf_variant != 1 || !a.out || !a.rows || !ctl_dev || !dev) return hipErrorInvalidValue;
    // exactly the workgroups that render: a small patch leaves the other CUs to whatever else wants the device
    const uint32_t wgs = a.n_voices << (a.log2_p - a.chunk_log2);
    hipLaunchKernelGGL(bank_stream_kernel, dim3(wgs), dim3(1024), 0, s, a, ctl_dev, dev, idle_ms ? idle_ms : BANK_STREAM_IDLE_MS);
    return hipGetLastError();
}

// Will launch_bank publish row-completion flags for these arguments (host_flags set)?  Only the time-major kernel with
// one chunk per voice and the FMA-form leaves does.
bool bank_publishes_rows(const BankArgs &a) {
    return a.host_flags && !a.small_call && !a.voices_per_wave && a.leaf_variant == 1 && a.chunk_log2 == a.log2_p &&
           (a.frames_per_lane == 1 || a.frames_per_lane == 2 || a.frames_per_lane == 4);
}

// Workgroups launch_bank uses for this shape.
uint64_t bank_blocks(const BankArgs &a) {
    uint64_t f = a.frames_per_lane;
    return ((a.n_times + 64 * f - 1) / (64 * f)) * a.n_voices << (a.log2_p - a.chunk_log2);
}

template <int F>
static hipError_t launch_bank_f(const BankArgs &a, hipStream_t s) {
    uint64_t nblocks64 = bank_blocks(a);
    if (nblocks64 == 0) return hipSuccess;
    if (nblocks64 > 0x7FFFFFFFull) return hipErrorInvalidValue;
    uint32_t tiles = (uint32_t)((a.n_times + 64 * F - 1) / (64 * F)), nblocks = (uint32_t)nblocks64;
    const bool w8 = a.waves_per_group == 8;
    if (a.leaf_variant == 0) {
        if (w8) hipLaunchKernelGGL((bank_kernel<F, 0, 8>), dim3(nblocks), dim3(512), 0, s, a, tiles, nblocks);
        else hipLaunchKernelGGL((bank_kernel<F, 0, 4>), dim3(nblocks), dim3(256), 0, s, a, tiles, nblocks);
    } else if (a.leaf_variant == 1 && a.host_flags && a.chunk_log2 == a.log2_p) {   // (bank_publishes_rows)
        if (w8) hipLaunchKernelGGL((bank_kernel<F, 1, 8, true>), dim3(nblocks), dim3(512), 0, s, a, tiles, nblocks);
        else hipLaunchKernelGGL((bank_kernel<F, 1, 4, true>), dim3(nblocks), dim3(256), 0, s, a, tiles, nblocks);
    } else if (a.leaf_variant == 1) {
        if (w8) hipLaunchKernelGGL((bank_kernel<F, 1, 8>), dim3(nblocks), dim3(512), 0, s, a, tiles, nblocks);
        else if (a.waves_per_group == 2 && a.chunk_log2 <= 12) hipLaunchKernelGGL((bank_kernel<F, 1, 2>), dim3(nblocks), dim3(128), 0, s, a, tiles, nblocks);
        else if (a.waves_per_group == 1 && a.chunk_log2 <= 11) hipLaunchKernelGGL((bank_kernel<F, 1, 1>), dim3(nblocks), dim3(64), 0, s, a, tiles, nblocks);
        else hipLaunchKernelGGL((bank_kernel<F, 1, 4>), dim3(nblocks), dim3(256), 0, s, a, tiles, nblocks);
    } else {
        hipLaunchKernelGGL((bank_kernel<F, 2, 4>), dim3(nblocks), dim3(256), 0, s, a, tiles, nblocks);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || a.chunk_log2 == a.log2_p) return e;
    uint64_t total = (uint64_t)a.n_voices * a.n_times;
    hipLaunchKernelGGL(bank_combine_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// Chooses the launch shape for this call.  Measured on MI355X at 64 voices x 4096 partials
// (tools/bank_bench.hip, profiles/r01_bank_variants.txt, profiles/r01_bank_small_calls.txt):
//  * one 64-frame tile per wave (F = 1) is never slower than 2 or 4;
//  * long calls (>= 512 workgroups): 4 waves x 1024 partials per workgroup; 8 waves measured equal;
//  * short calls: ONE workgroup of 8 waves per (voice, tile) beats splitting voices into chunks + a combine
//    pass (T = 32: 13.9 us vs 31 us) and beats the lanes-over-partials kernel from T = 8 up (13.8 vs 17.4 us;
//    T = 32: 13.9 vs 37 us); lanes-over-partials only ties at T = 1 (11.4 us), so it is used for T <= 2;
//  * voices larger than one workgroup's capacity (8192 / 16384 partials) are split into chunks.
static bool short_kernel_enabled() {   // FR_BANK_SHORT=0: A/B against the time-major kernel
    static const bool on = [] { const char *e = std::getenv("FR_BANK_SHORT"); return !(e && e[0] == '0'); }();
    return on;
}

void bank_shape(uint32_t log2_p, uint32_t n_voices, uint64_t n_times, uint32_t &chunk_log2, uint32_t &frames_per_lane,
                uint32_t &waves_per_group, uint32_t &small_call, uint32_t &voices_per_wave, bool many_pairs_whole) {
    frames_per_lane = 1;
    voices_per_wave = 0;
    small_call = 0;
    {
        // short calls: few (voice, tile) pairs.  Chunks of >= 512 partials until there are ~256 workgroups of 16 waves.
        // Measured at 64 x 4096 (tools/short_call_probe.py, profiles/r02_short_calls.txt), us per call, this kernel vs the
        // time-major one: T <= 64: 7.4 vs 11.2; 128: 8.3 vs 11.4; 256: 10.6 vs 11.6; 512: 19.5 vs 17.8 -- hence pairs <= 320.
        // 512 or 1024 workgroups (more, smaller chunks) cost 2-3 us more in ticket traffic; 8 waves +0.3 us, 4 waves +2.4.
        const uint64_t pairs = ((n_times + 63) / 64) * n_voices;
        static const uint64_t max_pairs = [] { const char *e = std::getenv("FR_SHORT_PAIRS"); return e ? (uint64_t)std::atoi(e) : 1000ull; }();
        if (short_kernel_enabled() && pairs <= max_pairs && log2_p >= 9 && log2_p <= 20 && pairs > 0) {
            // up to 320 pairs: ~256 workgroups of 16 waves; up to 1000 (a GPU's share of a voice-sharded job: 8 voices x 75
            // tiles): ~1200 workgroups of 8 waves -- 600 one-voice workgroups deal 2 or 3 to a CU (28 % idle), twice as
            // many half as long deal 4 or 5 (24.2 -> 21.9 us at 8 x 4096 x 4800; profiles/r02_short_calls.txt)
            const bool few = pairs <= 320;
            // (only where whole workgroups deal unevenly over the 256 CUs: 512 pairs are 2 per CU, and splitting them costs
            //  4 us of ticket traffic for nothing -- 17.7 -> 22.0 us at 64 x 4096 x 512)
            const bool lumpy = ((pairs + 255) / 256) * 256 * 100 >= pairs * 115;
            static const uint64_t target_env = [] { const char *e = std::getenv("FR_SHORT_WGS"); return e ? (uint64_t)std::atoi(e) : 0ull; }();
            static const uint32_t nw_env = [] { const char *e = std::getenv("FR_SHORT_NW"); return e ? (uint32_t)std::atoi(e) : 0u; }();
            const uint64_t target = target_env ? target_env : (few ? 256ull : 1200ull);
            uint32_t c = log2_p;
            uint64_t wgs = pairs;
            while (c > 9 && (wgs < target || c > 13)) { --c; wgs *= 2; }
            if (log2_p - c <= 8 && (few || (lumpy && c != log2_p && !many_pairs_whole))) {
                chunk_log2 = c;
                waves_per_group = nw_env ? nw_env : (few ? 16u : 8u);
                while ((1u << c) / waves_per_group < 8u) waves_per_group /= 2;   // a wave needs a whole group of 8
                small_call = 2;
                return;
            }
        }
    }
    if (n_times <= 2 && log2_p >= 8 && n_voices <= 65535u) {   // lanes over partials (only where the short-call kernel does not apply)
        small_call = 1;
        chunk_log2 = 8;
        waves_per_group = 4;
        return;
    }
    if (log2_p <= 8) {
        // many small voices: whole voices per wave (bank_multi_kernel).  Measured with tools/bank_bench at 4800 frames:
        // 4096 x 32 partials 2.1 -> 6.7 T partial-frames/s (8 voices in a row, 2 frames per lane), 1024 x 128 5.6 -> 8.2 and
        // 512 x 256 7.1 -> 8.5 (2 in a row); profiles/r01_small_and_silent_voices.txt.  Needs enough voices to fill the chip.
        const uint32_t F = (log2_p <= 5 && n_times >= 1024) ? 2u : 1u;
        const uint64_t tiles = (n_times + 64 * F - 1) / (64 * F);
        uint32_t vpw = std::max(2u, 256u >> log2_p);
        auto nblocks = [&](uint32_t per_wave) { return ((n_voices + 4ull * per_wave - 1) / (4ull * per_wave)) * tiles; };
        while (vpw > 1 && nblocks(vpw) < 2048) vpw >>= 1;
        if (nblocks(vpw) >= 1024) {
            voices_per_wave = vpw;
            frames_per_lane = F;
            chunk_log2 = log2_p;
            waves_per_group = 4;
            return;
        }
    }
    const uint64_t blocks = ((n_times + 63) / 64) * n_voices;
    // small voices: a wave's share of the partials is a handful of groups, so the fixed cost per workgroup dominates;
    // 2 or 4 frames per lane amortise it (measured with tools/bank_bench: 32 partials 2.1 -> 3.2 T partial-frames/s,
    // 128 partials 5.3 -> 6.2, 512 partials 8.5 -> 8.8; at 4096 one frame per lane is best)
    if (n_times >= 1024 && blocks >= 4096) frames_per_lane = log2_p <= 7 ? 4 : (log2_p <= 9 ? 2 : 1);
    {   // A/B switch for measurements
        static const uint32_t f_env = [] { const char *e = std::getenv("FR_BANK_F"); return e ? (uint32_t)std::atoi(e) : 0u; }();
        if (f_env == 1 || f_env == 2 || f_env == 4) frames_per_lane = f_env;
    }
    if (n_times >= 512 && blocks < 320 && log2_p >= 10) {
        // a few big voices on a long call: too few workgroups to hide the scalar-load latency of the parameter stream
        // (one 8-wave workgroup per tile leaves a SIMD with 1-2 waves).  Split the voices into chunks of >= 512 partials,
        // about 1024 workgroups in all, plus the combine pass (tools/bank_bench: 1 x 16384 at 4800 frames 25.7 -> 21.4 us,
        // 41 us with one 2^14 chunk; 4 x 4096 20.2 -> 17.8 us; at 512 frames 13.6 -> 11.2 us)
        uint32_t c = log2_p;
        uint64_t b2 = blocks;
        while (c > 9 && b2 < 1024) { --c; b2 *= 2; }
        chunk_log2 = c;
        waves_per_group = 4;
        frames_per_lane = 1;
        return;
    }
    // (64 x 4096 at 512 / 1024 frames, 512 / 1024 workgroups: 8 waves 20.7 / 32.3 us, 4 waves 23.4 / 35.5 us, chunks of 2^11 35 / 47 us)
    // (32 x 4096 x 4800, 2400 workgroups: 8 waves 65.2 us, 4 waves 66.9; 16 x 4096: 36.0 vs 38.5; 64 x 4096: equal)
    waves_per_group = (log2_p >= 14 || (blocks < 4096 && log2_p >= 6)) ? 8 : 4;
    {
        static const uint32_t nw_env = [] { const char *e = std::getenv("FR_BANK_NW"); return e ? (uint32_t)std::atoi(e) : 0u; }();
        if (nw_env == 4 && log2_p < 14) waves_per_group = 4;   // A/B
        if (nw_env == 8 && log2_p >= 6) waves_per_group = 8;
    }
    const uint32_t cmax = waves_per_group == 8 ? 14 : 13;
    chunk_log2 = log2_p < cmax ? log2_p : cmax;
}

hipError_t launch_bank(const BankArgs &a, hipStream_t s) {
    if (a.small_call == 2) {   // short calls: chunks over workgroups, LDS-staged parameters, in-launch combine
        if (a.chunk_log2 < 7 || a.chunk_log2 > 13 || a.chunk_log2 > a.log2_p || a.log2_p - a.chunk_log2 > 8) return hipErrorInvalidValue;
        if (a.chunk_log2 != a.log2_p && (!a.ws || !a.tickets)) return hipErrorInvalidValue;
        if ((8u << a.chunk_log2) / (8u * a.waves_per_group) < 8u) return hipErrorInvalidValue;   // a wave needs a whole group
        switch (a.waves_per_group) {
        case 4: return launch_bank_short<4>(a, s);
        case 8: return launch_bank_short<8>(a, s);
        case 16: return launch_bank_short<16>(a, s);
        default: return hipErrorInvalidValue;
        }
    }
    if (a.small_call) {   // lanes over partials
        if (a.log2_p < 8 || a.log2_p > 24 || a.n_times > SMALL_MAX_FRAMES || a.chunk_log2 != 8) return hipErrorInvalidValue;
        if (a.log2_p != 8 && !a.ws) return hipErrorInvalidValue;
        if (a.n_times == 0 || a.n_voices == 0) return hipSuccess;
        if (a.n_voices > 65535u) return hipErrorInvalidValue;
        hipLaunchKernelGGL(bank_small_kernel, dim3(1u << (a.log2_p - 8u), a.n_voices), dim3(256), 0, s, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess || a.log2_p == 8) return e;
        uint64_t total = (uint64_t)a.n_voices * a.n_times;
        hipLaunchKernelGGL(bank_combine_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    if (a.voices_per_wave) {   // many small voices: whole voices per wave
        if (a.log2_p < 3 || a.log2_p > 8 || a.chunk_log2 != a.log2_p || a.voices_per_wave > 64) return hipErrorInvalidValue;
        if (a.n_times == 0 || a.n_voices == 0) return hipSuccess;
        const uint32_t F = a.frames_per_lane;
        if (F != 1 && F != 2) return hipErrorInvalidValue;
        const uint64_t tiles = (a.n_times + 64 * F - 1) / (64 * F);
        const uint64_t vgroups = ((uint64_t)a.n_voices + 4ull * a.voices_per_wave - 1) / (4ull * a.voices_per_wave);
        const uint64_t nb = tiles * vgroups;
        if (nb > 0x7FFFFFFFull) return hipErrorInvalidValue;
        const int mode = a.leaf_variant > 2 ? 1 : (int)a.leaf_variant;
        if (F == 1) {
            if (mode == 0) hipLaunchKernelGGL((bank_multi_kernel<1, 0>), dim3((uint32_t)nb), dim3(256), 0, s, a, (uint32_t)tiles, (uint32_t)nb);
            else if (mode == 1) hipLaunchKernelGGL((bank_multi_kernel<1, 1>), dim3((uint32_t)nb), dim3(256), 0, s, a, (uint32_t)tiles, (uint32_t)nb);
            else hipLaunchKernelGGL((bank_multi_kernel<1, 2>), dim3((uint32_t)nb), dim3(256), 0, s, a, (uint32_t)tiles, (uint32_t)nb);
        } else {
            if (mode == 0) hipLaunchKernelGGL((bank_multi_kernel<2, 0>), dim3((uint32_t)nb), dim3(256), 0, s, a, (uint32_t)tiles, (uint32_t)nb);
            else if (mode == 1) hipLaunchKernelGGL((bank_multi_kernel<2, 1>), dim3((uint32_t)nb), dim3(256), 0, s, a, (uint32_t)tiles, (uint32_t)nb);
            else hipLaunchKernelGGL((bank_multi_kernel<2, 2>), dim3((uint32_t)nb), dim3(256), 0, s, a, (uint32_t)tiles, (uint32_t)nb);
        }
        return hipGetLastError();
    }
    if (a.waves_per_group != 4 && a.waves_per_group != 8 && a.waves_per_group != 2 && a.waves_per_group != 1) return hipErrorInvalidValue;
    if (a.waves_per_group < 4 && (a.leaf_variant != 1 || a.host_flags)) return hipErrorInvalidValue;   // (1- and 2-wave workgroups: FMA-form leaves only)
    // a wave sums 8 .. 2048 partials (whole groups of 8, at most 8 carry levels)
    const uint32_t cmin = a.waves_per_group == 8 ? 6 : a.waves_per_group == 4 ? 5 : a.waves_per_group == 2 ? 4 : 3;
    const uint32_t cmax = a.waves_per_group == 8 ? 14 : a.waves_per_group == 4 ? 13 : a.waves_per_group == 2 ? 12 : 11;
    if (a.log2_p < cmin || a.log2_p > 24 || a.chunk_log2 < cmin || a.chunk_log2 > cmax || a.chunk_log2 > a.log2_p)
        return hipErrorInvalidValue;
    if (a.chunk_log2 != a.log2_p && !a.ws) return hipErrorInvalidValue;
    switch (a.frames_per_lane) {
    case 1: return launch_bank_f<1>(a, s);
    case 2: return launch_bank_f<2>(a, s);
    case 4: return launch_bank_f<4>(a, s);
    default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------------
// General voices: the same leaf, any Sum2 tree (odd carries, unbalanced, non-power-of-two partial counts).
// The host cuts the tree into items = maximal complete sub-trees of up to 2048 consecutive leaves and a post-order
// schedule "item, then m merges"; one workgroup per (voice, 64-frame tile) runs it (gbank_tile below).  Same bits as
// the graph: every add is the tree's own add.
// ---------------------------------------------------------------------------------------------------
// The evaluation stack lives in LDS as [level][lane] columns (a lane only touches its own column: no barriers, no
// bank conflicts); its pointer is wave-uniform.  (A first version kept the stack in 16 named registers selected by
// compare ladders: 32 v_cndmask per group, 4x slower than the balanced kernel; profiles/r01_general_tree.txt.)
constexpr uint32_t GB_MAX_DEPTH = 16;

template <bool FAST>
__device__ __forceinline__ float gbank_group(const ParamGroup &pg, uint32_t j, float t) {
    float l0 = bank_leaf<FAST, false>(t, pg.w[0], pg.A[0]);
    float v = l0;
    if (j >= 1u) {
        float l1 = bank_leaf<FAST, false>(t, pg.w[1], pg.A[1]);
        v = l0 + l1;
        if (j >= 2u) {
            float l2 = bank_leaf<FAST, false>(t, pg.w[2], pg.A[2]);
            float l3 = bank_leaf<FAST, false>(t, pg.w[3], pg.A[3]);
            v = v + (l2 + l3);
            if (j >= 3u) {
                float l4 = bank_leaf<FAST, false>(t, pg.w[4], pg.A[4]);
                float l5 = bank_leaf<FAST, false>(t, pg.w[5], pg.A[5]);
                float l6 = bank_leaf<FAST, false>(t, pg.w[6], pg.A[6]);
                float l7 = bank_leaf<FAST, false>(t, pg.w[7], pg.A[7]);
                v = v + ((l4 + l5) + (l6 + l7));
            }
        }
    }
    return v;
}

// One workgroup (4 waves) per (voice, 64-frame tile).  Items of >= 32 leaves are complete sub-trees: each wave sums a
// quarter of the item with the balanced kernel's inner loop (parameters through the scalar cache, register carry
// chain), the four quarter sums meet in LDS in the tree's own association; smaller items are evaluated by wave 0
// alone.  Wave 0 then runs the merge schedule on its LDS stack.  One barrier per big item (double-buffered hand-over:
// item i+2 reuses item i's buffer only after the barrier of item i+1, which wave 0 reaches after reading item i's).
template <bool FAST>
__device__ __forceinline__ float gbank_tile(const float *params, const uint32_t *gmeta, uint32_t nitems, float t, uint32_t wave, uint32_t lane,
                                            float *stack /* wave 0: [GB_MAX_DEPTH][64] + lane */, float (*sm)[64]) {
    uint32_t sp = 0;
    const_f32_ptr p = (const_f32_ptr)params;
    typedef uint32_t __attribute__((address_space(4))) const *const_u32_ptr;
    const_u32_ptr gm = (const_u32_ptr)gmeta;
    auto finish = [&](float v, uint32_t meta) {        // wave 0 only
        for (uint32_t m = meta >> 4; m != 0u; --m) {   // v = pop() + v
            --sp;
            v = stack[sp * 64u] + v;
        }
        stack[sp * 64u] = v;
        ++sp;
    };
    uint32_t goff = 0;          // parameter group (8 pairs) the current item starts at
    uint32_t nbig = 0;
    for (uint32_t i = 0; i < nitems; ++i) {
        const uint32_t meta = gm[i];
        const uint32_t k = meta & 15u;
        const float tt[1] = {t};
        float res[1];
        if (k >= 5u) {          // 32..2048 leaves: a quarter (2^(k-5) groups of 8) per wave
            const uint32_t gq = 1u << (k - 5u);
            bank_wave_sum<1, FAST, false>(params + ((size_t)goff + (size_t)wave * gq) * 16u, gq, k - 5u, tt, res);
            float(*buf)[64] = sm + 4u * (nbig & 1u);   // double-buffered: the next item's sums do not wait for wave 0 to read these
            ++nbig;
            buf[wave][lane] = res[0];
            __syncthreads();
            if (wave == 0u) finish((buf[0][lane] + buf[1][lane]) + (buf[2][lane] + buf[3][lane]), meta);
            goff += 1u << (k - 3u);
        } else if (wave == 0u) {
            if (k == 4u) {      // 16 leaves: two groups
                bank_wave_sum<1, FAST, false>(params + (size_t)goff * 16u, 2u, 1u, tt, res);
                finish(res[0], meta);
            } else {
                ParamGroup cur;
                load_group(cur, p, goff);
                __builtin_amdgcn_s_waitcnt(0xC07F);
                finish(gbank_group<FAST>(cur, k, t), meta);
            }
            goff += k == 4u ? 2u : 1u;
        } else {
            goff += k == 4u ? 2u : 1u;
        }
    }
    return wave == 0u ? stack[0] : 0.0f;   // a well-formed schedule leaves exactly the root
}

__global__ void __launch_bounds__(256) gbank_kernel(BankArgs a, uint32_t tiles, uint32_t nblocks) {
    __shared__ float stack_mem[GB_MAX_DEPTH][64];
    __shared__ float sm[8][64];
    uint32_t b = blockIdx.x;
    uint32_t lid = (nblocks % 8u == 0u) ? (b % 8u) * (nblocks / 8u) + b / 8u : b;
    const uint32_t voice = lid / tiles;
    const uint32_t tile = lid - voice * tiles;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t ti = (uint64_t)tile * 64u + lane;
    const float t = bank_time(a, ti);
    const bool fast = a.fast_ok && __all(t >= 0.0f && t <= 4294967296.0f);   // same for the 4 waves: same 64 frames
    const uint32_t i0 = a.group_off[2u * voice], ni = a.group_off[2u * voice + 2u] - i0;
    const float2 *vparams = a.params + (size_t)a.group_off[2u * voice + 1u] * 8u;
    float *stack = &stack_mem[0][lane];
    float r = fast ? gbank_tile<true>((const float *)vparams, a.groups + i0, ni, t, wave, lane, stack, sm)
                   : gbank_tile<false>((const float *)vparams, a.groups + i0, ni, t, wave, lane, stack, sm);
    float *orow = a.out + (size_t)a.rows[voice] * a.out_stride;
    const bool live = ti < a.n_times;
    if (wave == 0u && live) orow[bank_out_index(a, ti)] = r;
    // sign of a zero result: -0 iff every (valid) leaf is -0, see leaves_all_negzero
    __shared__ unsigned long long zshared;
    if (wave == 0u) {
        unsigned long long z = __ballot(live && r == 0.0f);
        if (lane == 0u) zshared = z;
    }
    __syncthreads();
    unsigned long long zm = zshared;   // workgroup-uniform
    if (zm == 0ull) return;
    if (__builtin_popcountll(zm) > 4) {
        // many zero frames (a silent voice): all 64 frames at once, items of >= 32 leaves split over the 4 waves like
        // the sums were, the few leaves of smaller items on wave 0 (padding pairs are not leaves and are skipped)
        bool ok = true;
        uint32_t goff = 0;
        const float *fp = (const float *)vparams;
        for (uint32_t i = 0; i < ni; ++i) {
            const uint32_t kk = a.groups[i0 + i] & 15u;
            if (kk >= 5u) {
                const uint32_t gq = 1u << (kk - 5u);
                const float *q = fp + ((size_t)goff + (size_t)wave * gq) * 16u;
                ok = ok && (fast ? wave_leaves_all_negzero<true>(q, gq, t, zm) : wave_leaves_all_negzero<false>(q, gq, t, zm));
                goff += 1u << (kk - 3u);
            } else {
                if (wave == 0u) {
                    const uint32_t sz = 1u << kk;
                    for (uint32_t j = 0; j < sz; ++j) {
                        const float2 pr = vparams[(size_t)goff * 8u + j];   // wave-uniform address
                        const float lf = fast ? bank_leaf<true, true>(t, pr.x, pr.y) : bank_leaf<false, true>(t, pr.x, pr.y);
                        ok = ok && __float_as_uint(lf) == 0x80000000u;
                    }
                }
                goff += kk == 4u ? 2u : 1u;
            }
        }
        sm[wave][lane] = ok ? 1.0f : 0.0f;     // (every wave is past the last hand-over of gbank_tile: zshared's barrier)
        __syncthreads();
        if (wave == 0u && ((zm >> lane) & 1ull)) {
            const bool all = sm[0][lane] != 0.0f && sm[1][lane] != 0.0f && sm[2][lane] != 0.0f && sm[3][lane] != 0.0f;
            orow[bank_out_index(a, ti)] = all ? -0.0f : 0.0f;
        }
        return;
    }
    if (wave != 0u) return;
    while (zm) {
        uint32_t l = (uint32_t)__builtin_ctzll(zm);
        zm &= zm - 1;
        uint64_t tz_i = (uint64_t)tile * 64u + l;
        float tz = bank_time(a, tz_i);
        bool ok = true;
        uint32_t pair0 = 0;   // first parameter pair of the item
        for (uint32_t i = 0; i < ni && ok; ++i) {
            const uint32_t kk = a.groups[i0 + i] & 15u, sz = 1u << kk;
            for (uint32_t q = lane; q < sz && ok; q += 64u) {
                float2 p = vparams[pair0 + q];
                ok = __float_as_uint(bank_leaf<false, true>(tz, p.x, p.y)) == 0x80000000u;
            }
            ok = __all(ok);
            pair0 += sz < 8u ? 8u : sz;
        }
        if (lane == 0) orow[bank_out_index(a, tz_i)] = ok ? -0.0f : 0.0f;
    }
}

// Many small general voices: like bank_multi_kernel, a wave runs WHOLE voices (items + merge schedule), several in a
// row, for one 64-frame tile; each wave has its own LDS stack; no barriers.
template <bool FAST>
__device__ __forceinline__ float gbank_voice(const float *params, const uint32_t *gmeta, uint32_t nitems, float t, float *stack /* [GB_MAX_DEPTH][64] + lane */) {
    uint32_t sp = 0;
    const_f32_ptr p = (const_f32_ptr)params;
    typedef uint32_t __attribute__((address_space(4))) const *const_u32_ptr;
    const_u32_ptr gm = (const_u32_ptr)gmeta;
    auto finish = [&](float v, uint32_t meta) {
        for (uint32_t m = meta >> 4; m != 0u; --m) {   // v = pop() + v
            --sp;
            v = stack[sp * 64u] + v;
        }
        stack[sp * 64u] = v;
        ++sp;
    };
    uint32_t goff = 0;
    for (uint32_t i = 0; i < nitems; ++i) {
        const uint32_t meta = gm[i];
        const uint32_t k = meta & 15u;
        if (k > 3u) {
            const float tt[1] = {t};
            float res[1];
            bank_wave_sum<1, FAST, false>(params + (size_t)goff * 16u, 1u << (k - 3u), k - 3u, tt, res);
            finish(res[0], meta);
            goff += 1u << (k - 3u);
        } else {
            ParamGroup cur;
            load_group(cur, p, goff);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            finish(gbank_group<FAST>(cur, k, t), meta);
            goff += 1u;
        }
    }
    return stack[0];
}

__global__ void __launch_bounds__(256) gbank_multi_kernel(BankArgs a, uint32_t tiles, uint32_t nblocks) {
    __shared__ float stack_mem[4][GB_MAX_DEPTH][64];
    uint32_t b = blockIdx.x;
    uint32_t lid = (nblocks % 8u == 0u) ? (b % 8u) * (nblocks / 8u) + b / 8u : b;
    const uint32_t vb = lid / tiles, tile = lid - vb * tiles;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t ti = (uint64_t)tile * 64u + lane;
    const float t = bank_time(a, ti);
    const bool fast = a.fast_ok && __all(t >= 0.0f && t <= 4294967296.0f);
    const bool live = ti < a.n_times;
    float *stack = &stack_mem[wave][0][lane];
    const uint32_t v0 = (vb * 4u + wave) * a.voices_per_wave;
    for (uint32_t j = 0; j < a.voices_per_wave; ++j) {
        const uint32_t voice = v0 + j;
        if (voice >= a.n_voices) break;                      // wave-uniform
        const uint32_t i0 = a.group_off[2u * voice], ni = a.group_off[2u * voice + 2u] - i0;
        const float2 *vparams = a.params + (size_t)a.group_off[2u * voice + 1u] * 8u;
        const float r = fast ? gbank_voice<true>((const float *)vparams, a.groups + i0, ni, t, stack)
                             : gbank_voice<false>((const float *)vparams, a.groups + i0, ni, t, stack);
        float *orow = a.out + (size_t)a.rows[voice] * a.out_stride;
        if (live) orow[bank_out_index(a, ti)] = r;
        // sign of exact zeros (see leaves_all_negzero), within the wave
        unsigned long long zm = __ballot(live && r == 0.0f);
        if (zm == 0ull) continue;
        if (__builtin_popcountll(zm) > 4) {                  // a silent voice: all 64 frames in one pass over its items
            bool ok = true;
            uint32_t goff = 0;
            for (uint32_t i = 0; i < ni; ++i) {
                const uint32_t kk = a.groups[i0 + i] & 15u;
                if (kk > 3u) {
                    const float *q = (const float *)vparams + (size_t)goff * 16u;
                    ok = ok && (fast ? wave_leaves_all_negzero<true>(q, 1u << (kk - 3u), t, zm) : wave_leaves_all_negzero<false>(q, 1u << (kk - 3u), t, zm));
                    goff += 1u << (kk - 3u);
                } else {
                    for (uint32_t l = 0; l < (1u << kk); ++l) {
                        const float2 pr = vparams[(size_t)goff * 8u + l];   // wave-uniform address
                        const float lf = fast ? bank_leaf<true, true>(t, pr.x, pr.y) : bank_leaf<false, true>(t, pr.x, pr.y);
                        ok = ok && __float_as_uint(lf) == 0x80000000u;
                    }
                    goff += 1u;
                }
            }
            if ((zm >> lane) & 1ull) orow[bank_out_index(a, ti)] = ok ? -0.0f : 0.0f;
            continue;
        }
        while (zm) {
            const uint32_t l = (uint32_t)__builtin_ctzll(zm);
            zm &= zm - 1;
            const uint64_t tz_i = (uint64_t)tile * 64u + l;
            const float tz = bank_time(a, tz_i);
            bool ok = true;
            uint32_t pair0 = 0;
            for (uint32_t i = 0; i < ni && ok; ++i) {
                const uint32_t kk = a.groups[i0 + i] & 15u, sz = 1u << kk;
                for (uint32_t q = lane; q < sz && ok; q += 64u) {
                    float2 p = vparams[pair0 + q];
                    ok = __float_as_uint(bank_leaf<false, true>(tz, p.x, p.y)) == 0x80000000u;
                }
                ok = __all(ok);
                pair0 += sz < 8u ? 8u : sz;
            }
            if (lane == 0) orow[bank_out_index(a, tz_i)] = ok ? -0.0f : 0.0f;
        }
    }
}

hipError_t launch_gbank(const BankArgs &a, hipStream_t s) {
    if (!a.groups || !a.group_off) return hipErrorInvalidValue;
    const uint64_t tiles = (a.n_times + 63) / 64;
    if (a.voices_per_wave) {   // many small voices: whole voices per wave
        if (a.voices_per_wave > 64) return hipErrorInvalidValue;
        const uint64_t nb = tiles * (((uint64_t)a.n_voices + 4ull * a.voices_per_wave - 1) / (4ull * a.voices_per_wave));
        if (nb == 0) return hipSuccess;
        if (nb > 0x7FFFFFFFull) return hipErrorInvalidValue;
        hipLaunchKernelGGL(gbank_multi_kernel, dim3((uint32_t)nb), dim3(256), 0, s, a, (uint32_t)tiles, (uint32_t)nb);
        return hipGetLastError();
    }
    const uint64_t nblocks64 = tiles * a.n_voices;   // one workgroup (4 waves) per (voice, tile)
    if (nblocks64 == 0) return hipSuccess;
    if (nblocks64 > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gbank_kernel, dim3((uint32_t)nblocks64), dim3(256), 0, s, a, (uint32_t)tiles, (uint32_t)nblocks64);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Staged evaluator: one thread per (program, frame).  Temporaries live in LDS as [register][thread] columns
// (conflict-free, no barriers: a thread only touches its own column).  Ring reads are the materialised
// form of the reference's "re-evaluate the source at t - d" (reference.rs:213-215): the ring holds exactly the
// values that re-evaluation would produce, because it was filled by the same graph from the same input history.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float stage_input(const StageArgs &a, uint32_t slot, uint64_t t) {
    if (slot >= a.n_inputs) return 0.0f;
    DevInput s = a.n_inputs <= STAGE_INLINE_INPUTS ? a.inline_inputs[slot] : a.inputs[slot];
    if (t < s.base || t >= s.len) return 0.0f;
    return s.data[t - s.base];
}

__device__ __forceinline__ float stage_load(const StageArgs &a, const StageInstr &in, uint64_t t) {
    switch (in.op) {
    case S_CONST: return __uint_as_float(in.imm);
    case S_INPUT: return stage_input(a, in.imm, t);
    case S_READ: return t >= in.d_lo ? a.rings[(size_t)in.buf * (a.ring_mask + 1) + ((t - in.d_lo) & a.ring_mask)] : 0.0f;
    case S_READ_INPUT: return t >= in.d_lo ? stage_input(a, in.imm, t - in.d_lo) : 0.0f;
    default: return t >= in.d_lo ? __uint_as_float(in.imm) : 0.0f;   // S_STEP
    }
}

__global__ void __launch_bounds__(256) stage_kernel(StageArgs a) {
    __shared__ float tmp[STAGE_REGS][256];
    // (kernels.hpp STAGE_CARRY: what this thread stored one iteration ago; dynamic LDS, only launches of feedback plans ask for it:
    //  the interpreter's 48 KB of registers already allow only three workgroups per CU)
    extern __shared__ float carry_mem[];
    auto carry = [&](uint32_t par, uint32_t slot) -> float & { return carry_mem[((size_t)par * STAGE_CARRY + slot) * 256u + threadIdx.x]; };
    const uint64_t wi0 = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    // one frame per thread, or (stride != 0) the frames wi0, wi0 + stride, ... of the window in order: the planner's
    // fused_stride divides every delay with which this launch reads a ring it also writes, so a thread reads only what it
    // stored itself earlier in this loop, or an earlier launch did
    const uint64_t span = a.stride ? a.stride : a.w_len;
    if (wi0 >= span) return;
    const StageProg pg = a.progs[blockIdx.y];
    const uint32_t tid = threadIdx.x;
    const StageInstr *gins = a.instrs + pg.first_instr;
    auto fetch = [&](uint32_t i) -> StageInstr { return gins[i]; };
    uint32_t par = 0;
    for (uint64_t wi = wi0; wi < a.w_len; wi += span, par ^= 1u) {
    const uint64_t t = a.w0 + wi;
    const bool carried = wi != wi0 && a.use_carry != 0u;   // (the first iteration reads what an earlier launch stored)
    auto load = [&](const StageInstr &in) -> float {
        if (in.op == S_READ && in.imm != 0u && in.imm <= STAGE_CARRY && carried) return carry(par ^ 1u, in.imm - 1u);
        return stage_load(a, in, t);
    };
    // 1. the program's loads (ring reads at t - d, inputs, constants), all in flight together
    {
        float ld[STAGE_MAX_HOISTED];
#pragma unroll
        for (uint32_t i = 0; i < STAGE_MAX_HOISTED; ++i)
            if (i < pg.n_loads) ld[i] = load(fetch(i));
#pragma unroll
        for (uint32_t i = 0; i < STAGE_MAX_HOISTED; ++i)
            if (i < pg.n_loads) tmp[fetch(i).dst][tid] = ld[i];
    }
    // 2. everything else in program order
    for (uint32_t i = pg.n_loads; i < pg.n_instr; ++i) {
        const StageInstr in = fetch(i);
        float v;
        switch (in.op) {
        case S_SUM2: v = tmp[in.a][tid] + tmp[in.b][tid]; break;
        case S_MUL: v = tmp[in.a][tid] * tmp[in.b][tid]; break;
        case S_DIV: v = tmp[in.a][tid] / tmp[in.b][tid]; break;
        case S_MOD: v = prim_mod(tmp[in.a][tid], tmp[in.b][tid]); break;
        case S_MIN: v = prim_min(tmp[in.a][tid], tmp[in.b][tid], a.sparkle != 0u); break;
        case S_STORE:
            a.rings[(size_t)in.buf * (a.ring_mask + 1) + (t & a.ring_mask)] = tmp[in.a][tid];
            if (in.imm != 0u && in.imm <= STAGE_CARRY && a.use_carry) carry(par, in.imm - 1u) = tmp[in.a][tid];
            continue;
        case S_READ_DYN: case S_READ_INPUT_DYN: case S_STEP_DYN: {   // Delay by a signal amount (reference.rs:200-215)
            uint64_t fr;
            v = 0.0f;
            if (delay_frames(tmp[in.a][tid], fr, a.sparkle != 0u) && t >= fr) {
                if (in.op == S_READ_DYN) v = a.rings[(size_t)in.buf * (a.ring_mask + 1) + ((t - fr) & a.ring_mask)];
                else if (in.op == S_READ_INPUT_DYN) v = stage_input(a, in.imm, t - fr);
                else v = __uint_as_float(in.imm);
            }
            break;
        }
        default: v = load(in); break;   // a load that was not hoisted (register budget)
        }
        tmp[in.dst][tid] = v;
    }
    const float r = tmp[pg.result_reg][tid];
    if (pg.dst_ring != 0xFFFFFFFFu) a.rings[(size_t)pg.dst_ring * (a.ring_mask + 1) + (t & a.ring_mask)] = r;
    if (pg.out_row >= 0 && t >= a.idx) a.out[(size_t)pg.out_row * a.n_times + (t - a.idx)] = r;
    if (a.stride && !a.carry_only) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this stride's ring stores are in memory before the next one reads them
    }
}

hipError_t launch_stage(const StageArgs &a, hipStream_t s) {
    if (a.n_progs == 0 || a.w_len == 0) return hipSuccess;
    if (a.n_progs > 65535u) return hipErrorInvalidValue;
    uint64_t bx = ((a.stride ? std::min(a.stride, a.w_len) : a.w_len) + 255) / 256;
    if (bx > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const size_t carry_bytes = a.use_carry ? (size_t)2 * STAGE_CARRY * 256 * sizeof(float) : 0;
    hipLaunchKernelGGL(stage_kernel, dim3((uint32_t)bx, a.n_progs), dim3(256), carry_bytes, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Partial-block exchange, one level of the voices' Sum2 trees (kernels.hpp ShardCombineArgs).  HBM-bound: 12 bytes per
// frame of a row; rows are contiguous and lanes run over frames, so every access is a full 256-byte line per wave.
__global__ void __launch_bounds__(256) shard_combine_kernel(ShardCombineArgs a) {
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint32_t row = blockIdx.y;
    if (t >= a.len) return;
    const size_t e = (size_t)row * a.len + t;
    const float v = a.lo[e] + a.hi[e];
    if (a.dst_ws) { a.dst_ws[e] = v; return; }
    const uint32_t d = a.dst[row];
    if (d & 0x80000000u) a.rings[(size_t)(d & 0x7FFFFFFFu) * (a.ring_mask + 1) + ((a.ring_t0 + t) & a.ring_mask)] = v;
    else if (t >= a.out_skip) a.out[(size_t)d * a.out_stride + (t - a.out_skip)] = v;
}

hipError_t launch_shard_combine(const ShardCombineArgs &a, hipStream_t s) {
    if (a.n_rows == 0 || a.len == 0) return hipSuccess;
    const uint64_t bx = (a.len + 255) / 256;
    if (bx > 0x7FFFFFFFull) return hipErrorInvalidValue;
    for (uint32_t r0 = 0; r0 < a.n_rows; r0 += 65535u) {   // grid.y limit
        ShardCombineArgs c = a;
        c.n_rows = std::min<uint32_t>(a.n_rows - r0, 65535u);
        c.lo += (size_t)r0 * a.len;
        c.hi += (size_t)r0 * a.len;
        if (c.dst_ws) c.dst_ws += (size_t)r0 * a.len;
        if (c.dst) c.dst += r0;
        hipLaunchKernelGGL(shard_combine_kernel, dim3((uint32_t)bx, c.n_rows), dim3(256), 0, s, c);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ---------------------------------------------------------------------------------------------------
// Pieces of a voice -> the voice (kernels.hpp ChunkCombineArgs).  HBM-bound and tiny: 4 B x 2^C per output sample.
template <int C>
__global__ void __launch_bounds__(256) chunk_combine_kernel(ChunkCombineArgs a) {
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint32_t voice = blockIdx.y;
    if (t >= a.n_times) return;
    float v[1 << C];
#pragma unroll
    for (int i = 0; i < (1 << C); ++i) v[i] = a.ws[(((size_t)voice << C) | (size_t)i) * a.n_times + t];
#pragma unroll
    for (int n = 1 << C; n > 1; n >>= 1)
#pragma unroll
        for (int i = 0; i < n / 2; ++i) v[i] = v[2 * i] + v[2 * i + 1];
    a.out[(size_t)a.rows[voice] * a.out_stride + t] = v[0];
}

hipError_t launch_chunk_combine(const ChunkCombineArgs &a, hipStream_t s) {
    if (a.n_voices == 0 || a.n_times == 0) return hipSuccess;
    if (a.log2_c < 1 || a.log2_c > 6 || a.n_voices > 65535u) return hipErrorInvalidValue;
    const dim3 grid((uint32_t)((a.n_times + 255) / 256), a.n_voices), block(256);
    switch (a.log2_c) {
    case 1: hipLaunchKernelGGL(chunk_combine_kernel<1>, grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL(chunk_combine_kernel<2>, grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL(chunk_combine_kernel<3>, grid, block, 0, s, a); break;
    case 4: hipLaunchKernelGGL(chunk_combine_kernel<4>, grid, block, 0, s, a); break;
    case 5: hipLaunchKernelGGL(chunk_combine_kernel<5>, grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL(chunk_combine_kernel<6>, grid, block, 0, s, a); break;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
__global__ void pad_kernel(float *dst, uint64_t n, const float *src_last) {
    float v = src_last ? *src_last : 0.0f;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = v;
}

hipError_t launch_pad(float *dst, uint64_t n, const float *src_last, hipStream_t s) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(pad_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, dst, n, src_last);
    return hipGetLastError();
}

}  // namespace fr
