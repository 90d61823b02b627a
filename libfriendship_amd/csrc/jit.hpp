// jit.hpp -- hipRTC specialisation of voices whose leaves are NOT the hand-matched partial template.
//
// The reference's second renderer, SparkleRenderer, JIT-compiles one LLVM function per effect
// (reference src/render/sparkle.rs:169-243).  This is its GPU-idiomatic analogue, scoped to
// the structure that matters on this path: a voice = balanced Sum2 tree over 2^k leaves that all have the SAME
// expression shape (any DAG of the five arithmetic primitives over inputs at t and constants) and differ only in
// up to 8 constants.  The leaf expression is printed as straight-line HIP C++ (one separately rounded op per
// primitive node, -ffp-contract=off), dropped into the same time-major bank skeleton as the hand-written kernel
// (wave-uniform parameters through the scalar cache, binary-counter carry chain, LDS across 4 waves), compiled
// once per distinct shape with hipRTC and cached.  Results are bit-identical to the graph by construction:
// nothing is algebraically folded here.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "leafshape.hpp"

namespace fr {

#define FR_STR2(...) #__VA_ARGS__
#define FR_STR(...) FR_STR2(__VA_ARGS__)

// Device-side argument block of a JIT bank kernel (kept in sync with the generated source by construction: the
// struct's text below is compiled into both).
#define FR_JIT_ARGS_TEXT                                                                                       \
    struct JitBankArgs {                                                                                       \
        const float *params;            /* [n_voices][P][K] */                                                 \
        const float *in[4];             /* input rows, starting at window frame in_skip[i] */                  \
        unsigned long long in_skip[4];                                                                         \
        unsigned long long in_valid[4];                                                                        \
        float *out;                                                                                            \
        const unsigned int *rows;                                                                              \
        unsigned long long out_stride;                                                                         \
        unsigned long long ring_mask;                                                                          \
        unsigned long long ring_t0;                                                                            \
        unsigned long long n_times;                                                                            \
        unsigned int n_voices;                                                                                 \
        unsigned int log2_p;                                                                                   \
        unsigned int tiles;                                                                                    \
        unsigned int nblocks;                                                                                  \
        unsigned int fract_ok;          /* host-proved: Modulo(x, 1) == fract(x) for inputs in [+0, 2^32] */   \
        unsigned int voices_per_wave;   /* > 0: jit_bank_multi, nblocks = tiles * ceil(n_voices / (4 * this)) */ \
        const float *tracks;            /* row 0 of the call's dense input matrix (LEAF_TRACK leaves), or null */ \
        unsigned long long track_stride; /* floats between its rows */                                          \
        unsigned int track_limit;       /* slots >= this were not supplied (or do not exist yet): they read +0 */ \
    };
FR_JIT_ARGS_TEXT

// ---- stage programs ----------------------------------------------------------------------------------
// The staged evaluator's register programs (stage.hpp) compiled instead of interpreted: programs with the same
// instruction skeleton (ops + register wiring; the usual case is one skeleton per voice position in an effects
// chain) share one straight-line function; what differs between them -- constants, ring ids, delays, input slots --
// comes from a per-program parameter row read through the scalar cache.  One kernel per plan, blockIdx.y = program.
#define FR_JIT_STAGE_ARGS_TEXT                                                                                 \
    struct JitInput { const float *data; unsigned long long base; unsigned long long len; };                   \
    struct JitStageProg { unsigned int shape, param_off, dst_ring; int out_row; };                            \
    struct JitStageArgs {                                                                                      \
        const unsigned int *ptab;       /* parameter rows */                                                   \
        const JitStageProg *progs;      /* programs of this launch (blockIdx.y) */                             \
        float *rings;                   /* [n_rings][ring_mask + 1] */                                         \
        unsigned long long ring_mask;                                                                          \
        const JitInput *inputs;         /* used when n_inputs > 8 */                                           \
        JitInput inline_inputs[8];                                                                             \
        unsigned int n_inputs;                                                                                 \
        float *out;                     /* [n_slots][n_times] of the call */                                   \
        unsigned long long n_times, idx, w0, w_len;                                                            \
        unsigned long long stride;      /* 0: thread wi computes frame w0 + wi; else frames w0 + wi + k * stride < w0 + w_len */ \
        unsigned int carry_only;        /* strided: every read of a ring this launch stores comes from the carry: no wait per stride */ \
    };
FR_JIT_STAGE_ARGS_TEXT
static_assert(sizeof(JitInput) == sizeof(DevInput), "JitInput mirrors DevInput");

struct StageJitPlan {
    std::vector<JitStageProg> progs;    // parallel to StagedPlan::progs
    std::vector<uint32_t> ptab;
    std::string source;
    uint32_t n_shapes = 0;
};
// Groups the programs by skeleton and writes the kernel source.  Returns false when specialisation is not worth a
// compile: more than `max_shapes` skeletons, or (unless `force`) fewer than 4 programs per skeleton on average.
// `block`: iterations of a strided thread whose frame-only loads are fetched together before any of them is computed (every shape
// is inlined once per iteration of a block).  Feedback loops with many iterations per thread take 16 when the stride leaves enough
// threads to be latency-bound (>= 16 frames); 1 otherwise: below that a lone wave per program is bound by its instruction stream and
// every extra instruction counts, and the 2-stride launches of effects chains measured the same at 1, 2 and 4
// (profiles/r03_feedback.txt; FR_STAGE_BLOCK overrides).
bool plan_stage_jit(const std::vector<StageProg> &progs, const std::vector<StageInstr> &instrs, uint32_t max_shapes, bool force,
                    StageJitPlan &out, bool sparkle = false, uint32_t block = 1, bool defer_stores = false);
// `defer_stores`: a block's ring and output stores are issued after its iterations (only for plans whose strided threads read their
// own rings through the carry alone, StagedPlan::fused_carry_only: nothing inside a block then reads what the block stores).

// Text of `template <bool FAST> float leaf(const float *x, float p0, ...)` for one leaf shape, preceded by the helper
// functions it calls.  FAST = the body in which Modulo(x, 1.0) is one v_fract_f32 (valid under the conditions of
// match.cpp's fract_form_is_exact + the kernel's per-wave input test).
struct LeafSource {
    std::string text;
    uint32_t k = 1;              // parameters per leaf (>= 1)
    bool has_mod1 = false;       // the leaf contains a Modulo(x, 1.0)
    uint32_t fract_inputs = 0;   // mask of the inputs its arguments depend on
    bool tracks = false;         // the leaf takes (trk, tstride, tlimit, tt) after x: it reads track rows (LEAF_TRACK)
    std::vector<uint32_t> track_params;   // parameters (0 .. k-1) that are per-leaf track slots: the leaf takes the row's VALUE there
};
LeafSource generate_leaf_source(const LeafShape &shape, const std::vector<bool> &varying, const std::vector<uint32_t> &literal_bits,
                                const std::vector<uint32_t> &alias, bool sparkle = false);

// One compiled specialisation.
struct JitKernel {
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    hipFunction_t fn_multi = nullptr;   // bank modules: the whole-voices-per-wave kernel for many small voices
    uint32_t k = 0;                     // varying constants per leaf
    ~JitKernel();
};

// hipRTC takes ~110 ms per distinct source.  By default that happens on a worker thread: get()/get_source() return
// null ("not yet") at once, the caller plans without the kernel (interpreted programs, the pull interpreter) and plans
// again when epoch() has moved -- fill_buffer never waits for a compiler.  set_async(false) compiles in the call
// (fr_config.flags & FR_CONFIG_SYNC_COMPILE: tests, benchmarks, offline rendering).
class JitCache {
public:
    JitCache();
    ~JitCache();   // waits for compiles still running
    JitCache(const JitCache &) = delete;
    JitCache &operator=(const JitCache &) = delete;
    // `varying[c]` says whether constant column c is a per-leaf parameter; literals[c] is its value otherwise.
    // Returns the compiled kernel (cached by generated source), or null while it is being compiled in the background.
    // Throws fr::Error on compile/load failure.  alias[c]: the column whose parameter column c shares (c itself if none).
    std::shared_ptr<JitKernel> get(const LeafShape &shape, const std::vector<bool> &varying, const std::vector<uint32_t> &literal_bits,
                                   const std::vector<uint32_t> &alias);
    // any generated source with one extern "C" kernel `fn_name` (cached by source text)
    std::shared_ptr<JitKernel> get_source(const std::string &src, const char *fn_name);
    static std::string generate_source(const LeafShape &shape, const std::vector<bool> &varying, const std::vector<uint32_t> &literal_bits,
                                       const std::vector<uint32_t> &alias, bool sparkle = false);
    void set_async(bool on) { async_ = on; }
    void set_sparkle(bool on) { sparkle_ = on; }   // FR_SEMANTICS_SPARKLE: baked into the generated leaves
    uint64_t epoch() const;        // bumped whenever a background compile finishes (successfully or not)
    size_t compiled() const;
    double compile_ms() const;
    size_t disk_hits() const;   // kernels whose code object came from FR_JIT_CACHE

private:
    struct Impl;
    Impl *impl_;
    bool async_ = true;
    bool sparkle_ = false;
};

hipError_t launch_jit_bank(const JitKernel &k, const JitBankArgs &a, hipStream_t s);
hipError_t launch_jit_stage(const JitKernel &k, const JitStageArgs &a, uint32_t n_progs, hipStream_t s);

}  // namespace fr
