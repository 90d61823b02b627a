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

#include "leafshape.hpp"

namespace fr {

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
    };
FR_JIT_ARGS_TEXT

// One compiled specialisation.
struct JitKernel {
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    uint32_t k = 0;                     // varying constants per leaf
    ~JitKernel();
};

class JitCache {
public:
    // `varying[c]` says whether constant column c is a per-leaf parameter; literals[c] is its value otherwise.
    // Returns the compiled kernel (cached by generated source).  Throws fr::Error on compile/load failure.
    // alias[c]: the column whose parameter column c shares (c itself if none).
    std::shared_ptr<JitKernel> get(const LeafShape &shape, const std::vector<bool> &varying, const std::vector<uint32_t> &literal_bits,
                                   const std::vector<uint32_t> &alias);
    static std::string generate_source(const LeafShape &shape, const std::vector<bool> &varying, const std::vector<uint32_t> &literal_bits,
                                       const std::vector<uint32_t> &alias);
    size_t compiled() const { return compiled_; }
    double compile_ms() const { return compile_ms_; }

private:
    std::map<std::string, std::shared_ptr<JitKernel>> cache_;
    size_t compiled_ = 0;
    double compile_ms_ = 0;
};

hipError_t launch_jit_bank(const JitKernel &k, const JitBankArgs &a, hipStream_t s);

}  // namespace fr
