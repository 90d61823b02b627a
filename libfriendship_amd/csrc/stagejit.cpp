// stagejit.cpp -- source generation for compiled stage programs (see jit.hpp).  No HIP runtime calls in this file:
// tests/cpp/plan_tests.cpp builds it for the CPU and runs the generated functions against the oracle.
#include <algorithm>
#include <cstdio>
#include <map>
#include <sstream>

#include "jit.hpp"

namespace fr {

// ---- stage programs ----------------------------------------------------------------------------------
static const char *kStageSkeleton = R"JIT(
typedef unsigned int __attribute__((address_space(4))) const *cu32;
typedef unsigned long long u64;
#if MAXP > 0
typedef const unsigned int *PRM;   // the program's parameter row, preloaded into registers by the kernel
#else
typedef cu32 PRM;
#endif

// A zero the optimiser cannot see through.  With a LITERAL zero in reach this toolchain's AMDGPU backend applies folds that
// are only valid without signed zeros -- `0.0 - y` becomes a negate modifier (-0 where IEEE gives +0 for y = +0), and
// `x < 0.0 ? x : 0.0` becomes v_min_f32 (-0 on the tie) -- found by tools/stress_parity.py, profiles/r02_stress_parity.txt.
// So generated code never shows the compiler a zero: constants +-0 and the zeros of out-of-range reads pass through here.
__device__ __forceinline__ float jit_opaque(float v) {
#if defined(__AMDGCN__)
    asm("" : "+v"(v));
#endif
    return v;
}
__device__ __forceinline__ float jit_mod(float a, float b) {   // reference.rs:254-261
    float rem = fmodf(a, b);
    return rem < 0.0f ? rem + b : rem;
}
__device__ __forceinline__ float jit_mod1(float a) {           // Modulo(a, 1.0) without the generic fmodf routine:
    float rem = __builtin_copysignf(a - truncf(a), a);          // a - trunc(a) == fmodf(a, 1) for finite a (inf - inf = NaN, like
    return rem < 0.0f ? rem + 1.0f : rem;                        // fmodf) except that a zero comes out +0: fmodf keeps a's sign
}                                                                // (-0 and the negative integers give -0); then the same fix-up
__device__ __forceinline__ float jit_min(float a, float b) {   // Rust >= 1.20 f32::min; FR_SPARKLE: select(a ult b, a, b)
#if FR_SPARKLE
    if (a != a) return a;
#endif
    const bool take_a = a < b || b != b;
    float other = b;   // not the compare's operand any more: `x < c ? x : c` with a non-NaN c would become v_min_f32 (-0 on ties)
#if defined(__AMDGCN__)
    asm("" : "+v"(other));
#endif
    return take_a ? a : other;
}
__device__ __forceinline__ float f32(unsigned int bits) { return __builtin_bit_cast(float, bits); }
__device__ __forceinline__ float in_at(const JitStageArgs &a, unsigned int slot, u64 t) {
    if (slot >= a.n_inputs) return jit_opaque(0.0f);
    JitInput s = a.n_inputs <= 8u ? a.inline_inputs[slot] : a.inputs[slot];
    if (t < s.base || t >= s.len) return jit_opaque(0.0f);
    return s.data[t - s.base];
}
__device__ __forceinline__ float in_delayed(const JitStageArgs &a, unsigned int slot, unsigned int d, u64 t) {
    return t >= d ? in_at(a, slot, t - d) : jit_opaque(0.0f);
}
__device__ __forceinline__ float ring_read(const JitStageArgs &a, unsigned int buf, unsigned int d, u64 t) {
    return t >= d ? a.rings[(size_t)buf * (a.ring_mask + 1) + ((t - d) & a.ring_mask)] : jit_opaque(0.0f);
}
// The same reads in two halves, for the block-wise fetch of strided programs: the load itself -- always issued, from a harmless
// address when the frame is out of range -- and the test that decides between its result and +0 afterwards.  (Written as one
// expression the compiler moves the load under the test's branch again and waits for it where the branch ends.)
__device__ __forceinline__ bool in_ok(const JitStageArgs &a, unsigned int slot, u64 t) {
    if (slot >= a.n_inputs) return false;
    JitInput s = a.n_inputs <= 8u ? a.inline_inputs[slot] : a.inputs[slot];
    return t >= s.base && t < s.len;
}
__device__ __forceinline__ float in_raw(const JitStageArgs &a, unsigned int slot, u64 t) {
    const float *p = (const float *)a.ptab;   // (always a readable address in HBM)
    if (slot < a.n_inputs) {
        JitInput s = a.n_inputs <= 8u ? a.inline_inputs[slot] : a.inputs[slot];
        if (t >= s.base && t < s.len) p = s.data + (t - s.base);
    }
    return *p;
}
__device__ __forceinline__ float ring_raw(const JitStageArgs &a, unsigned int buf, unsigned int d, u64 t) {
    return a.rings[(size_t)buf * (a.ring_mask + 1) + (t >= d ? ((t - d) & a.ring_mask) : 0ull)];
}
__device__ __forceinline__ void ring_store(const JitStageArgs &a, unsigned int buf, u64 t, float v) {
    a.rings[(size_t)buf * (a.ring_mask + 1) + (t & a.ring_mask)] = v;
}
__device__ __forceinline__ float step(unsigned int bits, unsigned int d, u64 t) { return t >= d ? f32(bits) : jit_opaque(0.0f); }
// Delay by a signal amount (reference.rs:200-215): >= 2^64 -> the output is 0; negative / NaN -> 0 frames; else floor
__device__ __forceinline__ bool dyn_frames(float d, u64 t, u64 &at) {
    if (d >= 18446744073709551616.0f) return false;
#if FR_SPARKLE
    if (!(d >= 0.0f)) return false;   // sparkle.rs:531-534
#endif
    u64 fr = (d < 0.0f || d != d) ? 0ull : (u64)d;
    at = t - fr;
    return t >= fr;
}
__device__ __forceinline__ float ring_read_dyn(const JitStageArgs &a, unsigned int buf, float d, u64 t) {
    u64 at;
    return dyn_frames(d, t, at) ? a.rings[(size_t)buf * (a.ring_mask + 1) + (at & a.ring_mask)] : jit_opaque(0.0f);
}
__device__ __forceinline__ float in_delayed_dyn(const JitStageArgs &a, unsigned int slot, float d, u64 t) {
    u64 at;
    return dyn_frames(d, t, at) ? in_at(a, slot, at) : jit_opaque(0.0f);
}
__device__ __forceinline__ float step_dyn(unsigned int bits, float d, u64 t) {
    u64 at;
    return dyn_frames(d, t, at) ? f32(bits) : jit_opaque(0.0f);
}

SHAPE_FUNCTIONS

extern "C" __global__ void __launch_bounds__(256) jit_stage(JitStageArgs a) {
    const u64 wi = (u64)blockIdx.x * 256u + threadIdx.x;
    // One frame per thread -- or, stride != 0, the frames wi, wi + stride, ... of the window in this order: every delayed
    // read of a ring that this launch writes reaches back a multiple of `stride` frames (the planner's fused_stride), so a
    // thread reads only what it stored itself earlier in this loop (same-thread program order) or an earlier launch did.
    const u64 span = a.stride ? a.stride : a.w_len;
    if (wi >= span) return;
    const JitStageProg pg = a.progs[blockIdx.y];
#if MAXP > 0
    // the program's parameter row, once, into scalar registers (it was re-read through the scalar cache at every use of every
    // iteration, each read waited for)
    unsigned int P[MAXP];
    {
        cu32 prow = (cu32)(a.ptab + pg.param_off);
#pragma unroll
        for (int i = 0; i < MAXP; ++i) P[i] = prow[i];
    }
#else
    cu32 P = (cu32)(a.ptab + pg.param_off);
#endif
    // carry (kernels.hpp STAGE_CARRY): what this thread stored to its rings one iteration ago stays in registers
    float cy[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, ny[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    // A strided thread's loads that depend on the frame alone (inputs, rings of banks and of earlier launches) are requested a
    // BLOCK of BLK iterations at a time, then the block's iterations computed: the iterations of a short loop cost a memory
    // latency per BLK instead of one each (a one-sample loop runs one thread per program through every frame of the call).
    if (wi < a.w_len && a.stride) {
        const u64 t = a.w0 + wi;
        switch (pg.shape) {
CARRY_CASES
        default: break;
        }
    }
    float ldb[BLK][MAXLD];
#if DEFER
    // DEFER (plans whose strided threads read their own rings through the carry only): the block's ring and output stores are
    // issued together AFTER its iterations are computed.  vmcnt counts loads and stores alike and in order, so a thread that
    // stored in every iteration waited, at the next iteration's first load, for that store to reach L2 -- a memory round trip
    // per frame of a one-sample loop either way.  Nothing inside the block reads what it stores (that is what the carry is).
    float svb[BLK][MAXST], rb[BLK];
#endif
    for (u64 base = wi; base < a.w_len; base += (u64)BLK * span) {
#pragma unroll
        for (int b = 0; b < BLK; ++b) {
            // (no branch on the lane's frame around the loads: a lane past the window's end re-reads its last frame's operands --
            //  loads inside a divergent region are waited for where it ends, one memory latency each)
            const u64 off = base + (u64)b * span;
            const u64 t = a.w0 + (off < a.w_len ? off : a.w_len - 1);
            float *ldn = ldb[b];
#pragma unroll
            for (int i = 0; i < MAXLD; ++i) ldn[i] = 0.0f;
            switch (pg.shape) {
LOAD_CASES
            default: break;
            }
        }
        // every load of the block is issued before any is used: one wait here instead of one per load
#if defined(__AMDGCN__)
#pragma unroll
        for (int b = 0; b < BLK; ++b)
#pragma unroll
            for (int i = 0; i < MAXLD; ++i) asm volatile("" : "+v"(ldb[b][i]));
#endif
#pragma unroll
        for (int b = 0; b < BLK; ++b) {
            const u64 off = base + (u64)b * span;
            if (off < a.w_len) {
                const u64 t = a.w0 + off;
                const bool carried = off != wi;
                (void)carried;
                const float *ld = ldb[b];
#if DEFER
                float *sv = svb[b];
#else
                float *sv = nullptr;
#endif
                (void)sv;
                float r;
                switch (pg.shape) {
SHAPE_CASES
                default: r = 0.0f; break;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) cy[i] = ny[i];
#if DEFER
                rb[b] = r;
#else
                if (pg.dst_ring != 0xFFFFFFFFu) ring_store(a, pg.dst_ring, t, r);
                if (pg.out_row >= 0 && t >= a.idx) a.out[(size_t)pg.out_row * a.n_times + (t - a.idx)] = r;
#if defined(__AMDGCN__)
                if (a.stride && !a.carry_only) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this stride's ring stores are in memory before the next one reads them
#endif
#endif
            }
        }
#if DEFER
#pragma unroll
        for (int b = 0; b < BLK; ++b) {
            const u64 off = base + (u64)b * span;
            if (off < a.w_len) {
                const u64 t = a.w0 + off;
                const float *sv = svb[b];
                switch (pg.shape) {
STORE_CASES
                default: break;
                }
                if (pg.dst_ring != 0xFFFFFFFFu) ring_store(a, pg.dst_ring, t, rb[b]);
                if (pg.out_row >= 0 && t >= a.idx) a.out[(size_t)pg.out_row * a.n_times + (t - a.idx)] = rb[b];
            }
        }
#endif
    }
}
)JIT";

namespace {
// constants worth baking into the source: they enable peepholes (x mod 1.0, sign flips) and never come from a knob
bool literal_worthy(uint32_t bits) {
    switch (bits & 0x7FFFFFFFu) {
    case 0x00000000u: case 0x3F800000u: case 0x3F000000u: case 0x40000000u: return true;   // 0, 1, 0.5, 2 (either sign)
    default: return false;
    }
}
}  // namespace

bool plan_stage_jit(const std::vector<StageProg> &progs, const std::vector<StageInstr> &instrs, uint32_t max_shapes, bool force,
                    StageJitPlan &out, bool sparkle, uint32_t block, bool defer_stores) {
    // Plans without feedback keep the plain form -- every load where it is used, parameters through the scalar cache: their
    // launches compute one frame per thread (or a handful of strides) and the extra set-up of the deep form only costs (config
    // D's launch 8.9 -> 10.5 us with it).
    const bool deep = block > 1 || defer_stores;
    if (progs.empty()) return false;
    struct Shape { uint32_t first; std::vector<uint32_t> members; std::vector<bool> literal; };
    std::map<std::string, uint32_t> ids;
    std::vector<Shape> shapes;
    std::vector<uint32_t> shape_of(progs.size());
    for (size_t p = 0; p < progs.size(); ++p) {
        const StageProg &pg = progs[p];
        std::string key;
        key.reserve(pg.n_instr * 4 + 8);
        for (uint32_t i = 0; i < pg.n_instr; ++i) {
            const StageInstr &in = instrs[pg.first_instr + i];
            key.push_back((char)in.op); key.push_back((char)in.dst); key.push_back((char)in.a); key.push_back((char)in.b);
            if (in.op == S_READ || in.op == S_STORE) key.push_back((char)in.imm);   // carry slot + 1 (kernels.hpp STAGE_CARRY)
        }
        key.push_back((char)pg.result_reg);
        auto it = ids.emplace(std::move(key), (uint32_t)shapes.size());
        if (it.second) {
            if (shapes.size() >= max_shapes) return false;
            shapes.push_back(Shape{(uint32_t)p, {}, {}});
        }
        shape_of[p] = it.first->second;
        shapes[it.first->second].members.push_back((uint32_t)p);
    }
    if (!force && progs.size() < 4 * shapes.size()) return false;

    // literal constants: same bits in every member and a peephole-enabling value
    for (Shape &s : shapes) {
        const StageProg &p0 = progs[s.first];
        s.literal.assign(p0.n_instr, false);
        for (uint32_t i = 0; i < p0.n_instr; ++i) {
            const StageInstr &i0 = instrs[p0.first_instr + i];
            if (i0.op != S_CONST || !literal_worthy(i0.imm)) continue;
            bool same = true;
            for (uint32_t m : s.members) same = same && instrs[progs[m].first_instr + i].imm == i0.imm;
            s.literal[i] = same;
        }
    }

    // source
    constexpr uint32_t MAXLD = 16;   // loads per program fetched ahead, a block of iterations at a time (the rest where they are used)
    uint32_t max_nld = 1, max_nst = 1, max_np = 1;
    std::ostringstream fns, cases, load_cases, store_cases, carry_cases;
    for (size_t si = 0; si < shapes.size(); ++si) {
        const Shape &s = shapes[si];
        const StageProg &p0 = progs[s.first];
        // loads that depend on the frame alone: their values are parameters `ld[]` of the shape function, produced by shapeN_ld
        std::vector<int> ld_of(p0.n_instr, -1);
        {
            std::ostringstream lf;
            lf << "__device__ __forceinline__ void shape" << si << "_ld(const JitStageArgs &a, PRM P, u64 t, float *ld) {\n    (void)a; (void)P; (void)t; (void)ld;\n";
            uint32_t kk = 0, nld = 0;
            for (uint32_t i = 0; i < p0.n_instr; ++i) {
                const StageInstr &in = instrs[p0.first_instr + i];
                // (S_READ.imm != 0: a ring the program stores itself -- carried, or 0xFF: read through memory AFTER the stores of the
                //  iterations before, so not ahead of them)
                const bool pure = deep && (in.op == S_INPUT || in.op == S_READ_INPUT || (in.op == S_READ && in.imm == 0));
                if (pure && nld < MAXLD) {
                    lf << "    ld[" << nld << "] = ";
                    if (in.op == S_INPUT) lf << "in_raw(a, P[" << kk << "], t)";
                    else if (in.op == S_READ) lf << "ring_raw(a, P[" << kk << "], P[" << kk + 1 << "], t)";
                    else lf << "in_raw(a, P[" << kk << "], t >= P[" << kk + 1 << "] ? t - P[" << kk + 1 << "] : ~0ull)";
                    lf << ";\n";
                    ld_of[i] = (int)nld++;
                }
                switch (in.op) {   // parameters consumed (the same walk as below)
                case S_CONST: if (!s.literal[i]) kk += 1; break;
                case S_INPUT: case S_STORE: case S_READ_DYN: case S_READ_INPUT_DYN: case S_STEP_DYN: kk += 1; break;
                case S_READ: case S_READ_INPUT: case S_STEP: kk += 2; break;
                default: break;
                }
            }
            lf << "}\n";
            fns << lf.str();
            max_nld = std::max(max_nld, nld);
            max_np = std::max(max_np, kk);
        }
        {   // what a thread's carry starts as: the rings' values `stride` frames before its first frame (an earlier launch's, or +0)
            std::ostringstream cf;
            cf << "__device__ __forceinline__ void shape" << si << "_cy(const JitStageArgs &a, PRM P, u64 t, float *cy) {\n    (void)a; (void)P; (void)t; (void)cy;\n";
            uint32_t kk = 0;
            for (uint32_t i = 0; i < p0.n_instr; ++i) {
                const StageInstr &in = instrs[p0.first_instr + i];
                if (in.op == S_READ && in.imm != 0 && in.imm <= 8) cf << "    cy[" << in.imm - 1 << "] = ring_read(a, P[" << kk << "], P[" << kk + 1 << "], t);\n";
                switch (in.op) {
                case S_CONST: if (!s.literal[i]) kk += 1; break;
                case S_INPUT: case S_STORE: case S_READ_DYN: case S_READ_INPUT_DYN: case S_STEP_DYN: kk += 1; break;
                case S_READ: case S_READ_INPUT: case S_STEP: kk += 2; break;
                default: break;
                }
            }
            cf << "}\n";
            fns << cf.str();
            carry_cases << "        case " << si << ": shape" << si << "_cy(a, P, t, cy); break;\n";
        }
        {   // the program's ring stores as a function of their values (DEFER: issued after a block's iterations)
            std::ostringstream sf;
            sf << "__device__ __forceinline__ void shape" << si << "_st(const JitStageArgs &a, PRM P, u64 t, const float *sv) {\n    (void)a; (void)P; (void)t; (void)sv;\n";
            uint32_t kk = 0, nst = 0;
            for (uint32_t i = 0; i < p0.n_instr; ++i) {
                const StageInstr &in = instrs[p0.first_instr + i];
                if (in.op == S_STORE) sf << "    ring_store(a, P[" << kk << "], t, sv[" << nst++ << "]);\n";
                switch (in.op) {
                case S_CONST: if (!s.literal[i]) kk += 1; break;
                case S_INPUT: case S_STORE: case S_READ_DYN: case S_READ_INPUT_DYN: case S_STEP_DYN: kk += 1; break;
                case S_READ: case S_READ_INPUT: case S_STEP: kk += 2; break;
                default: break;
                }
            }
            sf << "}\n";
            fns << sf.str();
            max_nst = std::max(max_nst, nst);
        }
        fns << "__device__ __forceinline__ float shape" << si << "(const JitStageArgs &a, PRM P, u64 t, const float *ld, const float *cy, float *ny, bool carried, float *sv) {\n"
               "    (void)a; (void)P; (void)t; (void)ld; (void)cy; (void)ny; (void)carried; (void)sv;\n";
        uint32_t n_st = 0;
        int var_of[256];
        for (int &x : var_of) x = -1;
        std::vector<bool> is_one(p0.n_instr, false);
        uint32_t k = 0;
        for (uint32_t i = 0; i < p0.n_instr; ++i) {
            const StageInstr &in = instrs[p0.first_instr + i];
            char buf[96];
            if (ld_of[i] >= 0) {   // fetched ahead (raw): the range test decides between it and +0 here, on registers
                fns << "    float v" << i << " = (";
                if (in.op == S_INPUT) fns << "in_ok(a, P[" << k << "], t)";
                else if (in.op == S_READ) fns << "t >= P[" << k + 1 << "]";
                else fns << "t >= P[" << k + 1 << "] && in_ok(a, P[" << k << "], t - P[" << k + 1 << "])";
                fns << ") ? ld[" << ld_of[i] << "] : jit_opaque(0.0f);\n";
                k += in.op == S_INPUT ? 1 : 2;
                var_of[in.dst] = (int)i;
                continue;
            }
            if (in.op == S_STORE) {
                fns << "#if DEFER\n    sv[" << n_st++ << "] = v" << var_of[in.a] << ";\n#else\n    ring_store(a, P[" << k << "], t, v" << var_of[in.a] << ");\n#endif\n";
                if (in.imm != 0 && in.imm <= 8) fns << "    ny[" << in.imm - 1 << "] = v" << var_of[in.a] << ";\n";
                k += 1;
                continue;
            }
            fns << "    float v" << i << " = ";
            switch (in.op) {
            case S_CONST:
                if (s.literal[i]) {
                    std::snprintf(buf, sizeof buf, (in.imm & 0x7FFFFFFFu) == 0 ? "jit_opaque(f32(0x%08xu))" : "f32(0x%08xu)", in.imm);
                    fns << buf;
                    is_one[i] = in.imm == 0x3F800000u;
                } else { fns << "f32(P[" << k << "])"; k += 1; }
                break;
            case S_INPUT: fns << "in_at(a, P[" << k << "], t)"; k += 1; break;
            case S_READ:
                if (in.imm != 0 && in.imm <= 8) fns << "cy[" << in.imm - 1 << "]";   // (the thread's first frame: shapeN_cy read the ring)
                else fns << "ring_read(a, P[" << k << "], P[" << k + 1 << "], t)";
                k += 2;
                break;
            case S_READ_INPUT: fns << "in_delayed(a, P[" << k << "], P[" << k + 1 << "], t)"; k += 2; break;
            case S_STEP: fns << "step(P[" << k << "], P[" << k + 1 << "], t)"; k += 2; break;
            case S_READ_DYN: fns << "ring_read_dyn(a, P[" << k << "], v" << var_of[in.a] << ", t)"; k += 1; break;
            case S_READ_INPUT_DYN: fns << "in_delayed_dyn(a, P[" << k << "], v" << var_of[in.a] << ", t)"; k += 1; break;
            case S_STEP_DYN: fns << "step_dyn(P[" << k << "], v" << var_of[in.a] << ", t)"; k += 1; break;
            case S_SUM2: fns << "v" << var_of[in.a] << " + v" << var_of[in.b]; break;
            case S_MUL: fns << "v" << var_of[in.a] << " * v" << var_of[in.b]; break;
            case S_DIV: fns << "v" << var_of[in.a] << " / v" << var_of[in.b]; break;
            case S_MOD:
                if (is_one[var_of[in.b]]) fns << "jit_mod1(v" << var_of[in.a] << ")";
                else fns << "jit_mod(v" << var_of[in.a] << ", v" << var_of[in.b] << ")";
                break;
            default: fns << "jit_min(v" << var_of[in.a] << ", v" << var_of[in.b] << ")"; break;
            }
            fns << ";\n";
            var_of[in.dst] = (int)i;
        }
        fns << "    return v" << var_of[p0.result_reg] << ";\n}\n";
        cases << "        case " << si << ": r = shape" << si << "(a, P, t, ld, cy, ny, carried, sv); break;\n";
        store_cases << "                case " << si << ": shape" << si << "_st(a, P, t, sv); break;\n";
        load_cases << "        case " << si << ": shape" << si << "_ld(a, P, t, ldn); break;\n";
    }
    std::ostringstream src;
    src << "#pragma clang fp contract(off)\n#define FR_SPARKLE " << (sparkle ? 1 : 0) << "\n#define MAXLD " << max_nld << "\n#define MAXP " << ((deep && max_np <= 64) ? max_np : 0u) << "\n#define MAXST " << max_nst << "\n#define DEFER " << ((defer_stores && max_nst <= 8) ? 1 : 0)
        << "\n#define BLK " << std::max(1u, std::min(block, std::max(1u, 32u / (max_nld + ((defer_stores && max_nst <= 8) ? max_nst + 1 : 0))))) << "\n" << FR_STR(FR_JIT_STAGE_ARGS_TEXT) << "\n";
    std::string body = kStageSkeleton;
    auto put = [&](const std::string &tag, const std::string &text) { body.replace(body.find(tag), tag.size(), text); };
    put("SHAPE_FUNCTIONS", fns.str());
    put("SHAPE_CASES", cases.str());
    put("LOAD_CASES", load_cases.str());
    put("STORE_CASES", store_cases.str());
    put("CARRY_CASES", carry_cases.str());
    src << body;
    out.source = src.str();
    out.n_shapes = (uint32_t)shapes.size();

    // parameter rows, in the order the functions above consume them
    out.progs.resize(progs.size());
    out.ptab.clear();
    for (size_t p = 0; p < progs.size(); ++p) {
        const StageProg &pg = progs[p];
        const Shape &s = shapes[shape_of[p]];
        JitStageProg jp{shape_of[p], (uint32_t)out.ptab.size(), pg.dst_ring, pg.out_row};
        for (uint32_t i = 0; i < pg.n_instr; ++i) {
            const StageInstr &in = instrs[pg.first_instr + i];
            switch (in.op) {
            case S_CONST: if (!s.literal[i]) out.ptab.push_back(in.imm); break;
            case S_INPUT: out.ptab.push_back(in.imm); break;
            case S_READ: out.ptab.push_back(in.buf); out.ptab.push_back(in.d_lo); break;
            case S_READ_INPUT: case S_STEP: out.ptab.push_back(in.imm); out.ptab.push_back(in.d_lo); break;
            case S_STORE: case S_READ_DYN: out.ptab.push_back(in.buf); break;
            case S_READ_INPUT_DYN: case S_STEP_DYN: out.ptab.push_back(in.imm); break;
            default: break;
            }
        }
        out.progs[p] = jp;
    }
    out.ptab.insert(out.ptab.end(), max_np, 0u);   // (the kernel preloads MAXP words of every row: the last rows read into this)
    return true;
}

}  // namespace fr
