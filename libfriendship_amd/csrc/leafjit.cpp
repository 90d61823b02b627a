// leafjit.cpp -- source text of one shape-matched voice's leaf function (see jit.hpp).  No HIP runtime calls in this
// file: tests/cpp/plan_tests.cpp builds the generated leaf with g++ and compares it with the graph's own evaluation.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <sstream>

#include "graph.hpp"
#include "jit.hpp"

namespace fr {

// helpers every generated leaf may call (device functions; plain C++ otherwise)
static const char *kLeafHelpers = R"JIT(
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
// A track: input row number `slot` (its bits travel as a float parameter) of the call's dense [rows][stride] input matrix
// at frame ti (the kernel clamps ti to the call's last frame for the lanes of a partial tile); no matrix this call: +0.
// A slot at or beyond `limit` was not supplied (or does not exist for the reference: its input vectors number n_slots * n_times
// of the largest call so far, reference.rs:59-64, and further rows are dropped): +0, reference.rs:92-94.
__device__ __forceinline__ float jit_track(const float *trk, unsigned long long stride, unsigned limit, unsigned long long ti, float slot) {
    const unsigned s = __builtin_bit_cast(unsigned, slot);
    return (trk && s < limit) ? trk[(unsigned long long)s * stride + ti] : jit_opaque(0.0f);
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
)JIT";

LeafSource generate_leaf_source(const LeafShape &shape, const std::vector<bool> &varying, const std::vector<uint32_t> &literal_bits,
                                const std::vector<uint32_t> &alias, bool sparkle) {
    std::ostringstream leaf;
    uint32_t k = 0;
    std::vector<int> pidx(shape.n_consts, -1);
    for (uint32_t c = 0; c < shape.n_consts; ++c)
        if (varying[c]) pidx[c] = alias[c] == c ? (int)k++ : pidx[alias[c]];
    bool has_mod1 = false;
    uint32_t fract_inputs = 0;                       // inputs the argument of some Modulo(x, 1.0) depends on
    std::vector<uint32_t> dep(shape.ops.size(), 0);  // per op: mask of the inputs it depends on
    // maybe_negzero[i]: can op i evaluate to -0.0?  (conservative; a sum is -0 only if both terms are)
    std::vector<bool> maybe_negzero(shape.ops.size(), true);
    // structural equality of two sub-expressions of the tree form (constants: same literal, or the same parameter)
    std::function<bool(uint32_t, uint32_t)> same_expr = [&](uint32_t i, uint32_t j) -> bool {
        const LeafShape::Op &a = shape.ops[i], &b = shape.ops[j];
        if (a.op != b.op) return false;
        if (a.op == OP_INPUT) return a.a == b.a;
        if (a.op == OP_CONST || a.op == LEAF_TRACK) {
            if (varying[a.a] != varying[b.a]) return false;
            return varying[a.a] ? pidx[a.a] == pidx[b.a] : literal_bits[a.a] == literal_bits[b.a];
        }
        return same_expr(a.a, b.a) && same_expr(a.b, b.b);
    };
    // bound[i]: |value of op i| <= bound[i] for ANY inputs and parameters, or infinity.  (NaN is always possible and passes
    // through either form of a fold alike.)  Used by the one arithmetic fold below.
    std::vector<double> bound(shape.ops.size(), HUGE_VAL);
    auto pow2_literal = [&](uint32_t i) {   // a literal +-2^k, k >= 1: multiplying by it is exact unless it overflows (+-1 is a free sign flip)
        const LeafShape::Op &c = shape.ops[i];
        if (c.op != OP_CONST || varying[c.a]) return false;
        const uint32_t bits = literal_bits[c.a], e = (bits >> 23) & 0xFFu;
        return (bits & 0x007FFFFFu) == 0 && e >= 128 && e != 255;
    };
    auto is_literal = [&](uint32_t i, uint32_t bits) {
        const LeafShape::Op &c = shape.ops[i];
        return c.op == OP_CONST && !varying[c.a] && literal_bits[c.a] == bits;
    };
    bool tracks = false;
    for (const LeafShape::Op &o : shape.ops) tracks = tracks || o.op == LEAF_TRACK;
    leaf << "template <bool FAST>\n__device__ __forceinline__ float leaf(const float *x";
    if (tracks) leaf << ", const float *trk, unsigned long long tstride, unsigned tlimit, unsigned long long tt";
    for (uint32_t i = 0; i < (k ? k : 1); ++i) leaf << ", float p" << i;
    leaf << ") {\n    (void)x; (void)p0;\n";
    for (size_t i = 0; i < shape.ops.size(); ++i) {
        const LeafShape::Op &o = shape.ops[i];
        leaf << "    float v" << i << " = ";
        char buf[160];
        switch (o.op) {
        case OP_CONST:
            if (varying[o.a]) leaf << "p" << pidx[o.a];
            else {   // (a literal zero of either sign: opaque, see jit_opaque)
                const bool zero = (literal_bits[o.a] & 0x7FFFFFFFu) == 0;
                std::snprintf(buf, sizeof buf, zero ? "jit_opaque(__builtin_bit_cast(float, 0x%08xu))" : "__builtin_bit_cast(float, 0x%08xu)", literal_bits[o.a]);
                leaf << buf;
            }
            break;
        case OP_INPUT: leaf << "x[" << o.a << "]"; dep[i] = 1u << o.a; break;
        case LEAF_TRACK:
            // a per-leaf track: the kernel loads the value one group ahead and hands it in where the slot number's parameter
            // would be (jit.cpp wave_sum); a slot common to all leaves is loaded here (loop-invariant: once per wave)
            if (varying[o.a]) leaf << "p" << pidx[o.a];
            else { std::snprintf(buf, sizeof buf, "jit_track(trk, tstride, tlimit, tt, __builtin_bit_cast(float, 0x%08xu))", literal_bits[o.a]); leaf << buf; }
            break;
        case OP_SUM2: {
            // y + (+-2^k * v), k >= 1, |2^k * v| provably finite: the product is exact, so one fused multiply-add rounds the
            // same real number the graph's two operations round -- bit for bit, zero signs included (a product that is
            // exact keeps its sign into the sum either way).  E.g. a triangle's 1 + (-4 * |u|).
            int prod = -1, other = -1;
            static const bool fold = [] { const char *e = std::getenv("FR_JIT_FMA"); return !(e && e[0] == '0'); }();   // A/B switch
            for (int side = 0; fold && side < 2 && prod < 0; ++side) {
                const uint32_t m = side ? o.b : o.a;
                const LeafShape::Op &mo = shape.ops[m];
                if (mo.op != OP_MUL) continue;
                for (int ms = 0; ms < 2; ++ms) {
                    const uint32_t c = ms ? mo.b : mo.a, v = ms ? mo.a : mo.b;
                    if (pow2_literal(c) && bound[v] * std::fabs((double)f32_from_bits(literal_bits[shape.ops[c].a])) <= 1e38) {
                        prod = (int)m; other = (int)(side ? o.a : o.b);
                        char cb[64];
                        std::snprintf(cb, sizeof cb, "__builtin_bit_cast(float, 0x%08xu)", literal_bits[shape.ops[c].a]);
                        leaf << "__builtin_fmaf(" << cb << ", v" << v << ", v" << other << ")";
                        break;
                    }
                }
            }
            if (prod < 0) leaf << "v" << o.a << " + v" << o.b;
            break;
        }
        case OP_MUL: leaf << "v" << o.a << " * v" << o.b; break;
        case OP_DIV: leaf << "v" << o.a << " / v" << o.b; break;
        case OP_MOD: {   // a literal divisor of exactly 1.0 (every oscillator's phase wrap) avoids the generic fmodf routine
            const LeafShape::Op &d = shape.ops[o.b];
            if (d.op == OP_CONST && !varying[d.a] && literal_bits[d.a] == 0x3F800000u) {
                leaf << "(FAST ? __builtin_amdgcn_fractf(v" << o.a << ") : jit_mod1(v" << o.a << "))";
                has_mod1 = true;
                fract_inputs |= dep[o.a];
            }
            else leaf << "jit_mod(v" << o.a << ", v" << o.b << ")";
            break;
        }
        default: {
            // Minimum(x, -1 * x) == -|x| bit for bit unless x is -0.0 (the graph gives +0 there): the abs idiom the
            // reference's doc comment suggests (effect.rs:106-111), one sign-modifier instead of two compares + select
            const LeafShape::Op &nb = shape.ops[o.b];
            bool neg_of_a = nb.op == OP_MUL && ((is_literal(nb.a, 0xBF800000u) && same_expr(nb.b, o.a)) ||
                                                (is_literal(nb.b, 0xBF800000u) && same_expr(nb.a, o.a)));
            if (neg_of_a && !maybe_negzero[o.a]) leaf << "-__builtin_fabsf(v" << o.a << ")";
            else leaf << "jit_min(v" << o.a << ", v" << o.b << ")";
            break;
        }
        }
        leaf << ";\n";
        if (o.op != OP_CONST && o.op != OP_INPUT && o.op != LEAF_TRACK) dep[i] = dep[o.a] | dep[o.b];
        switch (o.op) {
        case OP_CONST: {
            const float c = f32_from_bits(literal_bits[o.a]);
            bound[i] = (!varying[o.a] && c == c && std::fabs(c) <= 3e38f) ? std::fabs((double)c) : HUGE_VAL;
            break;
        }
        case OP_INPUT: case OP_DIV: case LEAF_TRACK: bound[i] = HUGE_VAL; break;
        case OP_SUM2: bound[i] = bound[o.a] + bound[o.b]; break;
        case OP_MUL: bound[i] = (std::isinf(bound[o.a]) || std::isinf(bound[o.b])) ? HUGE_VAL : bound[o.a] * bound[o.b]; break;
        case OP_MOD: bound[i] = 2.0 * bound[o.b]; break;   // |fmod(a, b)| < |b|, and the fix-up adds b once (a finite literal b here)
        default: bound[i] = std::max(bound[o.a], bound[o.b]); break;   // OP_MIN
        }
        if (!(bound[i] <= 1e38)) bound[i] = HUGE_VAL;
        switch (o.op) {
        case OP_CONST: maybe_negzero[i] = varying[o.a] || literal_bits[o.a] == 0x80000000u; break;
        case LEAF_TRACK: maybe_negzero[i] = true; break;
        case OP_SUM2: maybe_negzero[i] = maybe_negzero[o.a] && maybe_negzero[o.b]; break;
        case OP_MIN: maybe_negzero[i] = maybe_negzero[o.a] || maybe_negzero[o.b]; break;
        default: maybe_negzero[i] = true; break;   // inputs, products, quotients, remainders: not analysed
        }
    }
    leaf << "    return v" << shape.ops.size() - 1 << ";\n}\n";
    LeafSource out;
    out.k = k ? k : 1;
    out.has_mod1 = has_mod1;
    out.fract_inputs = fract_inputs;
    out.tracks = tracks;
    for (const LeafShape::Op &o : shape.ops)
        if (o.op == LEAF_TRACK && varying[o.a]) {
            const uint32_t pi = (uint32_t)pidx[o.a];
            bool seen = false;
            for (uint32_t q : out.track_params) seen = seen || q == pi;
            if (!seen) out.track_params.push_back(pi);
        }
    out.text = std::string(sparkle ? "#define FR_SPARKLE 1\n" : "#define FR_SPARKLE 0\n") + kLeafHelpers + leaf.str();
    return out;
}

}  // namespace fr
