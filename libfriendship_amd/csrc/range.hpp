// range.hpp -- interval arithmetic over the primitives' values (host-side analysis only).
//
// Bounds are doubles, widened outward after every step so that f32 rounding cannot escape them.  `nan` = the value
// may also be NaN.  Infinite bounds make everything downstream unbounded.  Used to bound signal delay amounts
// (stage.cpp) and to prove a Modulo argument finite and non-negative in generated leaves (match.cpp).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>

#include "graph.hpp"

namespace fr {

struct Range {
    double lo, hi;
    bool nan;
    static Range unbounded() { return Range{-HUGE_VAL, HUGE_VAL, true}; }
    static Range exactly(float c) { return c != c ? Range{0.0, 0.0, true} : Range{(double)c, (double)c, false}; }
    bool finite() const { return std::isfinite(lo) && std::isfinite(hi); }
    static Range widened(double lo, double hi, bool nan) {
        if (std::isnan(lo) || std::isnan(hi)) return unbounded();
        const double FMAX = 3.4028234663852886e38;
        lo -= std::fabs(lo) * 1e-6 + 1e-30;
        hi += std::fabs(hi) * 1e-6 + 1e-30;
        if (lo < -FMAX) lo = -HUGE_VAL;
        if (hi > FMAX) hi = HUGE_VAL;
        return Range{lo, hi, nan};
    }
    // value range of op(a, b) for the five arithmetic primitives (reference.rs:221-262)
    static Range combine(uint32_t op, const Range &a, const Range &b, bool sparkle = false) {
        const bool nan = a.nan || b.nan;
        switch (op) {
        case OP_SUM2:
            if (!a.finite() || !b.finite()) return unbounded();
            return widened(a.lo + b.lo, a.hi + b.hi, nan);
        case OP_MUL: {
            if (!a.finite() || !b.finite()) return unbounded();
            double p[4] = {a.lo * b.lo, a.lo * b.hi, a.hi * b.lo, a.hi * b.hi};
            return widened(*std::min_element(p, p + 4), *std::max_element(p, p + 4), nan);
        }
        case OP_DIV: {
            if (!a.finite() || !b.finite() || (b.lo <= 0.0 && b.hi >= 0.0)) return unbounded();
            double q[4] = {a.lo / b.lo, a.lo / b.hi, a.hi / b.lo, a.hi / b.hi};
            return widened(*std::min_element(q, q + 4), *std::max_element(q, q + 4), nan);
        }
        case OP_MOD: {   // rem = fmod(a, b); rem < 0 ? rem + b : rem.  Any dividend (inf, NaN give NaN); |rem| < |b|
            if (!b.finite()) return unbounded();
            const bool may_nan = nan || !a.finite() || (b.lo <= 0.0 && b.hi >= 0.0);
            if (b.lo > 0.0) return widened(0.0, b.hi, may_nan);           // [0, b] (the sum can round up to b itself)
            const double B = std::max(std::fabs(b.lo), std::fabs(b.hi));
            return widened(-2.0 * B, B, may_nan);                         // a non-positive divisor: (-2|b|, |b|)
        }
        default: {
            if (sparkle) {   // select(a ult b, a, b): NaN iff a is; a NaN b selects a
                const double hi = b.nan ? a.hi : std::min(a.hi, b.hi);
                return Range{std::min(a.lo, b.lo), hi, a.nan};
            }
            // Minimum = (a < b || isnan(b)) ? a : b: NaN only if both are; a NaN on one side selects the other side
            double hi = (!a.nan && !b.nan) ? std::min(a.hi, b.hi) : !a.nan ? a.hi : !b.nan ? b.hi : std::max(a.hi, b.hi);
            return Range{std::min(a.lo, b.lo), hi, a.nan && b.nan};
        }
        }
    }
};

}  // namespace fr
