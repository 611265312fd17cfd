// k7_stats.hpp -- K7 "group_statistics": the mode="statistics" side-car (src/statistics.rs) for every group.
#pragma once
#include "common.hpp"

namespace pols {

struct StatsArgs {
    const void *y;
    const void *w;
    const void *x[POLS_MAX_FEATURES];
    const int64_t *offs;
    int64_t n_groups;
    const double *gram;   // n_groups x NZ x NZ from gram_stream (Z = [sqrt(w) X | sqrt(w) 1 | sqrt(w) y])
    const void *coef;     // n_groups x kt dispatcher coefficients, batch dtype
    double lambda;        // kwargs.alpha (src/expressions.rs:474)
    double *r2, *mae, *mse, *se, *tv, *pv;
    int32_t *status;
    int32_t k_user, kt;
    // LONG groups cut into segments (api.hip: ensure_segments) or nullptr: the row passes run one workgroup per SEGMENT, their sums
    // meet per group in segment order (k7_stats_launch then runs prepare / segment sums / finish instead of the one kernel)
    const int64_t *seg_offs;   // n_seg + 1
    const int32_t *seg_map;    // segment -> group
    const int32_t *seg_first;  // group -> first segment, n_groups + 1
    int64_t n_seg;
    double *seg_part;          // n_seg x 5 partial sums
    double *prep;              // n_groups x (3 kt + 1): dispatcher coefficients, A^-1 X'y, diag(A^-1), factorisation ok
};

int k7_stats_launch(pols_ctx *ctx, int dtype, const StatsArgs &a);

#if defined(__HIPCC__)
// Two-sided Student-t p-value 2 (1 - cdf(|t|)) == I_{df/(df+t^2)}(df/2, 1/2): regularised incomplete beta by Lentz' continued
// fraction (what statrs 0.17.1 evaluates for src/statistics.rs:45-49).  Shared by K7 and the wide statistics kernel (K8).
__device__ inline double k7_betacf(double a, double b, double x) {
    const double tiny = 1e-300;
    const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
    double c = 1.0, d = 1.0 - qab * x / qap;
    if (fabs(d) < tiny) d = tiny;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 500; ++m) {
        const int m2 = 2 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d; h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < 1e-16) break;
    }
    return h;
}

// regularised incomplete beta I_x(a, b)
__device__ inline double k7_betai(double a, double b, double x) {
    if (!(x > 0.0)) return (x != x) ? x : 0.0;
    if (x >= 1.0) return 1.0;
    const double bt = exp(lgamma(a + b) - lgamma(a) - lgamma(b) + a * log(x) + b * log1p(-x));
    if (x < (a + 1.0) / (a + b + 2.0)) return bt * k7_betacf(a, b, x) / a;
    return 1.0 - bt * k7_betacf(b, a, 1.0 - x) / b;
}

#endif

}  // namespace pols
