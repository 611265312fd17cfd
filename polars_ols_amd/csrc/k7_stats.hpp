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
};

int k7_stats_launch(pols_ctx *ctx, int dtype, const StatsArgs &a);

}  // namespace pols
