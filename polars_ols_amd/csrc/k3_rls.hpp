// k3_rls.hpp -- launch interface of K3 (see k3_rls.hip).
#pragma once
#include "common.hpp"

namespace pols {

struct K3Args {
    const void *y;
    const uint8_t *valid;                // row validity bytes (is_valid of ls.rs:574) or nullptr = all valid
    const void *x[POLS_MAX_FEATURES];
    const int64_t *offs;                 // device, n_groups + 1
    int64_t n_groups;
    void *coef;                          // n_rows x k (row-major, like the Array2 of ls.rs:584) or nullptr
    void *pred;                          // n_rows or nullptr
    const double *mean0;                 // device, k values (initial_state_mean) or nullptr
    double forgetting_factor;            // exp(ln 0.5 / half_life) or 1.0 (ls.rs:513-517)
    double initial_state_covariance;     // lam of ls.rs:520
    int32_t k;
};

int k3_launch(pols_ctx *ctx, int dtype, const K3Args &a);

}  // namespace pols
