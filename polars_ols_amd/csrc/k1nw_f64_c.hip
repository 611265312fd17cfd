// K1 null-policy family, resident multi-pass kernels, 24..27 columns, f64 (see k1w_tu.inl)
#define K1_NULLS_TU 1
#define K1W_T double
#define K1W_LO 24
#define K1W_FN k1nw_launch_f64_c
#include "k1w_tu.inl"
