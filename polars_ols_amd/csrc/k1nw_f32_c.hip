// K1 null-policy family, resident multi-pass kernels, 24..27 columns, f32 (see k1w_tu.inl)
#define K1_NULLS_TU 1
#define K1W_T float
#define K1W_LO 24
#define K1W_FN k1nw_launch_f32_c
#include "k1w_tu.inl"
