// K1 resident multi-pass kernels, 16..19 columns, f64 (see k1w_tu.inl)
#define K1W_T double
#define K1W_LO 16
#define K1W_FN k1w_launch_f64_a
#include "k1w_tu.inl"
