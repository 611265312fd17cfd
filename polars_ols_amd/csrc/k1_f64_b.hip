// K1 register-resident kernels, double, 8..15 columns.
#define K1_PART_T double
#define K1_PART_LO 8
#define K1_PART_HI 15
#define K1_PART_FN k1_launch_f64_b
#include "k1_kernel.inl"
