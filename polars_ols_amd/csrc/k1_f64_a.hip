// K1 register-resident kernels, double, 1..7 columns.
#define K1_PART_T double
#define K1_PART_LO 1
#define K1_PART_HI 7
#define K1_PART_FN k1_launch_f64_a
#include "k1_kernel.inl"
