// K1 register-resident kernels, float, 1..6 columns.
#define K1_PART_T float
#define K1_PART_LO 1
#define K1_PART_HI 6
#define K1_PART_FN k1_launch_f32_a
#include "k1_kernel.inl"
