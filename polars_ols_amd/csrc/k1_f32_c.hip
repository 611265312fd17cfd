// K1 register-resident kernels, float, 9..15 columns.
#define K1_PART_T float
#define K1_PART_LO 9
#define K1_PART_HI 15
#define K1_PART_FN k1_launch_f32_c
#include "k1_kernel.inl"
