// K1 register-resident kernels, float, 7..8 columns.
#define K1_PART_T float
#define K1_PART_LO 7
#define K1_PART_HI 8
#define K1_PART_FN k1_launch_f32_b
#include "k1_kernel.inl"
