// K1 null-policy family (src/expressions.rs:201-296 fused into the register-resident kernels), float, 1..7 columns.
#define K1_NULLS_TU 1
#define K1_PART_T float
#define K1_PART_LO 1
#define K1_PART_HI 7
#define K1_PART_FN k1n_launch_f32_a
#include "k1_kernel.inl"
