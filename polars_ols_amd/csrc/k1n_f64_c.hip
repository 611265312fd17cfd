// K1 null-policy family (src/expressions.rs:201-296 fused into the register-resident kernels), double, 11..15 columns.
#define K1_NULLS_TU 1
#define K1_PART_T double
#define K1_PART_LO 11
#define K1_PART_HI 15
#define K1_PART_FN k1n_launch_f64_c
#include "k1_kernel.inl"
