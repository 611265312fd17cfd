// K1 f32, null-policy family (src/expressions.rs:201-296 fused into the register-resident kernel).
#define K1_NULLS_TU 1
#include "k1_kernel.inl"
namespace pols { template int k1n_launch_t<float>(pols_ctx *, int, const K1Args &, int64_t); }
