// K1 f32 instantiations (cfg2: 10k groups x 1k rows x 8 feats f32 OLS).
#include "k1_kernel.inl"
namespace pols { template int k1_launch_t<float>(pols_ctx *, int, const K1Args &, int64_t); }
