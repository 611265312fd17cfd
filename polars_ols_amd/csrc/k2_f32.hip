// K2 (register-resident rows + MFMA Gram, every static solver) f32 instantiations.
#include "k2_kernel.inl"
namespace pols { template int k2_launch_t<float>(pols_ctx *, const K2Args &, int64_t); }
