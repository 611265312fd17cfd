// K2 (register-resident rows + MFMA Gram, every static solver) f64 instantiations (BASELINE configs[4]: 2 000 x 16 f64 elastic net).
#include "k2_kernel.inl"
namespace pols { template int k2_launch_t<double>(pols_ctx *, const K2Args &, int64_t); }
