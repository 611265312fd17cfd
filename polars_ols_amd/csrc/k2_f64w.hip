// K2 (register-resident rows + MFMA Gram, every static solver): double instantiations with sample weights.
#include "k2_kernel.inl"
namespace pols { template int k2_launch_t<double, true>(pols_ctx *, const K2Args &, int64_t); }
