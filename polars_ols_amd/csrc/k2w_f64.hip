// K2w (register-resident rows + two-tile MFMA Gram, 17..31 columns): double instantiations.
#include "k2w_kernel.inl"
namespace pols { template int k2w_launch_t<double>(pols_ctx *, const K2wArgs &, int64_t); }
