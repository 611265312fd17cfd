// K1 f64 instantiations (the reference's own arithmetic type, src/expressions.rs:33,47,80).
#include "k1_kernel.inl"
namespace pols { template int k1_launch_t<double>(pols_ctx *, int, const K1Args &, int64_t); }
