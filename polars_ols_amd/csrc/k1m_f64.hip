// K1m (LDS tile + MFMA Gram) f64 instantiations.
#include "k1m_kernel.inl"
namespace pols {
template int k1m_launch_t<double>(pols_ctx *, int, const K1Args &, int64_t);
template bool k1m_fits<double>(int, bool, int64_t);
}
