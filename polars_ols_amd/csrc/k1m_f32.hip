// K1m (LDS tile + MFMA Gram) f32 instantiations.
#include "k1m_kernel.inl"
namespace pols {
template int k1m_launch_t<float>(pols_ctx *, int, const K1Args &, int64_t);
template bool k1m_fits<float>(int, bool, int64_t);
}
