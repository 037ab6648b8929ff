// k_pf_persistent for the reference's key structure, GJX_RNG_JAX32 (instantiations only; the kernel is gjx_pfilter.inl)
#include "gjx_pfilter.inl"
namespace gjx {
const void* pf_kernel_jax(int dx, int spl, bool move) { return pf_kernel_of<GJX_RNG_JAX32>(dx, spl, move); }
}  // namespace gjx
