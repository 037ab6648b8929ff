// k_pf_persistent for the FLAT stream layout (instantiations only; the kernel is gjx_pfilter.inl)
#include "gjx_pfilter.inl"
namespace gjx {
const void* pf_kernel_flat(int dx, int spl, bool move) { return pf_kernel_of<GJX_RNG_FLAT>(dx, spl, move); }
}  // namespace gjx
