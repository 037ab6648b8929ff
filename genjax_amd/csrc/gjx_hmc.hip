// gjx_hmc.hip — placeholder until the HMC kernels land (entry points exist so the ABI is complete).
#include "gjx_host.h"
extern "C" int gjx_hmc(const gjx_program*, uint32_t, uint32_t, int64_t, int64_t, float, int32_t, int32_t, int32_t,
                       float*, float*, float*, float*, void*, size_t, void*) {
  return gjx_fail(GJX_EUNSUPPORTED, "gjx_hmc: not built yet");
}
extern "C" int gjx_score_grad(const gjx_program*, int64_t, const float*, float*, float*, void*) {
  return gjx_fail(GJX_EUNSUPPORTED, "gjx_score_grad: not built yet");
}
