#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void fill(float* p, int64_t n, float v){ int64_t i=blockIdx.x*(int64_t)blockDim.x+threadIdx.x; if(i<n) p[i]=v+i; }
extern "C" int probe_fill(float* p, int64_t n, float v, void* stream){
  hipLaunchKernelGGL(fill, dim3((n+255)/256), dim3(256), 0, (hipStream_t)stream, p, n, v);
  return (int)hipGetLastError();
}
extern "C" int probe_devcount(){ int c=-1; hipError_t e=hipGetDeviceCount(&c); return e==hipSuccess? c : -(int)e; }
