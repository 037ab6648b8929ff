#include "../genjax_amd/csrc/gjx_run.hip"
template __global__ void gjx::k_run_gmm<0,16,4,256>(gjx::GmmArgs);
