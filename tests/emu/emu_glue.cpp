// TEST INFRASTRUCTURE ONLY — storage for the emulated HIP built-ins (see hip/hip_runtime.h).
#include <hip/hip_runtime.h>
dim3 threadIdx, blockIdx, blockDim, gridDim;
uint4 zkw_lds[160 * 1024 / 16];
