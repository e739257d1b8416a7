// TEST INFRASTRUCTURE ONLY — storage for the emulated HIP built-ins (see hip/hip_runtime.h).
#include <hip/hip_runtime.h>
dim3 threadIdx, blockIdx, blockDim, gridDim;
uint4 zkw_lds[160 * 1024 / 16];
#if ZKW_EMU_WAVE == 1
uint32_t zkw_emu_sregs_1[64][16];  // the scalar registers of a one-lane "wave" (stream cursors), per thread of the workgroup
#endif
// lane-cycles by path (zkw_kernels.hip: ZKW_EMU_COUNT) — read and optionally cleared by the tests
extern "C" unsigned long long zkw_emu_path_counts[8];
unsigned long long zkw_emu_path_counts[8];
extern "C" void zkw_emu_get_path_counts(unsigned long long* out, int reset) {
  for (int i = 0; i < 8; i++) {
    out[i] = zkw_emu_path_counts[i];
    if (reset) zkw_emu_path_counts[i] = 0;
  }
}
