// TEST INFRASTRUCTURE ONLY — exposes the Goldilocks helpers of the product header (csrc/zkw_goldilocks.hip.h, portable
// path, compiled by g++ against the HIP stand-in) so that tests/test_goldilocks_layers.py can compare them with
// big-integer arithmetic at the extremes of their input ranges.
#include <hip/hip_runtime.h>
#include "zkw_goldilocks.hip.h"
dim3 threadIdx, blockIdx, blockDim, gridDim;
extern "C" {
void t_external(u64* s) { gl_external(s); }
void t_internal(u64* s) { gl_internal(s); }
u64 t_fold(u64 l, u64 h) { return gl_fold_halves(l, h); }
u64 t_mulred(u64 a, u64 b) { return gl_mulred(a, b); }
u64 t_pow7(u64 a) { return gl_pow7(a); }
u64 t_add_rc(u64 s, u64 rc) { return gl_add_rc(s, rc); }
void t_permute(const u64* rc, u64* s) { gl_permute(rc, s); }
}
