"""MI355X-native out-of-circuit EraVM witness generator (host-side Python bindings).

The product is `csrc/` (hand-written HIP kernels + the C ABI of include/zkw.h, built into
libzkw.so).  This package only binds that library with ctypes (`capi`), generates the
synthetic workloads of SURVEY.md §8d (`synth`) and drives the compiler (`build`).
There is no CPU execution path here: `capi.load_product()` raises if libzkw.so is missing and
every run fails with ZKW_ERR_DEVICE when no GPU is present.
"""
from . import build, capi, synth  # noqa: F401  (shard imports torch: import it explicitly)

__all__ = ["build", "capi", "synth"]
