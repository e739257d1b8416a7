"""Compiler driver: builds the in-tree shared objects.

  libzkw.so          hipcc --offload-arch=gfx950  (the product: HIP kernels + C ABI)
  libzkw_isa.so      g++   (host-only ISA helpers of include/zkw.h, also linked into libzkw.so)
(The checker — oracle/ — is built by tests/_oracle.py, not from here.)
"""
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libzkw.so")
ISA_LIB = os.path.join(PKG, "libzkw_isa.so")

HIP_SOURCES = ["zkw_kernels.hip", "zkw_commit.hip", "zkw_blake2s.hip", "zkw_expand.hip", "zkw_pack.hip", "zkw_runtime.cpp", "isa_default.cpp"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def _run(cmd, cwd=None):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def hip_deps():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(ROOT, "include", "zkw.h"))
    return deps


def build_lib(force=False, extra_flags=()):
    """hipcc -> libzkw.so (gfx950). Cross-compiles without a GPU."""
    if force or _stale(LIB, hip_deps()):
        srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
        # -structurizecfg-skip-uniform-regions: the cycle kernel's control flow is almost entirely wave-uniform (scalar
        # decode per opcode-word group); leaving those regions unstructurized removes ~35 % of the AGPR spill reloads
        # (measured: lone launch 0.986 -> 0.888 ms, fused 1.25 -> 1.20 ms, profiles/r01_kernel_variants.md)
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-structurizecfg-skip-uniform-regions",
               "-I", os.path.join(ROOT, "include"), "-o", LIB] + list(extra_flags) + srcs + ["-ldl"]
        _run(cmd)
    return LIB


def build_isa(force=False):
    src = os.path.join(CSRC, "isa_default.cpp")
    if force or _stale(ISA_LIB, [src, os.path.join(ROOT, "include", "zkw.h")]):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", ISA_LIB, src])
    return ISA_LIB


def build_all(force=False):
    build_isa(force)
    build_lib(force)
