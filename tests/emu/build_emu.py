"""TEST INFRASTRUCTURE ONLY: builds tests/emu/libzkw_emu.so / libzkw_emu64.so = the product sources compiled by g++
against the HIP stand-in of tests/emu/hip (single-lane waves / 64-lane waves on the SIMT engine of emu_simt.cpp).
Used only by the `-m "not gpu"` tests."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "era-zk_evm_amd", "csrc")
OUT = os.path.join(HERE, "libzkw_emu.so")
OUT64 = os.path.join(HERE, "libzkw_emu64.so")


def build(force=False, wave=1, defines=(), tag=""):
    """wave = 1: one-lane waves (fast); wave = 64: the SIMT engine (every lane a fiber, cross-lane operations emulated);
    defines / tag: a variant build of the kernels (-D...) under its own file name"""
    out = OUT if wave == 1 else OUT64
    if tag:
        out = out[:-3] + "_" + tag + ".so"
    srcs = [os.path.join(CSRC, "zkw_kernels.hip"), os.path.join(CSRC, "zkw_commit.hip"), os.path.join(CSRC, "zkw_blake2s.hip"), os.path.join(CSRC, "zkw_expand.hip"), os.path.join(CSRC, "zkw_pack.hip"), os.path.join(CSRC, "zkw_runtime.cpp"), os.path.join(CSRC, "isa_default.cpp"),
            os.path.join(HERE, "emu_glue.cpp"), os.path.join(HERE, "emu_simt.cpp")]
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "zkw.h"),
                                                                os.path.join(HERE, "emu_glue.cpp"), os.path.join(HERE, "emu_simt.cpp")]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wno-unknown-pragmas", "-DZKW_EMU_WAVE=%d" % wave] + ["-D" + d for d in defines] + ["-I", HERE, "-I", os.path.join(ROOT, "include"), "-o", out]
    for s in srcs:
        cmd += ["-x", "c++", s]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("emu build failed:\n" + r.stdout)
    return out


if __name__ == "__main__":
    print(build(force=True))
    print(build(force=True, wave=64))
