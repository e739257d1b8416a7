"""TEST INFRASTRUCTURE ONLY: builds tests/emu/libzkw_emu.so = the product sources compiled by g++
against the single-lane HIP stand-in.  Used only by the `-m "not gpu"` tests."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "era-zk_evm_amd", "csrc")
OUT = os.path.join(HERE, "libzkw_emu.so")


def build(force=False):
    srcs = [os.path.join(CSRC, "zkw_kernels.hip"), os.path.join(CSRC, "zkw_commit.hip"), os.path.join(CSRC, "zkw_blake2s.hip"), os.path.join(CSRC, "zkw_expand.hip"), os.path.join(CSRC, "zkw_pack.hip"), os.path.join(CSRC, "zkw_runtime.cpp"), os.path.join(CSRC, "isa_default.cpp"),
            os.path.join(HERE, "emu_glue.cpp")]
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "zkw.h"),
                                                                os.path.join(HERE, "emu_glue.cpp")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-I", HERE, "-I", os.path.join(ROOT, "include"), "-o", OUT]
    for s in srcs:
        cmd += ["-x", "c++", s]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("emu build failed:\n" + r.stdout)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
