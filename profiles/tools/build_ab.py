"""Builds era-zk_evm_amd/ab_<name>.so from a git revision (or the working tree) of era-zk_evm_amd/csrc + include/, for
same-box A/B runs (profiles/tools/r02_ab_libs.sh copies each ab_*.so over libzkw.so in turn).
usage: python profiles/tools/build_ab.py <name> [<git rev> | WORK] [extra hipcc flags...]"""
import os, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name, rev = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "WORK")
extra = sys.argv[3:]
tmp = tempfile.mkdtemp()
try:
    if rev == "WORK":
        shutil.copytree(os.path.join(ROOT, "era-zk_evm_amd", "csrc"), os.path.join(tmp, "era-zk_evm_amd", "csrc"))
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    else:
        subprocess.run("git -C %s archive %s era-zk_evm_amd/csrc include | tar -x -C %s" % (ROOT, rev, tmp), shell=True, check=True)
    src = os.path.join(tmp, "era-zk_evm_amd", "csrc")
    out = os.path.join(ROOT, "era-zk_evm_amd", "ab_%s.so" % name)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-structurizecfg-skip-uniform-regions", "-I", os.path.join(tmp, "include"),
           "-o", out] + extra + [os.path.join(src, f) for f in ("zkw_kernels.hip", "zkw_commit.hip", "zkw_blake2s.hip", "zkw_expand.hip", "zkw_pack.hip", "zkw_runtime.cpp", "isa_default.cpp") if os.path.exists(os.path.join(src, f))] + ["-ldl"]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    print(out)
finally:
    shutil.rmtree(tmp)
