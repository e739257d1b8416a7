"""TEST INFRASTRUCTURE ONLY — builds and loads the oracle (oracle/: the CPU restatement of the reference).

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg import this module; the package
`era-zk_evm_amd/` holds no path to oracle/.  The oracle exports the C ABI of include/zkw.h with a `zkwo_`
prefix, so the same ctypes harness (capi.Backend) drives it and the product.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "_build", "libzkw_oracle.so")


def build_oracle(force=False, native=False):
    """g++ -> oracle/_build/libzkw_oracle[_native].so; `native=True` is the -march=native flavour bench.py times."""
    out = ORACLE_LIB if not native else ORACLE_LIB.replace(".so", "_native.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("vm.cpp", "zkwo_api.cpp")]
    deps = srcs + [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith(".hpp")] + [os.path.join(ROOT, "include", "zkw.h")]
    stale = force or not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)
    if stale:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        flags = ["-O3", "-std=c++17", "-fPIC", "-shared", "-pthread"] + (["-march=native"] if native else [])
        r = subprocess.run(["g++"] + flags + ["-o", out] + srcs, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + r.stdout)
    return out


def load_oracle(native=False):
    import era_zk_evm_amd  # noqa: F401
    from era_zk_evm_amd import capi

    return capi.Backend(build_oracle(native=native), "zkwo_")
