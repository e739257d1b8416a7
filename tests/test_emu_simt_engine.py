"""The SIMT engine behind tests/emu/libzkw_emu64.so (tests/emu/emu_simt.cpp) tested on its own — who tests the tester: small
kernels with closed-form results (tests/emu/simt_selftest.cpp) check which lanes take part in a cross-lane operation under the
divergence scopes, early exits, ragged loops, the cursor registers, shuffles and the workgroup barrier; and a cross-lane
operation in a divergent region WITHOUT an annotation must abort the process with the source line, not pass."""
import os
import subprocess

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
EXE = os.path.join(HERE, "simt_selftest")


def _build():
    srcs = [os.path.join(HERE, f) for f in ("simt_selftest.cpp", "emu_simt.cpp", "emu_glue.cpp")]
    deps = srcs + [os.path.join(HERE, "hip", "hip_runtime.h")]
    if os.path.exists(EXE) and all(os.path.getmtime(d) <= os.path.getmtime(EXE) for d in deps):
        return
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-DZKW_EMU_WAVE=64", "-I", HERE, "-o", EXE] + srcs, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout


def test_scopes_collectives_and_barriers():
    _build()
    r = subprocess.run([EXE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout


def test_an_unannotated_divergent_collective_aborts():
    _build()
    r = subprocess.run([EXE, "unannotated"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode != 0 and "NOT DETECTED" not in r.stdout, r.stdout
    assert "DIFFERENT cross-lane operations" in r.stdout and "simt_selftest.cpp" in r.stdout, r.stdout
