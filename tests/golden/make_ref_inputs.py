#!/usr/bin/env python3
"""Step 2 of the reference-fixture recipe (rust/zkw-refdump/Cargo.toml): with the REAL ISA table dumped by
`zkw-refdump dump-isa tests/golden/ref_isa.bin`, synthesise the parity workloads (their tapes are then encoded with the
reference's own variant numbering and constants) and write tests/golden/ref_inputs_<name>.bin for `zkw-refdump run`.

Without tests/golden/ref_isa.bin (this image: no cargo) `--self-check` writes the same files with the build's default
table and fabricates the "reference" outputs from the oracle, so that the loader test's plumbing is exercised; such files
go to a scratch directory, never to tests/golden/."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import era_zk_evm_amd  # noqa: E402,F401
from era_zk_evm_amd import capi as K, synth  # noqa: E402
import ref_container as RC  # noqa: E402

CASES = {
    "cfg0": lambda isa: synth.make(0, isa),
    "cfg1": lambda isa: synth.make(1, isa, n_instances=64),
    "cfg2": lambda isa: synth.make(2, isa, n_instances=64),
    "cfg3": lambda isa: synth.make(3, isa, n_instances=8, keccak_k=(1, 2, 3, 8), sha_rounds=(1, 2, 3, 5)),
    "cfg4": lambda isa: synth.make(4, isa, n_instances=32, n_cycles=1024),
    "fuzz": lambda isa: synth.fuzz_workload(isa, n_instances=64, n_ops=96, seed=0xF022),
}


def load_isa(path):
    s = RC.read_container(path)
    return K.Isa(table=np.frombuffer(s["isa"], dtype=K.ISA_TABLE).copy())


def main():
    self_check = "--self-check" in sys.argv
    out_dir = HERE
    if self_check:
        out_dir = sys.argv[sys.argv.index("--self-check") + 1]
        isa = K.Isa()
        RC.write_container(os.path.join(out_dir, "ref_isa.bin"), {"isa": isa.table.tobytes()})
    else:
        isa = load_isa(os.path.join(HERE, "ref_isa.bin"))
    for name, make in CASES.items():
        wl = make(isa)
        RC.write_container(os.path.join(out_dir, "ref_inputs_%s.bin" % name), RC.workload_sections(wl))
        if self_check:  # stand-in for `zkw-refdump run`: the oracle's own trace in the reference-output layout
            from _oracle import load_oracle
            orc = load_oracle().open(isa)
            b = orc.create_batch(wl)
            b.reset(); b.run(wl.n_cycles); b.sync()
            s = {"meta": RC.workload_sections(wl)["meta"]}
            for i in range(wl.n_instances):
                t = b.trace(i)
                import struct
                s["status%d" % i] = struct.pack("<I", t["status"])
                for k, sec in (("records", "rec"), ("mem", "mem"), ("log", "log"), ("aux", "aux"), ("mem_off", "memoff"), ("log_off", "logoff"), ("aux_off", "auxoff")):
                    s["%s%d" % (sec, i)] = np.ascontiguousarray(t[k]).tobytes()
                s["final%d" % i] = t["final_state"].tobytes()
            RC.write_container(os.path.join(out_dir, "ref_%s.bin" % name), s)
            orc.close()
        print("wrote", name)


if __name__ == "__main__":
    main()
