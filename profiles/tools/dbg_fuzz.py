"""Debug aid: fuzz tapes on full waves against the oracle; first differing cycle of the first differing instances with the
instruction word that ran in it (decoded with the ISA table)."""
import sys
sys.path.insert(0, '.')
import numpy as np
import era_zk_evm_amd
from era_zk_evm_amd import capi as K, synth
from tests._oracle import load_oracle
seed = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0xf001
isa = K.Isa()
prod = K.load_product().open(isa)
orc = load_oracle().open(isa)
wl = synth.fuzz_workload(isa, n_instances=384, n_ops=96, seed=seed) if len(sys.argv) < 3 else synth.make(int(sys.argv[2]), isa, n_instances=64)
wl.limits["lanes_per_wave"] = 64
bo = orc.create_batch(wl); bo.reset(); bo.run(wl.n_cycles)
bp = prod.create_batch(wl); bp.reset(); bp.run(wl.n_cycles); bp.sync()
shown = 0
for i in range(wl.n_instances):
    a, b = bo.trace(i), bp.trace(i)
    if int(b["status"]) == K.STATUS_LIMIT:
        continue
    ok, why = K.traces_equal(a, b)
    if ok:
        continue
    shown += 1
    if shown > 6:
        continue
    ra, rb = a["records"], b["records"]
    k = 0
    while k < min(len(ra), len(rb)) and ra[k].tobytes() == rb[k].tobytes():
        k += 1
    print("instance", i, why[:80], "| first differing cycle", k, "of", len(ra), len(rb), "status", a["status"], b["status"])
    if k < min(len(ra), len(rb)):
        ta, tb = ra[k]["tail"], rb[k]["tail"]
        print("   tail oracle ", ta)
        print("   tail product", tb)
        diff = [r + 1 for r in range(15) if ra[k]["registers"][r].tobytes() != rb[k]["registers"][r].tobytes()]
        print("   regs differing", diff)
        for r in diff[:2]:
            print("     r%d oracle " % r, [hex(int(x)) for x in ra[k]["registers"][r - 1]])
            print("     r%d product" % r, [hex(int(x)) for x in rb[k]["registers"][r - 1]])
        if k:
            print("     registers before:", {r + 1: [hex(int(x)) for x in ra[k - 1]["registers"][r]] for r in range(15) if any(int(x) for x in ra[k - 1]["registers"][r])})
        lo_, hi_ = int(a["mem_off"][k]), int(a["mem_off"][k + 1])
        print("     oracle query values", [[hex(int(x)) for x in q["value"]] for q in a["mem"][lo_:hi_]])
    # the memory queries of that cycle: the first is the code fetch or operand
    for name, t in (("oracle", a), ("product", b)):
        if k < len(t["records"]):
            lo, hi = int(t["mem_off"][k]), int(t["mem_off"][k + 1])
            print("   ", name, "mem queries of the cycle:", [(int(q["page"]), int(q["index"]), hex(int(q["meta"]))) for q in t["mem"][lo:hi]])
    if k and k < len(b["records"]):
        lo = int(b["mem_off"][k])
        pcb = int(ra[k - 1]["tail"]["pc"])
        q0 = b["mem"][lo]
        if int(q0["index"]) == pcb >> 2:
            w = int(q0["value"][3 - (pcb & 3)]) if hasattr(q0["value"], "__len__") else None
            e = isa.table["entries"][0][w & 2047]
            print("   instruction word %016x: entry" % w, {n: int(e[n]) for n in e.dtype.names}, "cond", (w >> 13) & 7, "src0", (w >> 16) & 15, "src1", (w >> 20) & 15, "dst0", (w >> 24) & 15, "dst1", (w >> 28) & 15, "imm0", (w >> 32) & 0xffff, "imm1", (w >> 48) & 0xffff)
        else:
            # the word was fetched in an earlier cycle: the latest code-type query (meta & 7 == 1) with that index
            hit = None
            for q in b["mem"][:lo][::-1]:
                if int(q["index"]) == pcb >> 2 and (int(q["meta"]) & 7) == 1:
                    hit = q
                    break
            if hit is not None:
                w = int(hit["value"][3 - (pcb & 3)])
                e = isa.table["entries"][0][w & 2047]
                print("   instruction word %016x (fetched earlier): entry" % w, {n: int(e[n]) for n in e.dtype.names}, "cond", (w >> 13) & 7, "src0", (w >> 16) & 15, "src1", (w >> 20) & 15, "dst0", (w >> 24) & 15, "dst1", (w >> 28) & 15, "imm0", (w >> 32) & 0xffff, "imm1", (w >> 48) & 0xffff)
    if k:
        print("   pc before", int(ra[k - 1]["tail"]["pc"]), "sp before", int(ra[k - 1]["tail"]["sp"]))
print("differing:", shown)
