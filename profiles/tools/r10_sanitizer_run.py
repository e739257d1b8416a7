"""Round 6: a tour through the kernels and the host runtime of an emulation build that was compiled with sanitizers (profiles/tools/r10_sanitizers.sh):
every BASELINE configuration, divergent and shared fuzz tapes, far-call chains, delivery + replay, net states, both restage forms — each compared with the oracle."""
import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tests/emu')
import era_zk_evm_amd
from era_zk_evm_amd import capi as K, synth
from _oracle import load_oracle
import numpy as np
isa = K.Isa()
orc = load_oracle().open(isa)
emu = K.Backend(os.environ["EMULIB"], "zkw_").open(isa)
def run(be, wl):
    b = be.create_batch(wl); b.reset(); b.run(wl.n_cycles); b.sync(); return b
cases = [synth.make(1, isa, n_instances=8), synth.make(2, isa, n_instances=6), synth.make(3, isa, n_instances=3, keccak_k=(1, 2, 3, 1), sha_rounds=(1, 2, 3, 5)),
         synth.make(4, isa, n_instances=4, n_cycles=512), synth.fuzz_workload(isa, n_instances=16, n_ops=96, seed=0xF0C1), synth.uniform_fuzz(isa, n_instances=8, n_ops=160, seed=0xF1C1),
         synth.many_far_calls(isa, n_calls=12, n_instances=3), synth.bootloader_returns(isa, "heap", n_instances=3)]
for wl in cases:
    bo, be = run(orc, wl), run(emu, wl)
    bad = 0
    for i in range(wl.n_instances):
        tp = be.trace(i)
        if int(tp["status"]) == K.STATUS_LIMIT: continue
        ok, why = K.traces_equal(bo.trace(i), tp)
        bad += 0 if ok else 1
    c_eq = np.array_equal(bo.commitments(), be.commitments())
    # delivery + restage + expand + net state paths
    dv = K.Delivery(emu, 2, K.Delivery.worst_case_bytes(emu, [be]), 2)
    for off in (31, 24, 16, 8, 0):  # every part of the link format on its own (ZKW_OPT_LINK_FLAGS_OFF), the traces rebuilt from the ring against the oracle
        emu.set_option(K.OPT_LINK_FLAGS_OFF, off)
        t = dv.submit([be]); dv.wait(t); n, acc = dv.replay(t)
        for i in range(wl.n_instances):
            tp = dv.trace(t, 0, i)
            if int(tp["status"]) == K.STATUS_LIMIT: continue
            ok, why = K.traces_equal(bo.trace(i), tp)
            bad += 0 if ok else 1
        dv.release(t)
    emu.set_option(K.OPT_LINK_FLAGS_OFF, 0)
    dv.close()
    be.net_state(0)
    print(wl.name, "mismatches", bad, "commitments", c_eq, "replayed", n, flush=True)
    bo.destroy(); be.destroy()
# restage paths
wl = synth.make(2, isa, n_instances=5); b = run(emu, wl)
w2 = synth.make(2, isa, n_instances=5, seed=0x5EED7733)
b.restage(w2.states, w2.heaps); b.run(wl.n_cycles); b.sync(); b.trace(0)
sv, hv = b.staging(); sv[:] = wl.states; hv[:] = wl.heaps; b.restage(sv, hv); b.run(wl.n_cycles); b.sync(); b.trace(1)
b.destroy(); emu.close(); orc.close()
print("done")
