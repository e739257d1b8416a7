"""replay rate of a delivered step by host thread count (round 5): python profiles/tools/delivery_probe.py [n_batches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from era_zk_evm_amd import capi as K, synth
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10
isa = K.Isa()
prod = K.load_product().open(isa)
wl = synth.make(2, isa, n_instances=4096, n_cycles=256)
wl.limits.update(max_mem_queries=2 * 256 + 64, max_log_queries=16, max_aux_events=32)
bs = [prod.create_batch(wl) for _ in range(nb)]
arr = prod.handle_array(bs)
st = torch.cuda.Stream()
prod.step_prepared_many(arr, 256, 4, st.cuda_stream)
torch.cuda.synchronize()
for th in (1, 8, 32, 64, 128, 256):
    dv = K.Delivery(prod, 1, 160 * (1 << 20) * nb, th)
    t = dv.submit(arr, st.cuda_stream)
    info = dv.wait(t)
    best = None
    for _ in range(3 if th > 1 else 1):
        t0 = time.perf_counter(); n, acc = dv.replay(t); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print("threads %3d: replay %8.2f ms  %7.1f M cycles/s  (%5.2f M/s per thread)  bytes/cycle %.1f pack %.2f ms %.1f GB/s" % (th, 1e3 * best, n / best / 1e6, n / best / 1e6 / th, info["bytes"] / n, info["pack_ms"], info["bytes"] / info["pack_ms"] / 1e6), flush=True)
    dv.release(t); dv.close()
