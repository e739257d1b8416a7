"""How long one zkw_batch_restage call keeps its caller (host side), copying and in-place form, and a plain numpy copy of the same bytes
into the same pinned staging for comparison.  python profiles/tools/restage_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from era_zk_evm_amd import capi as K, synth
torch.cuda.init()
isa = K.Isa()
prod = K.load_product().open(isa)
wl = synth.make(2, isa, n_instances=4096)
b = prod.create_batch(wl)
b.reset(); b.run(wl.n_cycles); b.sync()
w2 = synth.make(2, isa, n_instances=4096, seed=0x5EED9001)
st = torch.cuda.Stream()
def t(fn, n=8):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return "%.2f ms (min %.2f)" % (1e3 * sorted(ts)[len(ts) // 2], 1e3 * min(ts))
print("copying restage, host time of the call:", t(lambda: b.restage(w2.states, w2.heaps, st.cuda_stream)))
sv, hv = b.staging()
print("in-place restage:", t(lambda: b.restage(sv, hv, st.cuda_stream)))
print("numpy copy of the heaps into the pinned staging:", t(lambda: np.copyto(hv, w2.heaps)))
tmp = np.empty_like(w2.heaps)
print("numpy copy of the heaps into pageable memory:", t(lambda: np.copyto(tmp, w2.heaps)))
print("bytes", w2.heaps.nbytes + w2.states.nbytes)
# several batches restaged at once from a thread pool (bench.py's upload leg): does the host side scale?
from concurrent.futures import ThreadPoolExecutor
bs = [b] + [prod.create_batch(wl) for _ in range(9)]
for x in bs[1:]:
    x.reset(); x.run(wl.n_cycles); x.sync()
for nt in (1, 2, 5, 10):
    pool = ThreadPoolExecutor(max_workers=nt)
    print("10 copying restages on %d threads:" % nt, t(lambda: list(pool.map(lambda x: x.restage(w2.states, w2.heaps, st.cuda_stream), bs)), n=5))
    pool.shutdown()
dsts = [np.empty_like(w2.heaps) for _ in range(10)]
for nt in (1, 5, 10):
    pool = ThreadPoolExecutor(max_workers=nt)
    print("10 numpy copies of the heaps on %d threads:" % nt, t(lambda: list(pool.map(lambda d: np.copyto(d, w2.heaps), dsts)), n=5))
    pool.shutdown()
