"""variance of zkw_delivery_replay (round 5): repeated replays of one landed step, idle GPU vs a pack kernel in flight"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from era_zk_evm_amd import capi as K, synth
nb = 5
th = int(sys.argv[1]) if len(sys.argv) > 1 else 128
isa = K.Isa()
prod = K.load_product().open(isa)
wl = synth.make(2, isa, n_instances=4096, n_cycles=256)
wl.limits.update(max_mem_queries=2 * 256 + 64, max_log_queries=16, max_aux_events=32)
bs = [prod.create_batch(wl) for _ in range(nb)]
arr = prod.handle_array(bs)
st = torch.cuda.Stream()
prod.step_prepared_many(arr, 256, 4, st.cuda_stream)
torch.cuda.synchronize()
dv = K.Delivery(prod, 3, 160 * (1 << 20) * nb, th)
t = dv.submit(arr, st.cuda_stream)
dv.wait(t)
ts = []
for _ in range(30):
    t0 = time.perf_counter(); dv.replay(t); ts.append(round(1e3 * (time.perf_counter() - t0), 2))
print("idle GPU, %d threads:" % th, ts, flush=True)
dv.release(t)
ts = []
prev = dv.submit(arr, st.cuda_stream); dv.wait(prev)
for _ in range(20):
    nxt = dv.submit(arr, st.cuda_stream)      # packs while the host replays the previous step
    t0 = time.perf_counter(); dv.replay(prev); ts.append(round(1e3 * (time.perf_counter() - t0), 2))
    dv.release(prev)
    dv.wait(nxt)
    prev = nxt
print("rolling: replay of step k while step k + 1 is packed:", ts, flush=True)
ts = []
for _ in range(15):
    dv.release(prev)
    prev = dv.submit(arr, st.cuda_stream); dv.wait(prev)
    t0 = time.perf_counter(); dv.replay(prev); ts.append(round(1e3 * (time.perf_counter() - t0), 2))
print("no overlap: replay of a freshly landed step, GPU idle:", ts, flush=True)
