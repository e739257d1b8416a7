"""Throughput of zkw_blake2s256_device (era-zk_evm_amd/csrc/zkw_blake2s.hip) on one MI355X next to hashlib.blake2s on one
host core.  Buffers are resident in HBM when the timed region starts; 5 launches are timed with HIP events on the
launch stream.   python profiles/tools/blake2s_probe.py > gpurun_out/blake2s_probe.txt"""
import ctypes as C
import hashlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from era_zk_evm_amd import capi as K  # noqa: E402

torch.cuda.init()
be = K.load_product().open(K.Isa())
stream = torch.cuda.Stream()
rng = np.random.default_rng(1)
for n, size, ragged in ((1 << 20, 64, False), (1 << 20, 65, False), (1 << 20, 1024, False), (1 << 18, 4096, False), (1 << 20, 300, True)):
    lens = rng.integers(0, size + 1, size=n) if ragged else np.full(n, size)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens, dtype=np.uint64)
    total = int(offs[-1])
    data = rng.integers(0, 256, size=total + 8, dtype=np.uint8)
    d_data, d_offs = torch.from_numpy(data).cuda(), torch.from_numpy(offs.view(np.int64)).cuda()
    d_out = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")

    def launch():
        be.call("blake2s256_device", be.ctx, C.c_void_p(d_data.data_ptr()), C.c_uint64(total), C.c_void_p(d_offs.data_ptr()), C.c_uint32(n),
                C.c_void_p(d_out.data_ptr()), C.c_void_p(stream.cuda_stream))
    launch(); stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        e0.record(stream)
        for _ in range(5):
            launch()
        e1.record(stream)
    stream.synchronize()
    ms = e0.elapsed_time(e1) / 5
    out = d_out.cpu().numpy()
    raw = data.tobytes()
    for i in (0, n // 2, n - 1):
        assert out[i].tobytes() == hashlib.blake2s(raw[int(offs[i]):int(offs[i + 1])]).digest()
    k = min(n, 20000)
    t0 = time.perf_counter()
    for i in range(k):
        hashlib.blake2s(raw[int(offs[i]):int(offs[i + 1])]).digest()
    cpu_s = (time.perf_counter() - t0) * n / k
    blocks = int(np.maximum(1, (lens + 63) // 64).sum())
    print("%8d messages x %s%d B: %.3f ms  %.1f M messages/s  %.1f GB/s  %.2f G compressions/s   (hashlib, one core: %.2f M messages/s)"
          % (n, "0.." if ragged else "", size, ms, n / ms / 1e3, total / ms / 1e6, blocks / ms / 1e6, n / cpu_s / 1e6))
