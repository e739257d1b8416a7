"""Throughput of fully divergent tapes (every instance its own random arithmetic program) for several wave
widths: the case thin waves (limits.lanes_per_wave) exist for.  Run on the GPU box through gpurun."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from era_zk_evm_amd import capi as K, synth

isa = K.Isa()
be = K.load_product().open(isa)
# cfg 1 (arithmetic, 4096 x 256) with a different random tape for every instance
wl = synth.make(1, isa, n_instances=4096)
for i in range(1, 4096):
    wl.blobs.append(K.pack_code(synth.arith_tape(isa, 256, synth.ScalarRng(1000 + i))))
    wl.code_pages.append((i, 1, synth.BOOTLOADER_CODE_PAGE, len(wl.blobs) - 1))
for lanes in (0, 64, 16, 4, 1):  # 0 = the library's choice (thin waves: the instances were given different code)
    wl.limits["lanes_per_wave"] = lanes
    b = be.create_batch(wl)
    for rep in range(3):
        b.reset(); b.run(wl.n_cycles); b.sync()
    st = b.stats()
    print("lanes %2d: %8d cycles in %.3f ms kernel time = %.1f M cycles/s" % (lanes, int(st["cycles"]), float(st["kernel_ms"]), int(st["cycles"]) / float(st["kernel_ms"]) / 1e3))
    b.destroy()
# the same 4096 divergent instances per batch, several batches per fused launch (zkw_batches_run): full waves and the
# library's thin waves.  One batch is 64 full waves on 1024 SIMDs; a prover that owns many blocks fills the chip with them.
for lanes, nb in ((64, 8), (64, 16), (64, 32), (0, 4), (0, 8)):
    wl.limits["lanes_per_wave"] = lanes
    bs = [be.create_batch(wl) for _ in range(nb)]
    for rep in range(3):
        be.reset_many(bs); be.run_many(bs, wl.n_cycles); bs[0].sync()
    for b in bs[1:]:
        b.sync()
    cyc = sum(int(b.stats()["cycles"]) for b in bs)
    ms = float(bs[0].stats()["kernel_ms"])
    print("lanes %2d, %2d batches per launch: %9d cycles in %.3f ms kernel time = %.1f M cycles/s" % (lanes, nb, cyc, ms, cyc / ms / 1e3))
    for b in bs:
        b.destroy()
