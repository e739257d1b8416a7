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
