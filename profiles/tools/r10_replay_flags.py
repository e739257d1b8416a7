import sys, time, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tests/emu')
from era_zk_evm_amd import capi as K, synth
isa = K.Isa()
lib = os.environ.get("EMULIB", "/tmp/libzkw_emu64_O3.so")
emu = K.Backend(lib, "zkw_").open(isa)
wl = synth.make(int(os.environ.get("CFG", "2")), isa, n_instances=2048); wl.limits["lanes_per_wave"] = 64
b = emu.create_batch(wl); b.reset(); b.run(wl.n_cycles); b.sync()
cycles = int(b.stats()["cycles"])
dv = K.Delivery(emu, 1, K.Delivery.worst_case_bytes(emu, [b]), 1)
for off in (31, 0, 1, 2, 4, 8, 16, 24):
    emu.set_option(K.OPT_LINK_FLAGS_OFF, off)
    t = dv.submit([b]); info = dv.wait(t)
    best = 1e9
    for _ in range(12):
        t0 = time.time(); dv.replay(t); best = min(best, time.time() - t0)
    print("flags %2d: %.1f B/cycle, %.1f M cycles/s  (%.1f ns/cycle)" % (info["link_flags"], info["bytes"] / cycles, cycles / best / 1e6, best / cycles * 1e9))
    dv.release(t)
