import sys, time, os, ctypes, collections, subprocess
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tests/emu')
from era_zk_evm_amd import capi as K, synth
sp = ctypes.CDLL('/tmp/libsprof.so')
isa = K.Isa()
lib = "/tmp/libzkw_emu64_O3.so"
emu = K.Backend(lib, "zkw_").open(isa)
wl = synth.make(2, isa, n_instances=2048); wl.limits["lanes_per_wave"] = 64
b = emu.create_batch(wl); b.reset(); b.run(wl.n_cycles); b.sync()
dv = K.Delivery(emu, 1, K.Delivery.worst_case_bytes(emu, [b]), 1)
t = dv.submit([b]); info = dv.wait(t)
sp.sprof_start()
for _ in range(300): dv.replay(t)
sp.sprof_stop(b"/tmp/sprof.out")
maps = []; samples = []
for line in open('/tmp/sprof.out'):
    f = line.split()
    if f[0] == 'M':
        lo, hi = [int(x, 16) for x in f[1].split('-')]
        maps.append((lo, hi, int(f[3], 16), f[-1]))
    else: samples.append(int(f[1], 16))
base = min(lo - off for lo, hi, off, name in maps if 'libzkw_emu64_O3' in name)
inlib = collections.Counter(); other = 0
for s in samples:
    hit = False
    for lo, hi, off, name in maps:
        if lo <= s < hi:
            if 'libzkw_emu64_O3' in name: inlib[s - base] += 1; hit = True
            break
    if not hit: other += 1
addrs = list(inlib.keys())
out = subprocess.run(["addr2line", "-e", lib, "-i"] + [hex(a) for a in addrs], stdout=subprocess.PIPE, text=True).stdout
# -i prints inlined chain: several lines per address; use plain mode but take the OUTERMOST zkw_runtime.cpp line within walk_wave range
regions = [(1603, 1750, "setup"), (1751, 1783, "deltas"), (1784, 1795, "query count"), (1796, 1826, "mem queries"), (1827, 1846, "log / aux"), (1847, 1873, "tails"), (1874, 1883, "sink call"), (2064, 2089, "fold")]
reg = collections.Counter()
for a in addrs:
    chain = subprocess.run(["addr2line", "-e", lib, "-i", hex(a)], stdout=subprocess.PIPE, text=True).stdout.split("\n")
    lines = [int(c.split(":")[1].split(" ")[0]) for c in chain if "zkw_runtime.cpp:" in c and c.split(":")[1].split(" ")[0].isdigit()]
    name = "other in lib"
    # the innermost frame decides for the fold, otherwise the frame inside walk_wave's loop
    for ln in lines:
        for lo, hi, nm in regions:
            if lo <= ln <= hi: name = nm; break
        if name != "other in lib": break
    reg[name] += inlib[a]
tot = sum(reg.values()) + other
for k, c in reg.most_common(): print("%5.1f%%  %s" % (100.0 * c / tot, k))
print("%5.1f%%  outside the library (libc: malloc / memmove)" % (100.0 * other / tot), "samples", tot)
