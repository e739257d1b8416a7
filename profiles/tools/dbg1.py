import sys
sys.path.insert(0, '.')
import era_zk_evm_amd
from era_zk_evm_amd import capi as K, synth
from tests._oracle import load_oracle
isa = K.Isa()
prod = K.load_product().open(isa)
orc = load_oracle().open(isa)
wl = synth.make(0, isa)
outs = []
for be in (orc, prod):
    b = be.create_batch(wl); b.reset(); b.run(wl.n_cycles); b.sync()
    t = b.trace(0)
    fs = t["final_state"] if isinstance(t, dict) else t.final_state
    print(type(t))
    try:
        print({k: (fs[k] if k in fs.dtype.names else None) for k in ["timestamp","monotonic_cycle_counter","callstack_depth","previous_super_pc","previous_code_memory_page","register_ptr_bitmap","flags"]})
        print("current", fs["current"])
    except Exception as e:
        print("err", e, dir(t))
    print("status", t["status"] if isinstance(t, dict) else t.status, "n_cycles", t["n_cycles"] if isinstance(t, dict) else t.n_cycles)
    print(b.stats())
    print("mem0", t["mem"][:2])
    print("rec0 tail", t["records"][0]["tail"] if t["n_cycles"] else None)
    print("aux", t["aux"][:1]["type"] if len(t["aux"]) else None)
