"""The container format shared with rust/zkw-refdump (magic "ZKWREF01", then sections
{u32 name_len, name, u64 data_len, data}) and the (de)serialisation of a synth.Workload into it."""
import os
import struct

import numpy as np

MAGIC = b"ZKWREF01"
HERE = os.path.dirname(os.path.abspath(__file__))


def write_container(path, sections):
    with open(path, "wb") as f:
        f.write(MAGIC)
        for name, data in sections.items():
            data = bytes(data)
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)) + nb + struct.pack("<Q", len(data)) + data)


def read_container(path):
    buf = open(path, "rb").read()
    assert buf[:8] == MAGIC, "bad magic in %s" % path
    at, out = 8, {}
    while at < len(buf):
        (nl,) = struct.unpack_from("<I", buf, at)
        at += 4
        name = buf[at:at + nl].decode()
        at += nl
        (dl,) = struct.unpack_from("<Q", buf, at)
        at += 8
        out[name] = buf[at:at + dl]
        at += dl
    return out


def workload_sections(wl):
    """synth.Workload -> sections of ref_inputs_<name>.bin (what `zkw-refdump run` reads)"""
    from era_zk_evm_amd import capi as K
    inner = np.ascontiguousarray(wl.inner)
    depth = inner.shape[1] if inner.ndim == 2 else 0
    heap_words = 0 if wl.heaps is None else int(np.asarray(wl.heaps).shape[1])
    s = {"meta": np.array([wl.n_instances, wl.n_cycles, depth, len(wl.blobs), heap_words, int(wl.zkporter_is_available)], dtype="<u4").tobytes(),
         "states": np.ascontiguousarray(wl.states).tobytes(), "inner": inner.tobytes(),
         "default_aa_code_hash": np.ascontiguousarray(wl.default_aa_code_hash, dtype="<u8").tobytes(),
         "code_pages": np.array([list(c) for c in wl.code_pages], dtype="<u4").reshape(-1, 4).tobytes(),
         "limits": repr(dict(wl.limits)).encode(), "name": wl.name.encode()}
    for i, b in enumerate(wl.blobs):
        s["blob%d" % i] = np.ascontiguousarray(b, dtype="<u8").tobytes()
    pre = b""
    for h, bi in wl.preimages:
        pre += np.ascontiguousarray(h, dtype="<u8").tobytes() + struct.pack("<I", bi)
    s["preimages"] = pre
    if wl.heaps is not None:
        heaps = np.ascontiguousarray(wl.heaps, dtype="<u8")
        for i in range(wl.n_instances):
            s["heap%d" % i] = heaps[i].tobytes()
    if wl.storage is not None:
        for i in range(wl.n_instances):
            st = np.ascontiguousarray(wl.storage[i])
            if len(st):
                assert st.dtype == K.STORAGE_SLOT
                s["storage%d" % i] = st.tobytes()
    return s


def workload_from_sections(s):
    """sections of ref_inputs_<name>.bin -> synth.Workload"""
    import ast
    from era_zk_evm_amd import capi as K, synth
    n, n_cycles, depth, n_blobs, heap_words, zkp = (int(x) for x in np.frombuffer(s["meta"], dtype="<u4")[:6])
    wl = synth.Workload(s["name"].decode(), n, n_cycles)
    wl.limits.update(ast.literal_eval(s["limits"].decode()))
    wl.blobs = [np.frombuffer(s["blob%d" % i], dtype="<u8").reshape(-1, 4).copy() for i in range(n_blobs)]
    pre = s["preimages"]
    wl.preimages = [(np.frombuffer(pre[36 * k:36 * k + 32], dtype="<u8").copy(), struct.unpack_from("<I", pre, 36 * k + 32)[0]) for k in range(len(pre) // 36)]
    wl.code_pages = [tuple(int(x) for x in row) for row in np.frombuffer(s["code_pages"], dtype="<u4").reshape(-1, 4)]
    wl.states = np.frombuffer(s["states"], dtype=K.VM_LOCAL_STATE).copy()
    wl.inner = np.frombuffer(s["inner"], dtype=K.CALLSTACK_ENTRY).reshape(n, depth).copy()
    wl.heaps = np.stack([np.frombuffer(s["heap%d" % i], dtype="<u8").reshape(-1, 4) for i in range(n)]).copy() if heap_words else None
    wl.storage = [np.frombuffer(s["storage%d" % i], dtype=K.STORAGE_SLOT).copy() if ("storage%d" % i) in s else np.zeros(0, dtype=K.STORAGE_SLOT) for i in range(n)]
    if not any(len(x) for x in wl.storage):
        wl.storage = None
    wl.default_aa_code_hash = np.frombuffer(s["default_aa_code_hash"], dtype="<u8").copy()
    wl.zkporter_is_available = zkp
    return wl


def reference_trace(s, i):
    """instance i of ref_<name>.bin (written by `zkw-refdump run`) in the shape of capi.Batch.trace()"""
    from era_zk_evm_amd import capi as K
    rec = np.frombuffer(s["rec%d" % i], dtype=K.CYCLE_RECORD)
    return {"status": struct.unpack("<I", s["status%d" % i])[0], "n_cycles": len(rec), "records": rec,
            "mem": np.frombuffer(s["mem%d" % i], dtype=K.MEM_QUERY), "log": np.frombuffer(s["log%d" % i], dtype=K.LOG_QUERY),
            "aux": np.frombuffer(s["aux%d" % i], dtype=K.AUX_EVENT),
            "mem_off": np.frombuffer(s["memoff%d" % i], dtype="<u4")[:len(rec) + 1], "log_off": np.frombuffer(s["logoff%d" % i], dtype="<u4")[:len(rec) + 1],
            "aux_off": np.frombuffer(s["auxoff%d" % i], dtype="<u4")[:len(rec) + 1],
            "final_state": np.frombuffer(s["final%d" % i], dtype=K.VM_LOCAL_STATE)[0]}
