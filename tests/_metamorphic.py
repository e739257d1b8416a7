"""TEST INFRASTRUCTURE.  "The table is host-uploaded" (SURVEY 7.1's mitigation for the absent zkevm_opcode_defs) made a
test: the same programs are built against tables that differ from the recalled default in what the absent crate decides.

  renumbered():  the 2048 variants in a random order, the Condition a 3-bit field value names permuted, clip_mode 1 (the
                 cfg tapes only ever clip values < 2^16).  Nothing of this changes what an instruction DOES, so after the
                 opcode words are mapped back the witness must be the witness under the default table — product against
                 oracle AND product against product-under-default.
  estranged():   on top: other prices, other forwarding-mode byte codes, other register conventions of far_call / ret.
                 These change the witness (ergs, ABI words, which register receives the calldata pointer), so the check is
                 product against oracle under the same table.
"""
import numpy as np

from era_zk_evm_amd import capi as K, synth


def _moved_always(make, seed):
    """the first table from `seed` on whose condition field 0 no longer names Always (the masked encodings then carry a field too)"""
    while True:
        isa = make(seed)
        if isa.cond_field(K.COND_ALWAYS) != 0:
            return isa
        seed += 1


def renumbered(seed):
    return _moved_always(lambda s: K.Isa.variant_of_default(s, permute=True, permute_conditions=True, clip_mode=1), seed)


def estranged(seed):
    return _moved_always(lambda s: K.Isa.variant_of_default(s, permute=True, permute_conditions=True, reprice=True, swap_forwarding=True, shift_registers=True), seed)


WORKLOADS = {
    "cfg1": lambda isa: synth.make(1, isa, n_instances=24),
    "cfg2": lambda isa: synth.make(2, isa, n_instances=12),
    "cfg4": lambda isa: synth.make(4, isa, n_instances=5, n_cycles=1024),
}


def _maps_back(isa_var, var_word, def_word):
    """a code word (4 x u64, opcode k in limb 3 - k) under `isa_var` is the default table's word: every opcode maps back, or is the
    same bits in both (padding / constants that were never built through the encoder)"""
    for k in range(4):
        v, d = int(var_word[k]), int(def_word[k])
        if v != d and isa_var.canonical_opcode(v) != d:
            return False
    return True


def same_witness(isa_var, t_def, t_var):
    """-> (ok, why): the trace under `isa_var` is the trace under the default table up to the renumbering.  What may differ:
    code words (memory queries of type Code, previous_code_word) — they must map back opcode by opcode — and what a code HASH
    flows into (the bytecode differs, so its versioned hash does: the deployer's storage slot that holds it, the decommit event)."""
    for k in ("status", "n_cycles"):
        if t_def[k] != t_var[k]:
            return False, "%s: %r != %r" % (k, t_def[k], t_var[k])
    for k in ("mem_off", "log_off", "aux_off", "records"):
        if t_def[k].tobytes() != t_var[k].tobytes():
            return False, "%s differs" % k
    md, mv = t_def["mem"], t_var["mem"]
    if md.shape != mv.shape:
        return False, "mem: shape"
    for f in ("timestamp", "page", "index", "lane", "seq", "meta"):
        if not np.array_equal(md[f], mv[f]):
            return False, "mem.%s differs" % f
    diff = np.flatnonzero((md["value"] != mv["value"]).any(axis=1))
    for j in diff:
        if (int(md["meta"][j]) & K.MQ_TYPE_MASK) != K.MEM_CODE:
            return False, "mem[%d]: a value differs in a query that is no code read" % j
        if not _maps_back(isa_var, mv["value"][j], md["value"][j]):
            return False, "mem[%d]: the code word does not map back to the default table's" % j
    ld, lv = t_def["log"], t_var["log"]
    if ld.shape != lv.shape:
        return False, "log: shape"
    deployer = K.address_bytes(0x8002)
    for j in range(len(ld)):
        if ld[j].tobytes() == lv[j].tobytes():
            continue
        a, b = ld[j].copy(), lv[j].copy()
        if not np.array_equal(a["address"], deployer):
            return False, "log[%d] differs outside the deployer's code-hash slots" % j
        a["read_value"] = b["read_value"] = 0
        a["written_value"] = b["written_value"] = 0
        if a.tobytes() != b.tobytes():
            return False, "log[%d]: more than the code hash differs" % j
    ad, av = t_def["aux"], t_var["aux"]
    if ad.shape != av.shape:
        return False, "aux: shape"
    for j in range(len(ad)):
        if ad[j].tobytes() == av[j].tobytes():
            continue
        if int(ad[j]["type"]) != K.AUX_DECOMMIT or int(av[j]["type"]) != K.AUX_DECOMMIT:
            return False, "aux[%d] differs and is no decommit" % j
        for f in ("lane", "seq", "flag", "a", "b", "c"):
            if ad[j][f] != av[j][f]:
                return False, "aux[%d].%s differs" % (j, f)
    fd, fv = t_def["final_state"].copy(), t_var["final_state"].copy()
    if not _maps_back(isa_var, fv["previous_code_word"], fd["previous_code_word"]):
        return False, "final_state.previous_code_word does not map back"
    fd["previous_code_word"] = fv["previous_code_word"] = 0
    if fd.tobytes() != fv.tobytes():
        return False, "final_state differs"
    return True, ""


def run(backend, wl, lanes=0):
    wl.limits["lanes_per_wave"] = lanes
    b = backend.create_batch(wl)
    b.reset()
    b.run(wl.n_cycles)
    b.sync()
    return b
