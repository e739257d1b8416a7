"""N > 1 path on CPU: two gloo ranks each own a block of instances (no data-path collective), compute
their queue digests, and the final all-gather / all-reduce reproduces the single-process result through the PRODUCT's own
entry point zkw_reduce_commitments (include/zkw.h) — the product sources built for the CPU by tests/emu compute
each rank's shard and its commitments.  The communicator comes from era-zk_evm_amd/shard.py `make_comm`, exactly as in
bench.py: it first tries the library's RCCL communicator, which the CPU build does not have, so every rank takes the
fallback the GPU job would take if its second communicator failed — the external transport over the process group that
is already up (gloo here, RCCL through torch there).  On the GPU box the RCCL path itself runs in
tests/test_gpu_parity.py and bench.py."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_covers_everything():
    from era_zk_evm_amd import shard
    for n in (1, 7, 64, 4096, 4097):
        for world in (1, 2, 3, 8):
            blocks = [shard.shard_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == n
            for (f0, c0), (f1, _) in zip(blocks, blocks[1:]):
                assert f0 + c0 == f1


def _worker_capi(rank, world, port, n_total, out_dir):
    """each rank: the emulated product (kernel logic + host runtime) on its shard, then zkw_reduce_commitments"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import torch
    import torch.distributed as dist
    import era_zk_evm_amd  # noqa: F401
    from era_zk_evm_amd import capi as K, synth, shard
    import build_emu
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    isa = K.Isa()
    prod = K.Backend(build_emu.build(), "zkw_").open(isa)
    wl_all = synth.make(2, isa, n_instances=n_total)
    first, count = shard.shard_range(n_total, rank, world)
    wl = synth.make(2, isa, n_instances=n_total)
    wl.n_instances = count
    wl.states, wl.inner, wl.heaps = wl_all.states[first:first + count], wl_all.inner[first:first + count], wl_all.heaps[first:first + count]
    wl.storage = wl_all.storage[first:first + count]
    wl.code_pages = [(0, count, p, b) for (_, _, p, b) in wl_all.code_pages]
    batches = [prod.create_batch(wl) for _ in range(2)]  # two batches per rank, as the fused bench groups are
    prod.step_many(batches, wl.n_cycles, 7)

    comm, transport = shard.make_comm(prod, rank, world)  # RCCL is not part of the CPU build: the fallback every rank agrees on
    assert transport.startswith("external over torch.distributed (gloo)"), transport
    gathered, n_max, sizes, total = comm.reduce(batches, 0b101, want_total=True)  # memory + decommit queues
    # a rank that changes its shard size on its own is refused (the other rank would not take part in a size exchange) ...
    if rank == 1:
        wl2 = synth.make(1, isa, n_instances=count + 1)
        odd = [prod.create_batch(wl2)]
        prod.step_many(odd, wl2.n_cycles, 7)
        try:
            comm.reduce(odd, 0b001, gathered=np.zeros((world, 1, 64, 1, 4), dtype="<u8"))
            refused = False
        except K.ZkwError as e:
            refused = "zkw_comm_exchange_sizes" in str(e)
        assert refused
    # ... the same sizes again are fine, from both ranks
    g2, _, _, _ = comm.reduce(batches, 0b100)
    assert np.array_equal(g2[:, :, :, 0], gathered[:, :, :, 1])
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), gathered)
        np.save(os.path.join(out_dir, "meta.npy"), np.array([n_max] + sizes + [int(total["cycles"]), int(total["mem_queries"]), int(total["log_queries"]),
                                                                       int(total["aux_events"])], dtype=np.int64))
    comm.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [6, 7])
def test_two_rank_reduce_commitments_entry(tmp_path, oracle, isa, n_total):
    """zkw_reduce_commitments (the C-ABI entry) over gloo: gathered digests in rank order, ragged shards padded"""
    import torch.multiprocessing as mp
    from era_zk_evm_amd import synth, shard
    port = _free_port()
    mp.spawn(_worker_capi, args=(2, port, n_total, str(tmp_path)), nprocs=2, join=True)
    gathered = np.load(tmp_path / "gathered.npy")  # [world][batches][n_max][2 queues][4]
    meta = np.load(tmp_path / "meta.npy")
    wl = synth.make(2, isa, n_instances=n_total)
    b = oracle.create_batch(wl)
    b.reset(); b.run(wl.n_cycles); b.sync()
    want = b.commitments()  # [n][3][4]
    sizes = [shard.shard_range(n_total, r, 2)[1] for r in range(2)]
    assert int(meta[0]) == max(sizes) and list(meta[1:3]) == sizes
    assert gathered.shape == (2, 2, max(sizes), 2, 4)
    for r in range(2):
        first = shard.shard_range(n_total, r, 2)[0]
        for j in range(2):
            assert np.array_equal(gathered[r, j, :sizes[r], 0], want[first:first + sizes[r], 0])  # memory queue
            assert np.array_equal(gathered[r, j, :sizes[r], 1], want[first:first + sizes[r], 2])  # decommit queue
            assert not gathered[r, j, sizes[r]:].any()  # padding rows
    st = b.stats()
    assert list(meta[3:]) == [2 * int(st["cycles"]), 2 * int(st["mem_queries"]), 2 * int(st["log_queries"]), 2 * int(st["aux_events"])]


# ---------------------------------------------------------------------------------------------------------------------
# bench.py's own N > 1 orchestration — launches, overlap over "streams", restore between uses, zkw_reduce_commitments per
# launch, barriers, the MAX / SUM all-reduces of the line — on the CPU build of the product under gloo
# (ZKW_BENCH_BACKEND=emu).  The 1 -> 8 GPU curve itself is unmeasured here: this covers the code path, not its speed.
# ---------------------------------------------------------------------------------------------------------------------
def _worker_bench(rank, world, port, argv, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      ZKW_BENCH_BACKEND="emu", ZKW_BENCH_DUMP_GATHERED=os.path.join(out_dir, "gathered.npy"))
    sys.path.insert(0, ROOT)
    sys.argv = ["bench.py", "--gpus", str(world)] + argv
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    if rank == 0:
        sys.stdout = open(os.path.join(out_dir, "stdout.txt"), "w")
    bench.main()
    sys.stdout.flush()


def _bench_flow(tmp_path, oracle, isa, world, argv, n_inst, fuse, groups):
    import json
    import torch.multiprocessing as mp
    import importlib.util
    port = _free_port()
    mp.spawn(_worker_bench, args=(world, port, argv, str(tmp_path)), nprocs=world, join=True)
    lines = [ln for ln in open(tmp_path / "stdout.txt").read().splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # ONE JSON line, from rank 0
    j = json.loads(lines[0])
    assert j["n_gpus"] == world and j["scaling"] == "weak" and j["config"]["backend"] == "emu"
    assert j["config"]["collective"].startswith("external over torch.distributed (gloo)"), j["config"]["collective"]
    assert j["checked"]["batches"] == world * groups * fuse and j["checked"]["instances_failed"] == 0
    assert j["checked"]["cycles_executed"] == world * groups * fuse * n_inst * 256
    assert abs(j["value"] - world * n_inst * 256 / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-6  # whole-job aggregate over all ranks
    # the digests the last full launch of group 0 gathered: every rank's shard (its own seed), in rank order
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    a = bench.parse_args(["--gpus", str(world)] + argv)
    gathered = np.load(tmp_path / "gathered.npy")  # [world][fuse][instances][2 queues][4]
    assert gathered.shape == (world, fuse, n_inst, 2, 4)
    for r in range(world):
        wl = bench.make_workload(a, isa, r)
        b = oracle.create_batch(wl)
        b.reset(); b.run(wl.n_cycles); b.sync()
        want = b.commitments()
        for k in range(fuse):
            assert np.array_equal(gathered[r, k, :, 0], want[:, 0]) and np.array_equal(gathered[r, k, :, 1], want[:, 2]), (r, k)
        b.destroy()
    return j


def test_bench_flow_world2(tmp_path, oracle, isa):
    """two groups in flight, a partial last launch, the restore of a reused group on its side 'stream'"""
    argv = ["--steps", "5", "--warmup", "2", "--fuse", "2", "--streams", "2", "--instances", "5", "--commit-mask", "5", "--no-cpu-baseline", "--min-warmup-s", "0"]
    j = _bench_flow(tmp_path, oracle, isa, 2, argv, 5, 2, 2)
    assert j["config"]["cycle_kernel_launches"] == 3 and j["config"]["restores_in_timed_region"] == 1


def test_bench_flow_world8(tmp_path, oracle, isa):
    """the driver's N = 8 launch shape: one rank per GPU under torch.distributed, one fused group per rank and launch"""
    argv = ["--steps", "4", "--warmup", "2", "--fuse", "2", "--streams", "2", "--instances", "3", "--commit-mask", "5", "--no-cpu-baseline", "--min-warmup-s", "0"]
    j = _bench_flow(tmp_path, oracle, isa, 8, argv, 3, 2, 2)
    assert j["config"]["cycle_kernel_launches"] == 2 and j["config"]["restores_in_timed_region"] == 0


def test_bench_flow_restore_every_step_region(tmp_path, oracle, isa):
    """the further regions of the line (--repeats: value_min / median / max, then the same K steps with the restore in front of
    every use) with two groups in flight and the restore on the side streams: the every-step region must start from restored
    groups although the regions before it leave their last uses unrestored (the default command `python bench.py` stopped on
    limits.max_cycles there; no CPU test ran a further region)"""
    argv = ["--steps", "8", "--warmup", "2", "--fuse", "2", "--streams", "2", "--instances", "3", "--commit-mask", "5", "--no-cpu-baseline", "--min-warmup-s", "0",
            "--repeats", "1", "--no-other-configs"]
    j = _bench_flow(tmp_path, oracle, isa, 2, argv, 3, 2, 2)
    assert j["timed_regions"] == 2 and j["restore_in_front_of_every_step"]["value"] > 0


def test_bench_host_legs_on_the_emulation_build(tmp_path, isa):
    """bench.py's `delivered` and `upload` legs — groups of batches delivered into the pinned ring behind their runs and
    replayed on host threads, a reused group restored only behind its delivery; fresh inputs restaged on a side 'stream' for
    every step, both the copying and the in-place form — run end to end on the CPU build (the code path, not its speed)"""
    import json
    import subprocess
    env = dict(os.environ, ZKW_BENCH_BACKEND="emu", ZKW_BENCH_HOST_THREADS="3", PYTHONPATH=ROOT)
    argv = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--fuse", "4", "--streams", "1", "--instances", "3", "--commit-mask", "4",
            "--no-cpu-baseline", "--min-warmup-s", "0", "--no-other-configs", "--host-legs", "--repeats", "0"]
    r = subprocess.run(argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    d, u, ui = j["delivered"], j["upload"], j["upload_in_place"]
    assert "error" not in d and "error" not in u, (d, u)
    assert d["steps"] >= 20 and d["cycles_delivered"] == d["steps"] * 3 * 256 and d["host_threads"] == 3 and 60 < d["bytes_per_cycle"] < 400
    assert u["steps"] >= 20 and u["bytes_per_step"] == 3 * (680 + 32 * 256) and not u["in_place"] and ui["in_place"]
    # the delivered leg ran in both link formats: the smallest one, and with the read values left on the link (more bytes, less host work)
    alt = d["with_read_values_on_the_link"]
    assert d["link_flags"] == 31 and alt["link_flags"] == 30 and alt["bytes_per_cycle"] > d["bytes_per_cycle"] and d["best_cycles_per_s"] >= d["cycles_per_s"]
    assert d["bound_by"].startswith(("host replay", "link"))
    e = j["end_to_end"]
    assert "error" not in e, e
    for form in ("copying", "in_place"):  # fresh inputs, run, delivery and replay in one pipeline
        assert e[form]["steps"] >= 20 and e[form]["cycles_delivered"] == e[form]["steps"] * 3 * 256 and e[form]["in_place"] == (form == "in_place")
    assert j["checked"]["instances_failed"] == 0
