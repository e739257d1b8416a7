"""N > 1 path on CPU: two gloo ranks each own a block of instances (no data-path collective), compute
their queue digests, and the final all-gather / all-reduce (era-zk_evm_amd/shard.py) reproduces the
single-process result.  The per-rank compute here is the oracle (no GPU in this suite); on the GPU box
bench.py runs the same reduce over RCCL."""
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


def _worker(rank, world, port, n_total, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from era_zk_evm_amd import capi as K, synth, shard
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    isa = K.Isa()
    from tests._oracle import load_oracle

    orc = load_oracle().open(isa)
    wl_all = synth.make(2, isa, n_instances=n_total)
    first, count = shard.shard_range(n_total, rank, world)
    wl = synth.make(2, isa, n_instances=n_total)
    # this rank's slice of the global batch
    wl.n_instances = count
    wl.states, wl.inner, wl.heaps = wl_all.states[first:first + count], wl_all.inner[first:first + count], wl_all.heaps[first:first + count]
    wl.storage = wl_all.storage[first:first + count]
    wl.code_pages = [(0, count, p, b) for (_, _, p, b) in wl_all.code_pages]
    b = orc.create_batch(wl)
    b.reset(); b.run(wl.n_cycles); b.sync()
    st = b.stats()
    dig = torch.from_numpy(b.commitments().view(np.int64))
    counters = torch.tensor([int(st["cycles"]), int(st["mem_queries"]), int(st["log_queries"]), int(st["aux_events"])], dtype=torch.int64)
    all_dig, total = shard.final_reduce(dig, counters)
    if rank == 0:
        np.save(os.path.join(out_dir, "digests.npy"), all_dig.numpy().view(np.uint64))
        np.save(os.path.join(out_dir, "counters.npy"), total.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [6, 7])
def test_two_rank_gloo_final_reduce(tmp_path, oracle, isa, n_total):
    import torch.multiprocessing as mp
    from era_zk_evm_amd import synth
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_total, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "digests.npy")
    counters = np.load(tmp_path / "counters.npy")
    wl = synth.make(2, isa, n_instances=n_total)
    b = oracle.create_batch(wl)
    b.reset(); b.run(wl.n_cycles); b.sync()
    assert np.array_equal(got, b.commitments())
    st = b.stats()
    assert list(counters) == [int(st["cycles"]), int(st["mem_queries"]), int(st["log_queries"]), int(st["aux_events"])]


def test_shard_range_covers_everything():
    from era_zk_evm_amd import shard
    for n in (1, 7, 64, 4096, 4097):
        for world in (1, 2, 3, 8):
            blocks = [shard.shard_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == n
            for (f0, c0), (f1, _) in zip(blocks, blocks[1:]):
                assert f0 + c0 == f1
