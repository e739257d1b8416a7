"""Multi-GPU sharding of a batch.

VM instances share no state (every reference `VmState` owns its oracles by value, mod.rs:167-174),
so ranks own contiguous blocks of instances and run them with NO data-path collective.  The only
exchange is the final one, zkw_reduce_commitments (include/zkw.h): an all-gather of the per-instance queue
digests and an all-reduce of the run counters — at <= 400 KB per rank, xGMI link bandwidth is irrelevant.

`make_comm` builds the communicator of that exchange for a torch.distributed job: RCCL inside libzkw.so when
every rank can create it, otherwise — on ALL ranks, decided together — the library's external transport with the
two collectives served by the job's existing process group (RCCL when its backend is "nccl", gloo on CPU).
"""
import numpy as np
import torch
import torch.distributed as dist

from . import capi as K


def shard_range(n_total, rank, world):
    """contiguous block [first, first+count) of rank `rank`; blocks differ by at most one instance"""
    base, rem = divmod(n_total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def _pg_device(device):
    return torch.device("cpu") if dist.get_backend() == "gloo" else (device if device is not None else torch.device("cuda", torch.cuda.current_device()))


def make_comm(backend, rank, world, device=None, prefer_rccl=True):
    """-> (capi.Comm, description of the transport that will carry zkw_reduce_commitments).

    world == 1 without a process group: a one-rank external communicator (no transport at all).  Otherwise the ranks
    first try the library's own RCCL communicator: every rank probes locally that librccl loads (zkw_comm_probe: dlopen
    + dlsym, no side effects), the ranks agree on that with one all-reduce, and only then rank 0 alone makes the id
    (ncclGetUniqueId starts a bootstrap root — a listening socket and a thread — which only the id in use should own), the
    id is broadcast and zkw_comm_create_rccl — a collective — runs everywhere, its outcome agreed on with a second all-reduce;
    if ANY rank failed at either step every rank falls back to zkw_comm_create_external with all-gather / all-reduce
    callbacks over the process group that is already up."""
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return K.Comm.external(backend, 0, 1), "single rank (no transport)"
    dev = _pg_device(device)
    comm, why = None, ""
    if prefer_rccl:
        # Step 1, local only and free of side effects: can this rank load librccl at all?  Agreed on BEFORE the collective
        # init — ncclCommInitRank is itself a collective, so a rank that cannot even reach it would leave the others blocked
        # in zkw_comm_create_rccl until the NCCL timeout instead of falling back with them.
        can = False
        try:
            can = K.Comm.probe(backend)
        except K.ZkwError as e:  # e.g. librccl cannot be loaded on this rank
            why = str(e)
        able = torch.tensor([1 if can else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(able, op=dist.ReduceOp.MIN)
        if int(able.item()) == 1:
            # Step 2: rank 0 makes the id (the only bootstrap root of the job), hands it to everyone, then the collective
            # init, then the ranks agree on its outcome
            my_id = None
            if rank == 0:
                try:
                    my_id = K.Comm.unique_id(backend)
                except K.ZkwError as e:
                    why = str(e)
            ids = [my_id]
            dist.broadcast_object_list(ids, src=0, device=dev)
            try:
                if ids[0] is None:
                    raise K.ZkwError("rank 0 could not make an RCCL id")
                comm = K.Comm.rccl(backend, rank, world, ids[0])
            except K.ZkwError as e:
                why = why or str(e)
            ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int64, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                return comm, "rccl (libzkw.so's own communicator)"
            if comm is not None:  # some other rank failed: everyone takes the fallback
                comm.close()
                comm = None

    def allgather(send):
        mine = torch.from_numpy(np.ascontiguousarray(send)).to(dev)
        outs = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(outs, mine)
        return torch.cat(outs).cpu().numpy()

    def allreduce_sum(a):
        x = torch.from_numpy(a.astype(np.int64)).to(dev)
        dist.all_reduce(x, op=dist.ReduceOp.SUM)
        return x.cpu().numpy().astype(np.uint64)

    desc = "external over torch.distributed (%s)" % dist.get_backend()
    if prefer_rccl:
        desc += "; RCCL communicator unavailable" + (": " + why[:120] if why else " on another rank")
    return K.Comm.external(backend, rank, world, allgather, allreduce_sum), desc
