"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl"
is RCCL on ROCm; "gloo" in CPU tests).  The read stream shards with no
data-path collective (ReadAnalyzer::operator() is const over a frozen index,
ReadAnalyzer.hpp:39, main.cpp:193); the only exchange is one all-reduce of the
per-gene assigned-read counts (<= 65 536 x 8 B) per run."""
import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend, device=None):
    rank, _, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if device is not None and backend == "nccl":
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def batch_owner(batch_index, world):
    """batch i of the input stream is classified by rank i mod world (same rule as `shark --gpus N`)"""
    return batch_index % world


def shard_batches(n_reads, batch_size, rank, world):
    """[(first, last)) read ranges owned by `rank`"""
    out = []
    for i, first in enumerate(range(0, n_reads, batch_size)):
        if batch_owner(i, world) == rank:
            out.append((first, min(n_reads, first + batch_size)))
    return out


def backend_name():
    """"nccl" (= RCCL on ROCm) unless SHARK_DIST_BACKEND overrides it (CPU tests / single-GPU dry runs use gloo)"""
    return os.environ.get("SHARK_DIST_BACKEND", "nccl")


def _reduce(t, op):
    if dist.get_backend() == "gloo" and t.is_cuda:   # gloo dry run on a GPU box: reduce a host copy
        c = t.cpu()
        dist.all_reduce(c, op=op)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op)
    return t


def allreduce_sum_(t):
    if dist.is_initialized() and dist.get_world_size() > 1:
        _reduce(t, dist.ReduceOp.SUM)
    return t


def max_over_ranks(seconds, device):
    if dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([seconds], dtype=torch.float64, device=device)
        _reduce(t, dist.ReduceOp.MAX)
        return float(t.item())
    return seconds


def gather_rows(row, device):
    """every rank's list of floats -> list of lists, on every rank (the job's own channel; world 1: [row])"""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [list(row)]
    t = torch.tensor(list(row), dtype=torch.float64, device=device)
    if dist.get_backend() == "gloo" and t.is_cuda:
        t = t.cpu()
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(x) for x in o.cpu().tolist()] for o in out]


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def finalize():
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
