"""Bin sharding over the GPUs of one node and the single gather of the QA table.

The reference's only parallelism is bin-level multiprocessing (checkm/markerGeneFinder.py:59-83);
bins are independent (Z and domZ are per bin), so the MI355X form is: one process per GPU, a
size-balanced static shard of bins per rank, the profile DB replicated, and ONE collective at the
end -- an all_gather of fixed-width QA rows over RCCL/xGMI (a few KB per bin: latency bound).
"""
import os

import numpy as np


def shard_bins(weights, world_size):
    """Greedy longest-first partition. weights[b] = residues x models of bin b. Returns list of index lists."""
    order = sorted(range(len(weights)), key=lambda b: (-weights[b], b))
    loads = [0] * world_size
    shards = [[] for _ in range(world_size)]
    for b in order:
        r = min(range(world_size), key=lambda i: (loads[i], i))
        shards[r].append(b)
        loads[r] += weights[b]
    for s in shards:
        s.sort()
    return shards


QA_WIDTH = 12   # bin index, n_markers, n_sets, hist[6], completeness, contamination, strain heterogeneity


def pack_qa_rows(bin_ids, n_markers, n_sets, hist, comp, cont):
    rows = np.zeros((len(bin_ids), QA_WIDTH), dtype=np.float64)
    rows[:, 0] = bin_ids
    rows[:, 1] = n_markers
    rows[:, 2] = n_sets
    rows[:, 3:9] = np.asarray(hist).reshape(len(bin_ids), 6)
    rows[:, 9] = comp
    rows[:, 10] = cont
    return rows


def emulated():
    """CKM_EMULATE_RANK=R/W: this single process behaves as rank R of W everywhere the product asks for its rank -- it scans rank R's
    shard and does all the all-bins host work of a rank -- but no process group exists and the gather returns the local rows
    (bench.py --emulate-rank: what configs[3] costs one rank, measured on one GPU)."""
    e = os.environ.get("CKM_EMULATE_RANK")
    if not e:
        return None
    r, w = e.split("/")
    return int(r), int(w)


def env_rank():
    emu = emulated()
    if emu is not None:
        return emu[0], int(os.environ.get("LOCAL_RANK", "0")), emu[1]
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend=None):
    """torch.distributed over RCCL ('nccl' on ROCm) when GPUs are used, gloo for the CPU tests."""
    import torch
    import torch.distributed as dist
    if emulated() is not None:
        return None
    rank, local_rank, world = env_rank()
    if world == 1 and not dist.is_initialized():
        return None
    if not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("CKM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("CHECKM_AMD_DEVICE", local_rank)))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist


def collective_device():
    """Device the gather buffers live on: the rank's GPU over RCCL, None (host) over gloo."""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return None


def shutdown():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def agree(ok):
    """Every rank says whether it is well; True only if all are (one all_reduce -- it also is the barrier of the place it stands at).
    A rank that has failed calls agree(False) BEFORE it exits, so that its peers, who meet it here, end too instead of waiting in a
    collective for a rank that is gone."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dev = collective_device()
    if dev is not None:
        t = t.to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.cpu()[0]))


def world_size():
    import torch.distributed as dist
    if emulated() is not None:
        return emulated()[1]
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def gather_qa_rows(rows, max_rows, device=None, even_alone=False):
    """all_gather of per-rank QA rows padded to max_rows; returns the concatenated valid rows (every rank).  A world of one rank returns
    its rows without touching the collective library unless even_alone is set (tests/test_gpu_dist.py pushes a QA buffer through RCCL
    that way: the only execution of the collective a one-GPU box allows)."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size() == 1 and not even_alone):
        return rows
    world = dist.get_world_size()
    buf = torch.full((max_rows, QA_WIDTH), -1.0, dtype=torch.float64)
    if len(rows):
        buf[: len(rows)] = torch.from_numpy(np.ascontiguousarray(rows))
    if device is not None:
        buf = buf.to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    allr = torch.cat(out).cpu().numpy()
    allr = allr[allr[:, 0] >= 0]
    return allr[np.argsort(allr[:, 0], kind="stable")]
