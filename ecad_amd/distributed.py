"""Multi-GPU plumbing: one process per GPU, replicas on the data path, scalars over the wire.

The hot path shards by independent units (image batches; NSGA-II individuals) -- SURVEY 8(e).  Nothing here ever
moves latents, images or caches between GPUs: the only collectives are a barrier, a max-over-ranks of elapsed time,
an all-gather of per-individual fitness (2 float64 each) and a broadcast of the next population's bit matrix
(pop x 1680 bytes).  Backend "nccl" is RCCL on ROCm; the CPU tests run the same code over "gloo".
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist


def force_dist() -> bool:
    """ECAD_FORCE_DIST=1: create the process group even at world size 1, so that the RCCL path (communicator setup, all-gather, broadcast,
    all-reduce, barrier on device buffers) can be exercised on a 1-GPU box.  Without it a single rank runs the collectives as identities."""
    return os.environ.get("ECAD_FORCE_DIST", "0") == "1"


def single_rank_rendezvous_env() -> None:
    """Rendezvous defaults for a forced world-1 group started without torchrun (127.0.0.1: the container hostname may not resolve)."""
    import socket

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
    for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
        os.environ.setdefault(k, v)


def init(backend: str | None = None) -> tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; initialises the process group when world > 1 (or when ECAD_FORCE_DIST=1)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force_dist()) and not dist.is_initialized():
        if world == 1:
            single_rank_rendezvous_env()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kw)
    return rank, world, local_rank


def _world() -> tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _local_only() -> bool:
    """No process group: the collectives below are identities.  A group of ONE rank (ECAD_FORCE_DIST=1) still goes through the backend."""
    return not (dist.is_available() and dist.is_initialized())


def _comm_device() -> torch.device:
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def shard_range(n: int, world: int, rank: int) -> range:
    """Contiguous blocks by ceil division: candidate c lives on rank c // ceil(n / world) (72 over 8 -> 9 per GPU)."""
    per = -(-n // world)
    return range(min(rank * per, n), min((rank + 1) * per, n))


def gather_fitness(local: np.ndarray, n_total: int) -> np.ndarray:
    """All-gather of the per-individual objective rows; `local` is [len(shard_range), n_obj] float64."""
    rank, world = _world()
    local = np.asarray(local, dtype=np.float64)
    if _local_only():
        assert local.shape[0] == n_total
        return local
    per = -(-n_total // world)
    n_obj = local.shape[1]
    dev = _comm_device()
    buf = torch.zeros((per, n_obj), dtype=torch.float64, device=dev)
    buf[: local.shape[0]] = torch.from_numpy(local).to(dev)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    rows = [out[r][: len(shard_range(n_total, world, r))].cpu().numpy() for r in range(world)]
    return np.concatenate(rows, axis=0)


def broadcast_population(bits: np.ndarray | None, shape: tuple[int, int]) -> np.ndarray:
    """Rank 0 owns tell/ask and ships the next [pop, n_bits] boolean matrix (121 KB for 72 x 1680)."""
    rank, world = _world()
    if _local_only():
        return np.asarray(bits, dtype=np.bool_)
    dev = _comm_device()
    t = torch.zeros(shape, dtype=torch.uint8, device=dev)
    if rank == 0:
        t.copy_(torch.from_numpy(np.asarray(bits, dtype=np.uint8)))
    dist.broadcast(t, src=0)
    return t.cpu().numpy().astype(np.bool_)


def broadcast_ints(values, n: int) -> list[int]:
    """Rank 0's `n` integers to everyone (generation header: error flag, generation number, population rows)."""
    rank, world = _world()
    if _local_only():
        return [int(v) for v in values]
    t = torch.zeros(n, dtype=torch.int64, device=_comm_device())
    if rank == 0:
        t.copy_(torch.tensor([int(v) for v in values], dtype=torch.int64))
    dist.broadcast(t, src=0)
    return [int(v) for v in t.cpu().tolist()]


def max_over_ranks(value: float) -> float:
    rank, world = _world()
    if _local_only():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=_comm_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def any_rank_failed(failed: bool) -> bool:
    """MAX all-reduce of a per-rank error flag: every rank calls it (also the one that caught an exception), every rank learns the outcome."""
    rank, world = _world()
    if _local_only():
        return bool(failed)
    t = torch.tensor([1 if failed else 0], dtype=torch.int64, device=_comm_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t.item()))


def barrier() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
