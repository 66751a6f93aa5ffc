"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in CPU tests).  Clips are independent units (asr_core.py:151 transcribes file by file), so
the path shards with no data-path collective: the only exchanges are the one-off weight broadcast from
rank 0 and the gather of the result token ids."""
from __future__ import annotations

import os
from typing import Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import synth
from .config import WhisperDims


def init_process_group(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # TTASR_DIST_BACKEND=gloo: plumbing smoke test with several ranks sharing one GPU (RCCL refuses that)
            backend = os.environ.get("TTASR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n_items for `rank`; the first n_items % world ranks get one extra."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _is_matrix(name: str, shape) -> bool:
    """Tensors the bf16 engine stores in bf16 (every weight matrix except the f32 encoder position table)."""
    return len(shape) >= 2 and name != "model.encoder.embed_positions.weight"


def broadcast_tensors(dims: WhisperDims, src_iter: Optional[Iterable[Tuple[str, np.ndarray]]], device=None,
                      bucket_bytes: int = 256 << 20, bf16_matrices: bool = False) -> Iterator[Tuple[str, object]]:
    """Rank 0 supplies (name, array) in tensor_specs order; every rank yields the same sequence.  Tensors travel in
    ~256 MB buckets (few, large broadcasts: xGMI is point-to-point, so per-call latency and per-link bandwidth, not
    switch fan-out, set the cost).

    RCCL ("nccl") backend: the buckets live in device memory and are handed to the engine as DeviceTensor views - GPU to
    GPU over xGMI, no host staging on the receiving ranks; with `bf16_matrices` (bf16 engines) the weight matrices are
    rounded to bf16 ONCE on rank 0 and travel as bf16 (3.1 GB instead of 6.2 GB for large-v3; every rank, rank 0
    included, loads the same bits).  gloo backend (CPU tests): float32 host buckets, host arrays out."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        yield from src_iter
        return
    from .engine import DeviceTensor
    rank = dist.get_rank()
    on_dev = not (device is None or dist.get_backend() == "gloo")
    dev = torch.device(f"cuda:{device}") if on_dev else torch.device("cpu")
    specs = synth.tensor_specs(dims)
    it = iter(src_iter) if rank == 0 else None

    def dtype_of(k):
        return torch.bfloat16 if (on_dev and bf16_matrices and _is_matrix(specs[k][0], specs[k][1])) else torch.float32

    i = 0
    while i < len(specs):
        dt = dtype_of(i)
        esz = 2 if dt == torch.bfloat16 else 4
        j, n_el = i, 0
        while j < len(specs) and dtype_of(j) == dt and (j == i or (n_el + int(np.prod(specs[j][1]))) * esz <= bucket_bytes):
            n_el += int(np.prod(specs[j][1]))
            j += 1
        if rank == 0:
            flat = torch.empty(n_el, dtype=dt, device=dev)
            off = 0
            for k in range(i, j):
                name, arr = next(it)
                assert name == specs[k][0] and tuple(arr.shape) == tuple(specs[k][1]), (name, specs[k])
                n = int(np.prod(specs[k][1]))
                flat[off:off + n] = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32).ravel()).to(dev).to(dt)
                off += n
        else:
            flat = torch.empty(n_el, dtype=dt, device=dev)
        dist.broadcast(flat, src=0)
        if on_dev:
            torch.cuda.synchronize(dev)   # the engine reads the bucket on its own stream
            off = 0
            for k in range(i, j):
                n = int(np.prod(specs[k][1]))
                yield specs[k][0], DeviceTensor(flat.data_ptr() + off * esz, 1 if dt == torch.bfloat16 else 0, tuple(specs[k][1]))
                off += n
            del flat                      # the consumer has loaded every view (load_weights is synchronous)
        else:
            host = flat.numpy()
            off = 0
            for k in range(i, j):
                n = int(np.prod(specs[k][1]))
                yield specs[k][0], host[off:off + n].reshape(specs[k][1])
                off += n
        i = j


def broadcast_weights(engine, dims: WhisperDims, src_iter, device=None):
    from .config import COMPUTE_BF16
    engine.load_weights(broadcast_tensors(dims, src_iter, device, bf16_matrices=engine.compute_type == COMPUTE_BF16))


def gather_tokens(tokens: Sequence[Sequence[int]], max_len: int, device=None, pad: int = -1) -> np.ndarray:
    """All ranks' token ids as one int32 [world * B][max_len] array (padded with `pad`), rank order."""
    B = len(tokens)
    local = np.full((B, max_len), pad, dtype=np.int32)
    for b, t in enumerate(tokens):
        n = min(len(t), max_len)
        local[b, :n] = t[:n]
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    dev = torch.device("cpu") if device is None or dist.get_backend() == "gloo" else torch.device(f"cuda:{device}")
    mine = torch.from_numpy(local).to(dev)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return torch.cat(out, dim=0).cpu().numpy()


def gather_logits(logits: np.ndarray, device=None) -> np.ndarray:
    """Validation mode: every rank's float32 [rows][vocab] first-step logits as one
    [world][rows][vocab] array on every rank.  Used to prove the broadcast weights are bit-identical everywhere: all
    ranks decode the same probe clip and must produce the same logits."""
    local = np.ascontiguousarray(logits, dtype=np.float32)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local[None]
    dev = torch.device("cpu") if device is None or dist.get_backend() == "gloo" else torch.device(f"cuda:{device}")
    mine = torch.from_numpy(local).to(dev)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return torch.stack(out, dim=0).cpu().numpy()
