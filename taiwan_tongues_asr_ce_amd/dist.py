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


def force_collectives() -> bool:
    """TTASR_DIST_FORCE=1: run the process-group code paths (RCCL weight broadcast into device buckets, token / logits
    all-gathers, barriers) even at WORLD_SIZE = 1 - RCCL accepts a single rank per device, so a 1-GPU box executes exactly
    the transport calls an 8-GPU node will (tests/test_gpu_weights_and_launch.py, `bench.py --gpus 1`)."""
    return os.environ.get("TTASR_DIST_FORCE", "") not in ("", "0")


def resolve_backend(requested: Optional[str], local_world: int, n_devices: int, cuda: bool) -> Tuple[str, Optional[str]]:
    """(backend, note) for a group of `local_world` ranks on this node seeing `n_devices` GPUs.  RCCL refuses two ranks on one
    device ("ncclInvalidUsage: Duplicate GPU detected" - what `torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` died
    with on a 1-GPU box in round 5): ranks > devices talk over gloo and the note says so (a plumbing run, never a scaling
    number); an EXPLICIT request for nccl in that situation is a one-line error instead of RCCL's."""
    want = requested or os.environ.get("TTASR_DIST_BACKEND") or ("nccl" if cuda else "gloo")
    if want == "nccl" and cuda and local_world > n_devices:
        if requested == "nccl" or os.environ.get("TTASR_DIST_BACKEND") == "nccl":
            raise RuntimeError(f"{local_world} ranks on this node but {n_devices} visible GPU(s): RCCL needs one device per rank "
                               "(use TTASR_DIST_BACKEND=gloo for a shared-GPU plumbing run)")
        return "gloo", (f"{local_world} ranks share {n_devices} GPU(s): RCCL refuses two ranks per device, so the group runs over "
                        "gloo - plumbing only, not a scaling measurement")
    return want, None


def host_thread_plan(local_world: int, n_cpus: int) -> int:
    """Host threads per rank for torch / OpenMP work when `local_world` ranks share a host: every rank drives a ~45 000-launch
    decode chain per step from ONE thread and must not be crowded out by 8 x (all cores) OpenMP pools (risk
    (ii)).  An equal share of the cores, at least 1, at most 32 (torch's small ops get slower beyond that: conftest.py)."""
    return max(1, min(32, n_cpus // max(1, local_world)))


def _parse_cpulist(text: str) -> List[int]:
    cpus: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def numa_cpus_of_pci(pci_bdf: str, sysfs: str = "/sys") -> Optional[List[int]]:
    """CPUs of the NUMA node the PCI device `pci_bdf` ("0000:c1:00.0") hangs off, from sysfs; None when the kernel reports no
    node (-1: single-socket / virtualised hosts) or the files are missing."""
    try:
        with open(os.path.join(sysfs, "bus", "pci", "devices", pci_bdf, "numa_node")) as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")) as f:
            cpus = _parse_cpulist(f.read())
        return cpus or None
    except (OSError, ValueError):
        return None


def bind_rank_to_gpu_numa(local: int, local_world: int, sysfs: str = "/sys", pci_bdf: Optional[str] = None) -> dict:
    """Several ranks on one host (world > 1): cap this rank's torch / OpenMP threads at its share of the cores and bind the
    calling (= kernel-launching) thread to the CPUs of the NUMA node of its GPU, so that eight launch threads neither migrate
    across sockets nor fight eight full-size OpenMP pools.  Best effort, never fatal; returns what was done (bench.py prints it in
    config.host_binding).  `pci_bdf` (tests) overrides the lookup through torch.cuda.get_device_properties."""
    info: dict = {"threads": None, "numa_cpus": None}
    n_cpus = os.cpu_count() or 1
    threads = host_thread_plan(local_world, n_cpus)
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    try:
        torch.set_num_threads(min(threads, int(os.environ["OMP_NUM_THREADS"])))
        info["threads"] = torch.get_num_threads()
    except Exception as ex:   # pragma: no cover
        info["threads_error"] = str(ex)
    try:
        if pci_bdf is None and torch.cuda.is_available():
            pr = torch.cuda.get_device_properties(local)
            pci_bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        cpus = numa_cpus_of_pci(pci_bdf, sysfs) if pci_bdf else None
        if cpus and hasattr(os, "sched_setaffinity"):
            allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
            if allowed:
                os.sched_setaffinity(0, allowed)
                info["numa_cpus"] = len(allowed)
                info["pci"] = pci_bdf
    except Exception as ex:
        info["numa_error"] = str(ex)
    return info


HOST_BINDING: dict = {}     # what init_process_group did for this rank (bench.py: config.host_binding)
BACKEND_NOTE: Optional[str] = None


def init_process_group(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; no-op for a single process unless force_collectives()."""
    global BACKEND_NOTE
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world > 1 and not HOST_BINDING:
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        HOST_BINDING.update(bind_rank_to_gpu_numa(local % max(n_dev, 1), local_world))
    if (world > 1 or force_collectives()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "MASTER_PORT" not in os.environ:
            if world > 1:   # every rank would pick a different free port and the rendezvous would hang until its timeout
                raise RuntimeError("MASTER_PORT is not set (WORLD_SIZE=%d): launch through torch.distributed.run / bench.py --gpus N" % world)
            import socket       # forced single-rank group started from a bare shell
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s.getsockname()[1])
            s.close()
        # TTASR_DIST_BACKEND=gloo: plumbing smoke test with several ranks sharing one GPU; ranks > visible devices fall back to it
        # by themselves (resolve_backend) - whatever launcher started them
        cuda = torch.cuda.is_available()
        backend, BACKEND_NOTE = resolve_backend(backend, local_world, torch.cuda.device_count() if cuda else 0, cuda)
        if BACKEND_NOTE and rank == 0:
            import sys
            print(f"[ttasr.dist] {BACKEND_NOTE}", file=sys.stderr, flush=True)
        kw = {}
        if backend == "nccl":
            local = local % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)   # binds the communicator to this rank's GPU at creation (eager init)
        # a peer that died must not leave this rank parked in a collective for the backend's default 10-30 min: the group's
        # timeout ends it (RCCL: the watchdog aborts the process; gloo: the collective raises) - bench.py's launcher and
        # torch.distributed.run additionally stop the other ranks the moment one exits non-zero
        import datetime
        kw["timeout"] = datetime.timedelta(seconds=int(os.environ.get("TTASR_DIST_TIMEOUT_S", "600")))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def barrier(device: Optional[int] = None):
    """dist.barrier() that names the device under RCCL (otherwise the backend guesses it from the rank)."""
    if not dist.is_initialized():
        return
    if dist.get_backend() == "nccl" and device is not None:
        dist.barrier(device_ids=[device])
    else:
        dist.barrier()


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n_items for `rank`; the first n_items % world ranks get one extra."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _is_matrix(name: str, shape) -> bool:
    """Tensors the bf16 engine stores in bf16 (every weight matrix except the f32 encoder position table)."""
    return len(shape) >= 2 and name != "model.encoder.embed_positions.weight"


def bucket_plan(sizes: Sequence[int], kinds: Sequence[int], bucket_bytes: int, esz: Sequence[int]) -> List[List[int]]:
    """Tensors 0..n-1 (element counts `sizes`, dtype class `kinds[i]` with esz[kind] bytes per element) -> buckets of indices.
    One bucket stays open PER dtype class and closes when the next tensor of its class would overflow bucket_bytes, so the
    interleaving of bf16 matrices with f32 biases / LayerNorm vectors in the state-dict order does not cut the buckets short
    (~800 one-tensor broadcasts for large-v3 became 14).  Buckets are returned in the order in which their
    LAST tensor appears: rank 0, which receives the tensors in state-dict order, can send each bucket the moment it is complete
    with at most one open bucket per class in memory."""
    open_b = {}
    done: List[List[int]] = []
    fill = {}
    for i, (n, k) in enumerate(zip(sizes, kinds)):
        if k in open_b and (fill[k] + n) * esz[k] > bucket_bytes:
            done.append(open_b.pop(k))
        if k not in open_b:
            open_b[k], fill[k] = [], 0
        open_b[k].append(i)
        fill[k] += n
    done.extend(open_b.values())
    done.sort(key=lambda b: b[-1])
    return done


def broadcast_tensors(dims: WhisperDims, src_iter: Optional[Iterable[Tuple[str, np.ndarray]]], device=None,
                      bucket_bytes: int = 256 << 20, bf16_matrices: bool = False, matrix_dtype: Optional[str] = None,
                      host_16bit: bool = False) -> Iterator[Tuple[str, object]]:
    """Rank 0 supplies (name, array) in tensor_specs order; every rank yields every tensor once (bucket order - the engine's
    intake is keyed by name).  Tensors travel in ~256 MB buckets, one open bucket per dtype (few, large broadcasts: xGMI is
    point-to-point, so per-call latency and per-link bandwidth, not switch fan-out, set the cost).

    RCCL ("nccl") backend: the buckets live in device memory and are handed to the engine as DeviceTensor views - GPU to
    GPU over xGMI, no host staging on the receiving ranks; with matrix_dtype "bf16" / "f16" (16-bit engines;
    `bf16_matrices=True` is the older spelling of "bf16") the weight matrices are rounded to that type ONCE on rank 0 and
    travel as 16-bit words (3.1 GB instead of 6.2 GB for large-v3; every rank, rank 0 included, loads the same bits).
    gloo backend (CPU tests): float32 host buckets, host arrays out; `host_16bit=True` makes the gloo path use the SAME
    per-dtype bucket plan and 16-bit travel as the RCCL path (values come out as float32 arrays holding the rounded numbers), so
    that the multi-dtype send / receive order is exercised by the CPU multi-rank tests."""
    if matrix_dtype is None and bf16_matrices:
        matrix_dtype = "bf16"
    if matrix_dtype not in (None, "bf16", "f16"):
        raise ValueError(f"matrix_dtype={matrix_dtype!r}")
    if not dist.is_initialized():
        yield from src_iter
        return
    from .engine import DeviceTensor
    rank = dist.get_rank()
    on_dev = not (device is None or dist.get_backend() == "gloo")
    dev = torch.device(f"cuda:{device}") if on_dev else torch.device("cpu")
    specs = synth.tensor_specs(dims)
    sizes = [int(np.prod(sp[1])) for sp in specs]
    kinds = [1 if ((on_dev or host_16bit) and matrix_dtype and _is_matrix(sp[0], sp[1])) else 0 for sp in specs]
    esz = (4, 2)
    dts = (torch.float32, torch.float16 if matrix_dtype == "f16" else torch.bfloat16)
    codes = (0, 2 if matrix_dtype == "f16" else 1)     # TTASR_DTYPE_F32 / _BF16 / _F16
    plan = bucket_plan(sizes, kinds, bucket_bytes, esz)
    where = {}                                  # tensor index -> (bucket, element offset)
    for b, idx in enumerate(plan):
        off = 0
        for k in idx:
            where[k] = (b, off)
            off += sizes[k]
    flats = {}

    def flat_of(b):
        if b not in flats:
            flats[b] = torch.empty(sum(sizes[k] for k in plan[b]), dtype=dts[kinds[plan[b][0]]], device=dev)
        return flats[b]

    def emit(b):
        flat = flats.pop(b)
        dist.broadcast(flat, src=0)
        k0 = kinds[plan[b][0]]
        if on_dev:
            torch.cuda.synchronize(dev)       # the engine reads the bucket on its own stream
            for k in plan[b]:
                yield specs[k][0], DeviceTensor(flat.data_ptr() + where[k][1] * esz[k0], codes[k0], tuple(specs[k][1]))
        else:
            host = flat.float().numpy() if k0 else flat.numpy()
            for k in plan[b]:
                yield specs[k][0], host[where[k][1]:where[k][1] + sizes[k]].reshape(specs[k][1])
        del flat                              # the consumer has loaded every view (load_weights is synchronous)

    if rank == 0:
        closes = {idx[-1]: b for b, idx in enumerate(plan)}
        for k, (name, arr) in enumerate(src_iter):
            assert name == specs[k][0] and tuple(arr.shape) == tuple(specs[k][1]), (name, specs[k])
            b, off = where[k]
            flat_of(b)[off:off + sizes[k]] = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32).ravel()).to(dev).to(dts[kinds[k]])
            if k in closes:
                yield from emit(closes[k])
    else:
        for b in range(len(plan)):
            flat_of(b)
            yield from emit(b)


def broadcast_weights(engine, dims: WhisperDims, src_iter, device=None):
    from .config import COMPUTE_BF16, COMPUTE_F16
    md = {COMPUTE_BF16: "bf16", COMPUTE_F16: "f16"}.get(engine.compute_type)
    engine.load_weights(broadcast_tensors(dims, src_iter, device, matrix_dtype=md))


def gather_tokens(tokens: Sequence[Sequence[int]], max_len: int, device=None, pad: int = -1) -> np.ndarray:
    """All ranks' token ids as one int32 [world * B][max_len] array (padded with `pad`), rank order."""
    B = len(tokens)
    local = np.full((B, max_len), pad, dtype=np.int32)
    for b, t in enumerate(tokens):
        n = min(len(t), max_len)
        local[b, :n] = t[:n]
    if not dist.is_initialized():
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
    if not dist.is_initialized():
        return local[None]
    dev = torch.device("cpu") if device is None or dist.get_backend() == "gloo" else torch.device(f"cuda:{device}")
    mine = torch.from_numpy(local).to(dev)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return torch.stack(out, dim=0).cpu().numpy()
