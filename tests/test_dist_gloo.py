"""world_size-2 gloo run of the data-parallel plumbing (weight broadcast, token gather, sharding)."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from taiwan_tongues_asr_ce_amd import synth
    from taiwan_tongues_asr_ce_amd.config import PRESETS
    from taiwan_tongues_asr_ce_amd.dist import broadcast_tensors, gather_logits, gather_tokens, init_process_group, shard_range
    r, w, _ = init_process_group("gloo")
    dims = PRESETS["micro"]
    src = synth.iter_weights(dims) if r == 0 else None
    got = dict(broadcast_tensors(dims, src, bucket_bytes=1 << 20))
    ref = synth.state_dict(dims)
    ok = set(got) == set(ref) and all(np.array_equal(got[k], ref[k]) for k in ref)
    lo, hi = shard_range(5, r, w)
    toks = [[100 * c + j for j in range(c + 1)] for c in range(lo, hi)]
    # shards are ragged (3 + 2 clips): pad to the largest shard so all_gather shapes agree
    n_max = max(shard_range(5, rr, w)[1] - shard_range(5, rr, w)[0] for rr in range(w))
    toks += [[]] * (n_max - len(toks))
    allt = gather_tokens(toks, 6)
    lg = gather_logits(np.full((2, 7), float(r), np.float32))
    ok = ok and lg.shape == (w, 2, 7) and all(float(lg[rr].min()) == float(lg[rr].max()) == rr for rr in range(w))
    q.put((r, ok, allt.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == res[1][2]
    allt = np.array(res[0][2])
    assert allt.shape == (6, 6)
    assert allt[0].tolist() == [0, -1, -1, -1, -1, -1] and allt[2].tolist() == [200, 201, 202, -1, -1, -1]
    assert allt[3].tolist() == [300, 301, 302, 303, -1, -1] and allt[5].tolist() == [-1] * 6


class _Seg:
    def __init__(self, text):
        self.text = text


class _EchoModel:
    """Model double: 'transcribes' a clip into its sample count, so every rank's share is recognisable."""

    def transcribe(self, audio, **kw):
        return iter([_Seg(f"長度{len(audio)}")]), None


def _folder_worker(rank, world, port, folder, out_json, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from taiwan_tongues_asr_ce_amd import batch_cli
    from taiwan_tongues_asr_ce_amd.dist import init_process_group
    r, w, _ = init_process_group("gloo")
    seen = []
    final = batch_cli.process_audio_folder(folder, model=_EchoModel(), output_json=out_json, rank=r, world=w,
                                           load_audio=lambda p: (seen.append(os.path.basename(p)), np.zeros(int(os.path.basename(p)[1:3]) * 10, np.float32))[1],
                                           log=lambda *_: None)
    q.put((r, seen, [d["audio_file"] for d in final["detailed_results"]], final["summary"]["total_files"]))
    dist.barrier()
    dist.destroy_process_group()


def test_folder_tool_shards_files_across_ranks(tmp_path):
    """asr_core-equivalent folder tool under 2 ranks: files (never windows) are dealt round-robin, every rank ends
    with the complete, identically ordered result list, rank 0 alone writes the summary JSON."""
    import json
    folder = tmp_path / "audio"
    folder.mkdir()
    names = [f"c{n:02d}.wav" for n in (11, 12, 13, 14, 15)]
    for n in names:
        (folder / n).write_bytes(b"")
    out_json = str(tmp_path / "summary.json")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_folder_worker, args=(r, 2, port, str(folder), out_json, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == names[0::2] and res[1][1] == names[1::2]          # who transcribed what
    assert res[0][2] == res[1][2] == names and res[0][3] == 5             # merged view, original order
    on_disk = json.load(open(out_json, encoding="utf-8"))
    assert [d["asr_result"] for d in on_disk["detailed_results"]] == [f"長度{n}0" for n in (11, 12, 13, 14, 15)]
    for n in names:
        assert (folder / (n[:-4] + "_asr.txt")).exists()


def test_bucket_plan_keeps_one_open_bucket_per_dtype():
    """the state-dict order interleaves bf16 matrices with f32 vectors; buckets are per dtype, few and large,
    every tensor lands in exactly one, and the order of the buckets is the order in which rank 0 completes them."""
    from taiwan_tongues_asr_ce_amd import synth
    from taiwan_tongues_asr_ce_amd.config import PRESETS
    from taiwan_tongues_asr_ce_amd.dist import _is_matrix, bucket_plan
    specs = synth.tensor_specs(PRESETS["large-v3"])
    sizes = [int(np.prod(s[1])) for s in specs]
    kinds = [1 if _is_matrix(s[0], s[1]) else 0 for s in specs]
    plan = bucket_plan(sizes, kinds, 256 << 20, (4, 2))
    assert sorted(k for b in plan for k in b) == list(range(len(specs)))
    assert len(plan) <= 16                                               # was ~800 when a bucket ended at every dtype change
    for b in plan:
        assert len({kinds[k] for k in b}) == 1 and b == sorted(b)
        assert sum(sizes[k] for k in b) * (2 if kinds[b[0]] else 4) <= (256 << 20) or len(b) == 1
    assert [b[-1] for b in plan] == sorted(b[-1] for b in plan)
    # one dtype (gloo / f32 engines): plain consecutive runs
    plan1 = bucket_plan(sizes[:40], [0] * 40, 1 << 20, (4, 2))
    assert [k for b in plan1 for k in b] == list(range(40))


def _forced_single_rank(q):
    os.environ.update(TTASR_DIST_FORCE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    os.environ.pop("MASTER_PORT", None)
    import torch.distributed as dist
    from taiwan_tongues_asr_ce_amd import synth
    from taiwan_tongues_asr_ce_amd.config import PRESETS
    from taiwan_tongues_asr_ce_amd.dist import barrier, broadcast_tensors, gather_logits, gather_tokens, init_process_group
    r, w, _ = init_process_group("gloo")
    dims = PRESETS["micro"]
    got = dict(broadcast_tensors(dims, synth.iter_weights(dims), bucket_bytes=1 << 18))
    ref = synth.state_dict(dims)
    ok = dist.is_initialized() and (r, w) == (0, 1) and set(got) == set(ref) and all(np.array_equal(got[k], ref[k]) for k in ref)
    ok = ok and gather_tokens([[1, 2], [3]], 3).tolist() == [[1, 2, -1], [3, -1, -1]]
    ok = ok and gather_logits(np.ones((2, 5), np.float32)).shape == (1, 2, 5)
    barrier()
    dist.destroy_process_group()
    q.put(ok)


def test_forced_single_rank_group_runs_the_collective_paths():
    """TTASR_DIST_FORCE=1: the process-group code (bucketed broadcast, gathers, barrier) runs at WORLD_SIZE = 1 - the switch the
    GPU suite uses to execute the RCCL transport on a one-GPU box."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_single_rank, args=(q,))
    p.start()
    assert q.get(timeout=180) is True
    p.join(60)
    assert p.exitcode == 0
