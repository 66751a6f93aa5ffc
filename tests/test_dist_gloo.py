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
    from taiwan_tongues_asr_ce_amd.dist import broadcast_tensors, gather_tokens, init_process_group, shard_range
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
