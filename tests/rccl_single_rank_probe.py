"""Helper PROCESS of tests/test_gpu_weights_and_launch.py (not a test module): a ONE-rank RCCL process group
(TTASR_DIST_FORCE=1, backend "nccl", device_id bound) through which the product's multi-GPU code paths run on a one-GPU box -
dist.broadcast_tensors' device branch (bf16 / f32 buckets in HBM -> DeviceTensor views -> ttasr_load_tensor_device),
gather_tokens / gather_logits on device tensors, dist.barrier(device_ids=...).  Prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.update(TTASR_DIST_FORCE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    os.environ.pop("TTASR_DIST_BACKEND", None)
    import torch
    import torch.distributed as dist
    from taiwan_tongues_asr_ce_amd import synth
    from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, COMPUTE_F16, COMPUTE_F32, PRESETS
    from taiwan_tongues_asr_ce_amd.dist import barrier, broadcast_weights, gather_logits, gather_tokens, init_process_group
    from taiwan_tongues_asr_ce_amd.engine import Engine
    rank, world, local = init_process_group()
    out = {"backend": dist.get_backend(), "world": world}
    dims = PRESETS["tiny"]
    clips = [synth.noise_clip(0), synth.tonal_clip(1)]
    for compute, tag in ((COMPUTE_BF16, "bf16"), (COMPUTE_F16, "f16"), (COMPUTE_F32, "f32")):
        res = []
        for route in ("host", "rccl"):
            e = Engine(dims, compute, 2, device=local)
            if route == "host":
                if compute == COMPUTE_F16:   # the broadcast rounds the matrices to fp16 once on rank 0: give the host route the same values
                    from taiwan_tongues_asr_ce_amd.dist import _is_matrix
                    e.load_weights((n, a.astype(np.float16).astype(np.float32) if _is_matrix(n, a.shape) else a)
                                   for n, a in synth.iter_weights(dims))
                else:
                    e.load_weights(synth.iter_weights(dims))
            else:   # small buckets: several broadcasts per dtype
                from taiwan_tongues_asr_ce_amd.dist import broadcast_tensors
                e.load_weights(broadcast_tensors(dims, synth.iter_weights(dims), local, bucket_bytes=8 << 20,
                                                 matrix_dtype={COMPUTE_BF16: "bf16", COMPUTE_F16: "f16"}.get(compute)))
            st = e.special
            e.log_mel(clips, want_output=False)
            enc = e.encode(2, want_output=True)
            e.decode_reset(2)
            lg = [e.decode_step([t, t]) for t in (st.sot, st.lang_zh, st.transcribe)]
            toks = e.generate([[st.sot, st.lang_zh, st.transcribe]] * 2, e.gen_opts(6, True)).tokens
            res.append((enc, lg, toks))
            e.close()
        out[f"{tag}_encoder_equal"] = bool(np.array_equal(res[0][0], res[1][0]))
        out[f"{tag}_logits_equal"] = bool(all(np.array_equal(a, b) for a, b in zip(res[0][1], res[1][1])))
        out[f"{tag}_tokens_equal"] = res[0][2] == res[1][2]
        if tag == "bf16":
            all_t = gather_tokens(res[1][2], 6, device=local)                   # all_gather of a device tensor over RCCL
            out["gather_tokens"] = all_t.tolist() == [list(t) + [-1] * (6 - len(t)) for t in res[1][2]]
            all_l = gather_logits(res[1][1][0], device=local)
            out["gather_logits"] = bool(all_l.shape == (1,) + res[1][1][0].shape and np.array_equal(all_l[0], res[1][1][0]))
    # the whole-engine route used by bench.py
    e = Engine(dims, COMPUTE_BF16, 2, device=local)
    broadcast_weights(e, dims, synth.iter_weights(dims), device=local)
    e.log_mel(clips, want_output=False)
    out["broadcast_weights_runs"] = bool(np.isfinite(e.encode(2, want_output=True)).all())
    e.close()
    barrier(local)
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
