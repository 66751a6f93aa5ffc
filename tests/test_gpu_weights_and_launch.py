"""GPU: the device-side weight intake (ttasr_load_tensor_device: what the RCCL broadcast hands over, f32 or bf16 bits,
no host staging) gives bit-identical engines to the host intake; `python bench.py --gpus 2` launches itself from a bare
shell (two ranks sharing this box's one GPU over gloo: RCCL refuses two ranks per device, so this is the plumbing check -
weight broadcast, sharded clips, token gather, rank-to-rank logits validation)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, COMPUTE_F32, PRESETS

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("compute", [COMPUTE_BF16, COMPUTE_F32])
def test_device_weight_intake_is_bit_identical_to_host_intake(compute):
    import torch
    from taiwan_tongues_asr_ce_amd.dist import _is_matrix
    from taiwan_tongues_asr_ce_amd.engine import DeviceTensor, Engine
    dims = PRESETS["tiny"]
    clips = [synth.noise_clip(0), synth.tonal_clip(1)]
    outs = []
    for route in ("host", "device"):
        e = Engine(dims, compute, 2)
        if route == "host":
            e.load_weights(synth.iter_weights(dims))
        else:
            keep = []

            def views():
                for name, arr in synth.iter_weights(dims):
                    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).cuda()
                    as_bf16 = compute == COMPUTE_BF16 and _is_matrix(name, arr.shape)   # what dist.broadcast_weights sends
                    if as_bf16:
                        t = t.to(torch.bfloat16)
                    torch.cuda.synchronize()
                    keep.append(t)
                    yield name, DeviceTensor(t.data_ptr(), 1 if as_bf16 else 0, tuple(arr.shape))
            e.load_weights(views())
        st = e.special
        e.log_mel(clips, want_output=False)
        enc = e.encode(2, want_output=True)
        e.decode_reset(2)
        lg = [e.decode_step([t, t]) for t in (st.sot, st.lang_zh, st.transcribe)]
        outs.append((enc, lg))
        e.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert np.array_equal(a, b)


def test_device_intake_rejects_bad_arguments():
    import ctypes as C
    from taiwan_tongues_asr_ce_amd.engine import Engine
    e = Engine(PRESETS["micro"], COMPUTE_F32, 1)
    dims = (C.c_int64 * 1)(128)
    assert e.lib.ttasr_load_tensor_device(e.h, b"model.encoder.layer_norm.weight", None, 0, dims, 1) != 0
    assert e.lib.ttasr_load_tensor_device(e.h, b"model.encoder.layer_norm.weight", C.c_void_p(16), 7, dims, 1) != 0
    assert b"dtype" in e.lib.ttasr_last_error(e.h)
    e.close()


def test_bench_launches_its_own_ranks():
    """The driver's invocation form: `python bench.py --gpus 2` with no torchrun environment."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model", "tiny", "--batch", "2",
                        "--steps", "2", "--warmup", "1", "--new-tokens", "8", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["clips_per_gpu"] == 2
    assert out["config"]["rank_logits_spread"] == 0.0          # both ranks hold the same broadcast weights, bit for bit
    assert "share GPUs over gloo" in out["config"]["parallelism"]


def test_rccl_transport_runs_on_one_rank():
    """the on-device broadcast branch of dist.broadcast_tensors, the device-tensor gathers and the
    RCCL barrier had never executed anywhere.  A ONE-rank "nccl" process group (TTASR_DIST_FORCE=1; RCCL accepts a single rank
    per device) runs exactly those calls on this box: the engine loaded through the RCCL buckets is bit-identical (encoder
    output, logits, tokens) to the engine loaded from host arrays, in bf16 / fp16 (matrices travel as 16-bit words) and f32."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "TTASR_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_single_rank_probe.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["backend"] == "nccl" and out["world"] == 1
    for k in ("bf16_encoder_equal", "bf16_logits_equal", "bf16_tokens_equal", "f16_encoder_equal", "f16_logits_equal",
              "f16_tokens_equal", "f32_encoder_equal", "f32_logits_equal", "f32_tokens_equal", "gather_tokens", "gather_logits",
              "broadcast_weights_runs"):
        assert out[k] is True, (k, out)


def test_bench_single_gpu_through_the_forced_rccl_group():
    """`bench.py --gpus 1` with TTASR_DIST_FORCE=1: weights arrive through the RCCL broadcast, tokens leave through the RCCL
    all-gather, the validation probe gathers first-step logits - and the spread across the (one) rank is exactly 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "TTASR_DIST_BACKEND")}
    env["TTASR_DIST_FORCE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--model", "tiny", "--batch", "2",
                        "--steps", "2", "--warmup", "1", "--new-tokens", "8", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["config"]["transport"] == "nccl"
    assert out["config"]["rank_logits_spread"] == 0.0
    assert out["output_check"]["replay_bit_identical"] is True


_SHARE_CHILD = r"""
import json, sys, time
sys.path.insert(0, %r)
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import PRESETS, COMPUTE_BF16
from taiwan_tongues_asr_ce_amd.engine import Engine
dims = PRESETS["small"]; B = 8
e = Engine(dims, COMPUTE_BF16, B)
e.load_weights(synth.iter_weights(dims))
st = e.special
prompt = [st.sot, st.lang_zh, st.transcribe, st.no_timestamps]
opts = e.gen_opts(96, False, suppress_eot=True, check_interval=1 << 20)
e.log_mel([synth.noise_clip(i) for i in range(B)], want_output=False); e.encode(B); e.generate([prompt] * B, opts)
time.sleep(max(0.0, float(sys.argv[1]) - time.time()))      # both processes start their timed loop together
ts = []
for _ in range(6):
    t0 = time.perf_counter(); e.generate([prompt] * B, opts); ts.append(time.perf_counter() - t0)
print(json.dumps({"decode_ms": sorted(ts)[len(ts) // 2] * 1e3}), flush=True)
"""


def test_two_processes_sharing_the_gpu_overlap_their_decode_chains():
    """Round 5 regression guard.  Several service processes per GPU is a deployment the reference allows (one model per process,
    api/streaming_asr.py:85-86).  Two such processes' latency-bound decode chains OVERLAP on the GPU (each takes ~1.5 x the time
    it takes alone) - as long as every context owns exactly ONE stream: with a second stream per context merely existing, the GPU's
    scheduler time-slices the processes and each ran 3.2 x slower (profiles/r5_second_stream_two_processes.jsonl).
    whisper-small, 8 clips, 96 greedy tokens: two concurrent processes must each stay under 2.4 x the solo time."""
    import time

    def run(n):
        start = time.time() + (25 if n > 1 else 0)
        ps = [subprocess.Popen([sys.executable, "-c", _SHARE_CHILD % ROOT, str(start)], stdout=subprocess.PIPE, text=True) for _ in range(n)]
        out = []
        for p in ps:
            o, _ = p.communicate(timeout=600)
            assert p.returncode == 0
            out.append(json.loads([ln for ln in o.splitlines() if ln.startswith("{")][-1])["decode_ms"])
        return out
    solo = run(1)[0]
    both = run(2)
    assert max(both) < 2.4 * solo, (solo, both)
