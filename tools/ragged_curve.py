"""Decode-step time against the number of LIVE rows of a 32-row batch (round 6): every row but k gets a 4-token budget, k rows run
the full 128 tokens, so steps 4 ... 127 run with exactly k live rows (rows spread over the batch, or the first k).  Beside it: a
batch of k rows, all live (what a perfectly compacted batch would cost).  One JSON line per k."""
import json, sys, time
sys.path.insert(0, '.')
import numpy as np
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, PRESETS
from taiwan_tongues_asr_ce_amd.engine import Engine

name = sys.argv[1] if len(sys.argv) > 1 else "large-v3"
dims = PRESETS[name]
B, N = 32, 128
e = Engine(dims, COMPUTE_BF16, B)
e.load_weights(synth.iter_weights(dims))
clips = [synth.noise_clip(i) for i in range(B)]
st = e.special
prompt = [st.sot, st.lang_zh, st.transcribe, st.no_timestamps]
opts = e.gen_opts(N, False, suppress_eot=True, check_interval=1 << 20)
e.log_mel(clips, want_output=False); e.encode(B)
def run(caps, reps=3, n=B):
    ms = []
    for _ in range(reps):
        e.generate([prompt] * n, opts, row_max_new=caps)
        ms.append(e.phase_ms()["decode"])
    return float(np.median(ms))
full = run(None)
short = run(np.full(B, 4, np.int32))           # everybody leaves after 4 tokens: the cost of prefill + 4 steps
print(json.dumps({"all_live_ms": round(full, 2), "per_step_ms": round((full - short) / (N - 4), 4), "four_steps_ms": round(short, 2)}), flush=True)
if "--sweep" in sys.argv:      # where should the deep form take over?  (option xattn_deep_items; first k rows live)
    for k in (30, 28, 26, 24, 22, 20, 16):
        out = {"live_rows": k}
        caps = np.full(B, 4, np.int32); caps[:k] = N
        for items in (0, 448, 512, 576, 640):
            e.set_option("xattn_deep_items", items)
            out[f"deep_items_{items}_per_step_ms"] = round((run(caps) - short) / (N - 4), 4)
        print(json.dumps(out), flush=True)
    e.close()
    sys.exit(0)
for k in (32, 28, 26, 24, 20, 16, 13, 12, 8, 4, 2, 1):
    out = {"live_rows": k}
    for tag, rows in (("spread", np.linspace(0, B - 1, k).round().astype(int)), ("first", np.arange(k))):
        caps = np.full(B, 4, np.int32); caps[rows] = N
        for opt in (1, 0):
            e.set_option("ragged_exit", opt)
            t = run(caps)
            out[f"{tag}_{'exit' if opt else 'static'}_per_step_ms"] = round((t - short) / (N - 4), 4)
        e.set_option("ragged_exit", 1)
    e.log_mel(clips[:k], want_output=False); e.encode(k)
    t_k = run(None, n=k); t_k4 = run(np.full(k, 4, np.int32), n=k)
    out["batch_of_k_per_step_ms"] = round((t_k - t_k4) / (N - 4), 4)
    e.log_mel(clips, want_output=False); e.encode(B)
    print(json.dumps(out), flush=True)
e.close()
