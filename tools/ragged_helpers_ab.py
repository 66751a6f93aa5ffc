"""Lab: decode-step time at k live rows of 32 with and without PREFETCHING helper workgroups.  NEEDS A LAB BUILD: the helper
branch (workgroups behind the live ones stream sections of the live items' K / V with plain loads and discard them; switched by
bits 16+ of option xattn_deep_items) was measured and removed - DESIGN.md 4.12, profiles/r6_xattn_prefetch_helpers.jsonl."""
import json, sys
sys.path.insert(0, '.')
import numpy as np
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, PRESETS
from taiwan_tongues_asr_ce_amd.engine import Engine
dims = PRESETS["large-v3"]
B, N = 32, 128
e = Engine(dims, COMPUTE_BF16, B)
e.load_weights(synth.iter_weights(dims))
clips = [synth.noise_clip(i) for i in range(B)]
st = e.special
prompt = [st.sot, st.lang_zh, st.transcribe, st.no_timestamps]
opts = e.gen_opts(N, False, suppress_eot=True, check_interval=1 << 20)
e.log_mel(clips, want_output=False); e.encode(B)
def run(caps, reps=3):
    ms = []
    for _ in range(reps):
        r = e.generate([prompt] * B, opts, row_max_new=caps)
        ms.append(e.phase_ms()["decode"])
    return float(np.median(ms)), r
full, r_full = run(None)
short, _ = run(np.full(B, 4, np.int32))
for rnd in range(2):
  for nt in (1, 0):
    e.set_option("xattn_nontemporal", nt)
    for mode in (0, 1):
        e.set_option("xattn_deep_items", 512 + 65536 * mode)
        out = {"round": rnd, "consumer_nontemporal_loads": nt, "helpers": mode, "all_live_per_step_ms": round((run(None)[0] - short) / (N - 4), 4)}
        for k in (8, 4, 2, 1):
            caps = np.full(B, 4, np.int32); caps[:k] = N
            t, r = run(caps)
            out[f"live_{k}_per_step_ms"] = round((t - short) / (N - 4), 4)
            assert all(r.tokens[i] == r_full.tokens[i] for i in range(k))
        print(json.dumps(out), flush=True)
e.close()
