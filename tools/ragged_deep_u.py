"""Lab: decode-step time against the number of live rows for a given build of the library (the first k rows of 32 live).
    python tools/ragged_deep_u.py [libttasr.so]"""
import json, sys
sys.path.insert(0, '.')
import numpy as np
from taiwan_tongues_asr_ce_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = sys.argv[1]
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
import zlib
out = {"lib": sys.argv[1] if len(sys.argv) > 1 else "shipped", "all_live_ms": round(full, 2), "all_live_per_step_ms": round((full - short) / (N - 4), 4),
       "tokens_crc": zlib.crc32(np.asarray(r_full.tokens, dtype=np.int32).tobytes())}
for k in (24, 20, 16, 12, 8, 4, 1):
    caps = np.full(B, 4, np.int32); caps[:k] = N
    t, r = run(caps)
    out[f"live_{k}_per_step_ms"] = round((t - short) / (N - 4), 4)
    assert all(r.tokens[i] == r_full.tokens[i] for i in range(k))
print(json.dumps(out), flush=True)
e.close()
