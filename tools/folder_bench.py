import sys, time, warnings
sys.path.insert(0, '.')
import numpy as np
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.model import WhisperModel
warnings.simplefilter("ignore")
name = sys.argv[1] if len(sys.argv) > 1 else "large-v3"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 30      # decode rows of the engine: files in lock step = rows // beam
m = WhisperModel(f"synthetic:{name}", device="cuda", compute_type="bfloat16", max_batch=rows)
files = [np.concatenate([synth.tonal_clip(2 * i), synth.noise_clip(2 * i + 1)]) for i in range(12)]   # 12 x 60 s
kw = dict(language="zh", beam_size=5, temperature=0.0, log_prob_threshold=None, max_new_tokens=64)
list(m.transcribe(files[0], **kw)[0]); m.transcribe_many(files[:rows // 5], **kw)                        # warm-up
t = time.perf_counter()
for f in files:
    list(m.transcribe(f, **kw)[0])
t_seq = time.perf_counter() - t
t = time.perf_counter()
m.transcribe_many(files, **kw)
t_many = time.perf_counter() - t
print(f"{name}: 12 files x 60 s, beam 5, <= 64 tokens per window: one by one {t_seq:.2f} s ({720/t_seq:.0f} x real time), "
      f"{min(rows // 5, 12)} files in lock step {t_many:.2f} s ({720/t_many:.0f} x real time), speed-up {t_seq/t_many:.2f}")
