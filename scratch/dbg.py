import sys, warnings; sys.path.insert(0,'.')
warnings.simplefilter("ignore")
from taiwan_tongues_asr_ce_amd.model import WhisperModel
from taiwan_tongues_asr_ce_amd import synth
m = WhisperModel("synthetic:tiny", device="cuda", compute_type="float32", max_batch=4)
segs, info = m.transcribe(synth.noise_clip(0), language="zh", beam_size=5, vad_filter=True, initial_prompt="")
for s in segs: print(s.id, s.seek, s.start, s.end, s.tokens[:6], len(s.tokens), repr(s.text[:10]))
