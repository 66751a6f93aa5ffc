"""Folder path (WhisperModel.transcribe_groups) against the number of groups in flight: 24 synthetic recordings of 60 s in 4 groups
of 6 (beam 5, <= 64 tokens per window), pipeline_depth 1 ... 4 on engine contexts that share one copy of the weights.
One JSON line per depth: audio-s/s, wall time, results identical to depth 1, device memory the extra contexts cost."""
import json, sys, time, warnings
sys.path.insert(0, '.')
import numpy as np
import torch
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.model import WhisperModel
name = sys.argv[1] if len(sys.argv) > 1 else "large-v3"
MAXB = int(sys.argv[2]) if len(sys.argv) > 2 else 30         # decode rows per context: files per group = MAXB // 5
NF = int(sys.argv[3]) if len(sys.argv) > 3 else 24
DEPTHS = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 2, 3, 4]
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    wm = WhisperModel(f"synthetic:{name}", device="cuda", compute_type="bfloat16", max_batch=MAXB, pipeline_depth=4)
    files = [np.concatenate([synth.tonal_clip(2 * i), synth.noise_clip(2 * i + 1)]) for i in range(NF)]
    per = MAXB // 5
    groups = [files[i:i + per] for i in range(0, NF, per)]
    kw = dict(language="zh", beam_size=5, temperature=0.0, log_prob_threshold=None, max_new_tokens=64)
    free0 = torch.cuda.mem_get_info()[0]
    ref = None
    wm.transcribe_groups(groups, pipeline_depth=1, **kw)
    for depth in DEPTHS:
        f_before = torch.cuda.mem_get_info()[0]
        wm.transcribe_groups(groups, pipeline_depth=depth, **kw)                     # warm-up (contexts, graphs)
        cost = f_before - torch.cuda.mem_get_info()[0]
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            r = wm.transcribe_groups(groups, pipeline_depth=depth, **kw)
            best = min(best, time.perf_counter() - t0)
        flat = [[(sg.start, sg.end, tuple(sg.tokens)) for sg in segs] for g in r for segs, _ in g]
        if ref is None:
            ref = flat
        print(json.dumps({"model": name, "pipeline_depth": depth, "max_batch": MAXB, "files": NF, "groups": len(groups), "audio_s_per_s": round(NF * 60.0 / best, 1), "wall_s": round(best, 3),
                          "identical_to_depth_1": flat == ref, "contexts": len(wm._lanes),
                          "device_MB_added_by_this_depth": round(cost / 2 ** 20)}), flush=True)
    wm.close()
