"""Golden vectors for SURVEY.md row a11 (30-s window loop: seek, segment split on timestamp pairs, previous-text prompt):
what HF-Transformers' long-form Whisper generation ([HF] generation_whisper.py:785-903 window loop, :970-1116 fallback,
:1830-1990 segment retrieval - the implementation the reference trains / evaluates with) produces for a 70-s synthetic
recording on the seeded tiny-geometry model.  Run ONLY in the build container:  python oracle/make_golden_longform.py
Output: tests/golden/longform.json (segments: start, end, token ids per case).  Data only."""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import make_golden as G  # noqa: E402
from taiwan_tongues_asr_ce_amd import synth  # noqa: E402
from taiwan_tongues_asr_ce_amd.config import NON_SPEECH_TOKENS_MULTI, PRESETS  # noqa: E402

torch.set_grad_enabled(False)


def recording():
    """70 s: 30 s of noise, 30 s of tones, 10 s of noise (the synthetic clips of the other goldens, concatenated)."""
    return np.concatenate([synth.noise_clip(0), synth.tonal_clip(1), synth.noise_clip(2)[:160000]])


def main():
    dims = PRESETS["tiny"]
    model, st = G.hf_model(dims)
    fe = G.WhisperFeatureExtractor(feature_size=dims.n_mels)
    audio = recording()
    feat = fe(audio, sampling_rate=16000, return_tensors="pt", truncation=False, padding="longest", return_attention_mask=True)
    gc = model.generation_config
    suppress = sorted(set(list(NON_SPEECH_TOKENS_MULTI) + [st.translate, st.transcribe, st.sot, st.sot_prev, st.no_speech, st.sot_lm]))
    gc.suppress_tokens = suppress
    gc.begin_suppress_tokens = [220, st.eot]
    gc.no_timestamps_token_id = st.no_timestamps
    gc.lang_to_id = {"<|zh|>": st.lang_zh, "<|en|>": st.sot + 1}
    gc.task_to_id = {"transcribe": st.transcribe, "translate": st.translate}
    gc.is_multilingual = True
    gc.max_initial_timestamp_index = 50
    gc.prev_sot_token_id = st.sot_prev
    gc.max_length = dims.n_text_ctx
    gc.decoder_start_token_id, gc.eos_token_id, gc.pad_token_id, gc.bos_token_id = st.sot, st.eot, st.eot, st.eot
    out = {"audio": "noise_clip(0) + tonal_clip(1) + noise_clip(2)[:160000]", "n_samples": int(len(audio)),
           "suppress": suppress, "begin_suppress": [220, st.eot], "cases": {}}
    for name, kw in (("cond_prev_48", dict(condition_on_prev_tokens=True, max_new_tokens=48)),
                     ("no_cond_48", dict(condition_on_prev_tokens=False, max_new_tokens=48)),
                     ("cond_prev_120", dict(condition_on_prev_tokens=True, max_new_tokens=120))):
        r = model.generate(feat["input_features"], attention_mask=feat["attention_mask"], return_timestamps=True, language="zh",
                           task="transcribe", temperature=0.0, return_segments=True, do_sample=False, num_beams=1, **kw)
        segs = [{"start": round(float(s["start"]), 3), "end": round(float(s["end"]), 3), "tokens": s["tokens"].tolist()}
                for s in r["segments"][0]]
        out["cases"][name] = {"options": kw, "segments": segs}
        print(name, len(segs), [(s["start"], s["end"]) for s in segs][:6])
    # the whole-file features HF feeds its window loop (every 9th frame + the frames around the window seams and the end)
    feats = feat["input_features"][0].numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "longform_features.npz"), stride9=feats[:, ::9],
                        seam=feats[:, 2930:3010], tail=feats[:, -40:], n_frames=np.array(feats.shape[1]))
    with open(os.path.join(ROOT, "tests", "golden", "longform.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
