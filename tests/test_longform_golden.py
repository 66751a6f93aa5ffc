"""SURVEY.md row a11 (30-s window loop: seek, segment split on timestamp pairs, previous-text prompt, whole-file features)
pinned to HF-Transformers' long-form Whisper generation ([HF] generation_whisper.py:785-903, :1830-1990) on a 70-s synthetic
recording and the seeded tiny-geometry model: tests/golden/longform.json + longform_features.npz, written by
oracle/make_golden_longform.py.  The CPU suite drives WhisperModel's host-side loop with the oracle standing in for the
engine (tests/oracle_engine.py); the GPU suite runs the same assertions on the HIP engine.

Where faster-whisper (the implementation the product is a drop-in for) deliberately differs from HF, the test says so:
  * the last segment of a window: HF keeps BOTH closing timestamp tokens, openai-whisper / faster-whisper slice
    tokens[last:current] and keep one;
  * timestamps beyond the end of the recording (a random-weight model emits them freely): this build clamps segment times to
    the recording and drops segments that start after it; HF reports them as decoded."""
import json
import os
import warnings

import numpy as np
import pytest

from oracle import whisper_ref as R
from taiwan_tongues_asr_ce_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def recording():
    return np.concatenate([synth.noise_clip(0), synth.tonal_clip(1), synth.noise_clip(2)[:160000]])


def check_against_hf(model, case, audio_seconds):
    kw = case["options"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        segs, _ = model.transcribe(recording(), language="zh", beam_size=1, temperature=0.0,
                                   condition_on_previous_text=kw["condition_on_prev_tokens"], max_new_tokens=kw["max_new_tokens"],
                                   no_speech_threshold=None, log_prob_threshold=None, compression_ratio_threshold=None)
        segs = list(segs)
    want = [h for h in case["segments"] if h["start"] < audio_seconds]        # HF also reports segments past the recording
    assert len(segs) == len(want) >= 7
    seeks = set()
    for s, h in zip(segs, want):
        assert s.tokens in (h["tokens"], h["tokens"][:-1]), (s.tokens, h["tokens"])     # see the module docstring
        assert abs(s.start - h["start"]) < 1e-6
        assert abs(s.end - min(h["end"], audio_seconds)) < 1e-6
        seeks.add(s.seek)
    # three windows: 0, then where the last timestamp pair of each window ended (29.36 s; + 29.38 s resp. + 28.x s)
    assert len(seeks) == 3 and min(seeks) == 0
    return segs


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(GOLDEN, "longform.json")) as f:
        return json.load(f)


def test_oracle_whole_file_features_match_hf():
    g = np.load(os.path.join(GOLDEN, "longform_features.npz"))
    f = R.log_mel_file(recording(), 80)
    assert f.shape == (80, int(g["n_frames"])) == (80, 7000)
    np.testing.assert_allclose(f[:, ::9], g["stride9"], atol=1e-4)
    np.testing.assert_allclose(f[:, 2930:3010], g["seam"], atol=1e-4)      # frames around the first window seam: no reflection there
    np.testing.assert_allclose(f[:, -40:], g["tail"], atol=1e-4)
    w = R.file_window(f, 5874)                                              # the last window: 1126 real frames, then zeros
    assert w.shape == (80, 3000) and np.array_equal(w[:, :1126], f[:, 5874:]) and not w[:, 1126:].any()


@pytest.mark.parametrize("name", ["cond_prev_48", "no_cond_48", "cond_prev_120"])
def test_window_loop_reproduces_hf_long_form_on_the_oracle(golden, name):
    from oracle_engine import OracleEngine
    from taiwan_tongues_asr_ce_amd.model import WhisperModel
    m = WhisperModel("synthetic:tiny", device="cuda", compute_type="float32", max_batch=1, _engine_factory=OracleEngine)
    check_against_hf(m, golden["cases"][name], golden["n_samples"] / 16000.0)
    prompts = [c[1] for c in m.engine.calls if c[0] == "generate"]
    st = m.special
    assert len(prompts) == 3 and prompts[0][0] == st.sot
    if golden["cases"][name]["options"]["condition_on_prev_tokens"]:
        # later windows: <|startofprev|> + the tokens of every segment yielded so far (timestamps included) + the sot sequence
        assert prompts[1][0] == st.sot_prev and prompts[1][-3:] == [st.sot, st.lang_zh, st.transcribe]
        assert prompts[1][1:-3] == [50384, 13859, 50564, 50564, 13859, 51137, 51137, 13859, 51693, 51693, 13859, 51832]
        assert len(prompts[2]) > len(prompts[1])
    else:
        assert all(p == [st.sot, st.lang_zh, st.transcribe] for p in prompts)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cond_prev_48", "no_cond_48"])
def test_window_loop_reproduces_hf_long_form_on_the_gpu(golden, name):
    from taiwan_tongues_asr_ce_amd.model import WhisperModel
    m = WhisperModel("synthetic:tiny", device="cuda", compute_type="float32", max_batch=2)
    check_against_hf(m, golden["cases"][name], golden["n_samples"] / 16000.0)


@pytest.mark.gpu
def test_gpu_window_features_match_the_whole_file_oracle():
    """ttasr_log_mel_windows: frames of the whole-file STFT (neighbour samples across seams, reflection only at the file ends),
    zeros in feature space past the recording, the FILE's dynamic-range floor - against R.log_mel_file / R.file_window, and the
    per-window maxima the first pass returns."""
    from taiwan_tongues_asr_ce_amd.config import COMPUTE_F32, PRESETS
    from taiwan_tongues_asr_ce_amd.engine import Engine
    audio = recording()
    audio[480000:960000] *= 1e-3                     # a quiet second window (-60 dB) ...
    audio[600000:700000] *= 1e-4                     # ... with a near-silent stretch: 140 dB under the file maximum, 80 dB under
    f = R.log_mel_file(audio, 80)                    # the window's own - only the FILE-level floor clamps it (as the reference)
    e = Engine(PRESETS["tiny"], COMPUTE_F32, 4)
    seeks = [0, 3000, 5874, 6990]
    _, mx = e.log_mel_windows(audio, seeks, want_max=True)
    raw = f * 4.0 - 4.0
    file_max = float(max(mx))
    assert abs(file_max - raw.max()) < 1e-3
    got, _ = e.log_mel_windows(audio, seeks, floor_max=[file_max] * 4, want_output=True)
    for b, k in enumerate(seeks):
        np.testing.assert_allclose(got[b], R.file_window(f, k), atol=3e-4, err_msg=str(k))
    own, _ = e.log_mel_windows(audio, [3000], want_output=True)              # the window's own maximum: a different floor
    assert np.abs(own[0] - R.file_window(f, 3000)).max() > 0.1
    # two recordings in one call (files in lock step): each window uses its own file
    other = synth.tonal_clip(5)[:100000]
    got2, _ = e.log_mel_windows([audio, other], [2936, 0], floor_max=[file_max, float((R.log_mel_file(other, 80) * 4 - 4).max())],
                                want_output=True)
    np.testing.assert_allclose(got2[0], R.file_window(f, 2936), atol=3e-4)
    np.testing.assert_allclose(got2[1], R.file_window(R.log_mel_file(other, 80), 0), atol=3e-4)
    e.close()
