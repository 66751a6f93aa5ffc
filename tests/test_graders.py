"""CPU: the graders of the GPU parity tests grade (tests/oracle_checks.py).  teacher_forced_causal - one causal oracle pass per
row, used for the 128- / 444-token decodes at the measured width - must agree with the step-by-step teacher_forced on the
oracle's own greedy tokens (every step clear, gap 0) and must reject a row with one wrong token."""
import numpy as np
import pytest
import torch

from oracle import whisper_ref as R
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import PRESETS, SpecialTokens

from oracle_checks import teacher_forced, teacher_forced_causal

torch.set_grad_enabled(False)


@pytest.mark.parametrize("ts", [False, True])
def test_causal_grader_matches_stepwise_and_rejects_a_wrong_token(ts):
    d = PRESETS["micro"]
    rd = R.Dims(**d.as_dict())
    W = R.to_torch(synth.state_dict(d))
    n = d.n_frames * 160
    mel = np.stack([R.log_mel(synth.noise_clip(i, n), d.n_mels, n) for i in range(5)])
    enc = R.encoder_forward(torch.from_numpy(mel), W, rd)
    st = SpecialTokens.for_vocab(d.vocab)
    prompt = [st.sot, st.lang_zh, st.transcribe] + ([] if ts else [st.no_timestamps])
    rules = R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin, suppress=[1, 2, 7, st.sot],
                    begin_suppress=[5, st.eot], timestamps=ts, suppress_eot=True)
    ref = R.greedy_decode(enc, prompt, W, rd, rules, 20)
    assert all(len(t) == 20 for t in ref.tokens)
    a = teacher_forced_causal(ref.tokens, prompt, enc, W, rd, rules, tol=1e-4, margin=1e-4, rows_per_pass=2)
    b = teacher_forced(ref.tokens, prompt, enc, W, rd, rules, tol=1e-4, margin=1e-4)
    assert (a.n_steps, a.n_clear) == (b.n_steps, b.n_clear) == (100, 100) and a.worst < 1e-5 and b.worst < 1e-5
    bad = [list(t) for t in ref.tokens]
    bad[3][7] = (bad[3][7] + 1) % 400 + 10
    with pytest.raises(AssertionError, match="row 3 step 7"):
        teacher_forced_causal(bad, prompt, enc, W, rd, rules, tol=1e-4, margin=1e-4)


def test_oracle_activation_clamp_is_off_by_default_and_clamps_when_asked():
    """`R.activation_clamp` (used by the fp16 saturation test): no effect on unsaturated activations, restores the previous state,
    and clamps every linear output (q after its scaling) when the weights push them past the limit."""
    d = PRESETS["micro"]
    rd = R.Dims(**d.as_dict())
    sd = dict(synth.state_dict(d))
    n = d.n_frames * 160
    mel = torch.from_numpy(R.log_mel(synth.noise_clip(1, n), d.n_mels, n))[None]
    W = R.to_torch(sd)
    base = R.encoder_forward(mel, W, rd)
    with R.activation_clamp(65504.0):
        same = R.encoder_forward(mel, W, rd)
    assert torch.equal(base, same) and R._ACT_CLAMP is None
    sd["model.encoder.layers.0.fc1.weight"] = sd["model.encoder.layers.0.fc1.weight"] * np.float32(1e6)
    W2 = R.to_torch(sd)
    free = R.encoder_forward(mel, W2, rd)
    with R.activation_clamp(65504.0):
        clamped = R.encoder_forward(mel, W2, rd)
    assert torch.isfinite(clamped).all() and float((free - clamped).abs().max()) > 0.1
    x = torch.tensor([[1e9, -1e9, 3.0]])
    with R.activation_clamp(10.0):
        assert R._sat(x).tolist() == [[10.0, -10.0, 3.0]]
    assert R._sat(x).tolist() == x.tolist()
