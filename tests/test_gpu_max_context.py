"""GPU: the decoder context filled to its last position (SURVEY §8d "N_dec = 444 fills 448"): 28 KV pages of 16
tokens exactly full, the longest self-attention, the longest legal prompt through the prefill pass.  The f32 engine
is checked against the f32 oracle by teacher forcing: at every one of the positions the engine's choice must be the
oracle's argmax of the processed logits, or within 1e-3 of it (the north-star logit tolerance)."""
import numpy as np
import pytest
import torch

from oracle import whisper_ref as R
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, COMPUTE_F32, PRESETS

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _engine(name, compute, max_batch):
    from taiwan_tongues_asr_ce_amd.engine import Engine
    e = Engine(PRESETS[name], compute, max_batch)
    e.load_weights(synth.iter_weights(PRESETS[name]))
    return e


def _teacher_forced_gap(enc, prompt, tokens, W, dims, rules):
    """max over positions of (oracle's best allowed logit - logit of the engine's token); ONE causal oracle pass over
    prompt + tokens gives the logits of every position."""
    xkv = R.cross_kv(enc, W, dims)
    seq = list(prompt) + list(tokens[:-1])
    logits = R.decoder_forward(torch.tensor([seq]), R.SelfCache.empty(dims.dec_layers), xkv, W, dims)[0]
    worst = 0.0
    for i, t in enumerate(tokens):
        s = R.apply_rules(logits[len(prompt) - 1 + i], tokens[:i], rules)
        assert s[t] > -np.inf, f"position {i}: engine emitted a masked token"
        worst = max(worst, float(s.max() - s[t]))
    return worst


@pytest.mark.parametrize("compute,tol", [(COMPUTE_F32, 1e-3), (COMPUTE_BF16, 0.15)])
def test_fill_the_448_token_context(compute, tol):
    from taiwan_tongues_asr_ce_amd.engine import TtasrError, default_suppress
    name = "tiny"
    pd = PRESETS[name]
    dims = R.Dims(**pd.as_dict())
    W = R.to_torch(synth.state_dict(pd), round_bf16=compute == COMPUTE_BF16)
    e = _engine(name, compute, 2)
    st = e.special
    clips = [synth.noise_clip(3), synth.tonal_clip(4)]
    e.log_mel(clips, want_output=False)
    e.encode(2)
    enc = R.encoder_forward(torch.from_numpy(np.stack([R.log_mel(c, pd.n_mels) for c in clips])), W, dims)
    rules = R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin,
                    suppress=default_suppress(st, dims.vocab) + [st.eot], begin_suppress=[220, st.eot], timestamps=False)
    # (a) 4-token prompt + 444 new tokens = 448 positions (EOT suppressed so the row cannot stop early)
    prompt = [st.sot, st.lang_zh, st.transcribe, st.no_timestamps]
    res = e.generate([prompt] * 2, e.gen_opts(444, False, suppress_eot=True, check_interval=64))
    assert [len(t) for t in res.tokens] == [444, 444]
    assert np.isfinite(res.sum_logprob).all()
    for b in range(2):
        assert _teacher_forced_gap(enc[b:b + 1], prompt, res.tokens[b], W, dims, rules) <= tol
    # (b) asking for more than the context holds stops at the context, it does not overrun the KV pages
    res2 = e.generate([prompt] * 2, e.gen_opts(448, False, suppress_eot=True))
    assert [len(t) for t in res2.tokens] == [444, 444]
    # (c) the longest prompt faster-whisper builds: <|startofprev|> + 223 previous tokens + sot/lang/task, through the
    #     prefill pass, then decode to the end of the context (448 - 227 = 221 tokens)
    rng = np.random.default_rng(1)
    long_prompt = [st.sot_prev] + rng.integers(300, 20000, size=223).tolist() + [st.sot, st.lang_zh, st.transcribe]
    rules_ts = R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin,
                       suppress=default_suppress(st, dims.vocab) + [st.eot], begin_suppress=[220, st.eot], timestamps=True)
    res3 = e.generate([long_prompt], e.gen_opts(400, True, suppress_eot=True, sot_index=224))
    assert len(res3.tokens[0]) == 448 - len(long_prompt)
    assert _teacher_forced_gap(enc[0:1], long_prompt, res3.tokens[0], W, dims, rules_ts) <= tol
    # (d) a prompt that leaves no room is refused, not truncated
    for n in (448, 449):
        with pytest.raises(TtasrError):
            e.generate([[st.sot] * n], e.gen_opts(4, False))
    assert len(e.generate([[st.sot] * 447], e.gen_opts(4, False, suppress_eot=True, no_speech=False)).tokens[0]) == 1
    e.close()
