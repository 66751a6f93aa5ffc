"""GPU: word timestamps.  `ttasr_align` (teacher-forced pass, alignment heads' cross-attention rows + raw token
log-probs) against the HF-pinned oracle, and `WhisperModel.transcribe(word_timestamps=True)` end to end."""
import warnings

import numpy as np
import pytest
import torch

from oracle import whisper_ref as R
from taiwan_tongues_asr_ce_amd import alignment as A
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, COMPUTE_F32, PRESETS

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.mark.parametrize("compute,w_tol,lp_tol", [(COMPUTE_F32, 2e-5, 2e-3), (COMPUTE_BF16, 6e-3, 0.15)])
def test_align_weights_and_logprobs_vs_oracle(compute, w_tol, lp_tol):
    from taiwan_tongues_asr_ce_amd.engine import Engine, TtasrError
    pd = PRESETS["tiny"]
    dims = R.Dims(**pd.as_dict())
    e = Engine(pd, compute, 3)
    e.load_weights(synth.iter_weights(pd))
    st = e.special
    W = R.to_torch(synth.state_dict(pd), round_bf16=compute == COMPUTE_BF16)
    clips = [synth.noise_clip(0), synth.tonal_clip(7), synth.burst_clip(2)]
    e.log_mel(clips, want_output=False)
    e.encode(3)
    enc = R.encoder_forward(torch.from_numpy(np.stack([R.log_mel(c, pd.n_mels) for c in clips])), W, dims)
    rng = np.random.default_rng(3)
    tokens = [st.sot, st.lang_zh, st.transcribe, st.no_timestamps] + rng.integers(300, 20000, size=37).tolist() + [st.eot]
    heads = [(3, 0), (3, 5), (2, 1), (1, 4)]
    for clip in (1, 2):                                   # not clip 0: the K/V offset of the clip must be honoured
        w, lp = e.align(clip, tokens, heads)
        rw, rlp = R.alignment_weights(enc[clip:clip + 1], tokens, W, dims, heads, return_logprobs=True)
        assert w.shape == (4, len(tokens), 1500) and lp.shape == (len(tokens) - 1,)
        np.testing.assert_allclose(w.sum(-1), 1.0, atol=1e-4)
        assert np.abs(w - rw.numpy()).max() < w_tol
        assert np.abs(lp - rlp.numpy()).max() < lp_tol
        if compute == COMPUTE_F32:                        # same DTW path -> same token times
            got = A.token_start_times(w, 3, len(tokens) - 1, 3000)
            want = A.token_start_times(rw.numpy(), 3, len(tokens) - 1, 3000)
            assert np.mean(got == want) >= 0.9
    # a generate after align still works (align only invalidates step-level state) and misuse is an error
    assert len(e.generate([[st.sot, st.lang_zh, st.transcribe]] * 3, e.gen_opts(4, True)).tokens) == 3
    for bad in (dict(clip=3), dict(tokens=[st.sot]), dict(heads=[(4, 0)]), dict(heads=[(3, 0), (3, 0)])):
        kw = dict(clip=0, tokens=tokens, heads=heads)
        kw.update(bad)
        with pytest.raises(TtasrError):
            e.align(kw["clip"], kw["tokens"], kw["heads"])
    e.close()


def test_transcribe_with_word_timestamps():
    from taiwan_tongues_asr_ce_amd.model import WhisperModel
    m = WhisperModel("synthetic:tiny", device="cuda", compute_type="float32", max_batch=2)
    audio = synth.tonal_clip(3)[: 20 * 16000]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        segs, info = m.transcribe(audio, language="zh", word_timestamps=True, beam_size=1, temperature=0.0, max_new_tokens=40)
        segs = list(segs)
        plain, _ = m.transcribe(audio, language="zh", word_timestamps=False, beam_size=1, temperature=0.0, max_new_tokens=40)
        plain = list(plain)
    assert len(segs) == len(plain) > 0 and [s.text for s in segs] == [s.text for s in plain]
    assert all(s.words is None for s in plain)
    words = [w for s in segs for w in (s.words or [])]
    assert len(words) > 0
    assert all(hasattr(w, a) for w in words for a in ("word", "start", "end", "probability"))
    assert all(0.0 <= w.start <= w.end <= 20.0 + 1e-6 for w in words)
    assert all(0.0 <= w.probability <= 1.0 for w in words)
    starts = [w.start for w in words]
    assert starts == sorted(starts)                       # the DTW path is monotone
    for s in segs:
        if s.words:
            assert "".join(w.word for w in s.words) == s.text
            assert s.start <= s.words[0].end and s.words[-1].start <= s.end + 1e-6
