"""GPU: decode batches wider than 32 rows (bf16 skinny GEMM with 2-4 row groups per weight stream).  Rows holding the
same clip must decode alike whatever their position in a 128-row batch, and agree with the oracle under teacher
forcing (same gate as the B <= 32 bf16 test: engine's choice within 0.15 of the oracle's best allowed logit).
120 rows = 24 files x beam 5 is what a context of the folder tool carries by default (round 6)."""
import numpy as np
import pytest
import torch

from oracle import whisper_ref as R
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, PRESETS

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def test_rows_beyond_32_decode_like_the_first_32():
    from taiwan_tongues_asr_ce_amd.engine import Engine, default_suppress
    pd = PRESETS["tiny"]
    dims = R.Dims(**pd.as_dict())
    e = Engine(pd, COMPUTE_BF16, 128)
    e.load_weights(synth.iter_weights(pd))
    st = e.special
    base = [synth.noise_clip(0), synth.tonal_clip(1), synth.burst_clip(2), synth.noise_clip(3), synth.tonal_clip(4), synth.noise_clip(5)]
    prompt = [st.sot, st.lang_zh, st.transcribe]
    opts = e.gen_opts(10, True)
    results = {}
    for B in (6, 40, 70, 96, 120, 128):                    # 1, 2, 3, 3, 4 and 4 row groups; 70 / 120 are not multiples of 32
        clips = [base[b % 6] for b in range(B)]
        e.log_mel(clips, want_output=False)
        e.encode(B)
        results[B] = e.generate([prompt] * B, opts).tokens
        assert len(results[B]) == B and all(len(t) > 0 for t in results[B])
    for B in (40, 70, 96, 120, 128):
        same = [results[B][b][:4] == results[6][b % 6][:4] for b in range(B)]
        assert np.mean(same) >= 0.9, (B, np.mean(same))
    # rows 90..95 of the 96-row batch and rows 122..127 of the 128-row batch (fourth row group) against the oracle (bf16-rounded
    # weights), teacher-forced
    Wb = R.to_torch(synth.state_dict(pd), round_bf16=True)
    rules = R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin,
                    suppress=default_suppress(st, dims.vocab), begin_suppress=[220, st.eot], timestamps=True)
    enc = R.encoder_forward(torch.from_numpy(np.stack([R.log_mel(c, pd.n_mels) for c in base])), Wb, dims)
    for wide, b in [(96, b) for b in range(90, 96)] + [(128, b) for b in range(122, 128)]:
        toks = results[wide][b]
        xkv = R.cross_kv(enc[b % 6:b % 6 + 1], Wb, dims)
        cache = R.SelfCache.empty(dims.dec_layers)
        logits = None
        for t in prompt:
            logits = R.decoder_forward(torch.full((1, 1), t), cache, xkv, Wb, dims)[:, 0]
        for i, t in enumerate(toks):
            s = R.apply_rules(logits[0], toks[:i], rules)
            assert s[t] > -np.inf and float(s.max() - s[t]) < 0.15, (wide, b, i)
            logits = R.decoder_forward(torch.full((1, 1), t), cache, xkv, Wb, dims)[:, 0]
    # beam search over 24 clips x 5 hypotheses = 120 rows (shared-clip cross-attention over 24 groups, four row groups in every
    # GEMM, the looped fc1 form) against the same clips searched 6 at a time (30 rows: one row group): the 16-bit forms differ
    # by batch width, so near-ties may resolve differently - nearly all clips must agree, and every score must be close
    opts_b = e.gen_opts(12, False)
    prompt_b = prompt + [st.no_timestamps]
    clips = [base[b % 6] for b in range(24)]
    e.log_mel(clips, want_output=False)
    e.encode(24)
    wide = e.generate_beam([prompt_b] * 24, 5, opts_b)
    e.log_mel(base, want_output=False)
    e.encode(6)
    narrow = e.generate_beam([prompt_b] * 6, 5, opts_b)
    agree = [wide.tokens[a] == narrow.tokens[a % 6] for a in range(24)]
    assert np.mean(agree) >= 0.8, np.mean(agree)
    for a in range(24):
        if agree[a]:
            assert abs(float(wide.sum_logprob[a]) - float(narrow.sum_logprob[a % 6])) < 0.05 * max(1, len(wide.tokens[a])), a
    e.close()


def test_measured_width_at_120_rows():
    """The same at large-v3 WIDTH (d 1280, 20 heads, ffn 5120: the K splits, block heights and the looped fc1 form that the folder
    tool's 120-row contexts actually launch), two layers each: rows of the fourth row group graded by the oracle under teacher
    forcing, and the 24-clip beam search against the same clips searched six at a time."""
    from taiwan_tongues_asr_ce_amd.engine import Engine, default_suppress
    from oracle_checks import encode_chunked
    pd = PRESETS["large-v3-w2"]
    dims = R.Dims(**pd.as_dict())
    sd = synth.state_dict(pd)
    e = Engine(pd, COMPUTE_BF16, 120)
    e.load_weights(sd.items())
    st = e.special
    base = [synth.noise_clip(0), synth.tonal_clip(1), synth.burst_clip(2), synth.noise_clip(3), synth.tonal_clip(4), synth.noise_clip(5)]
    prompt = [st.sot, st.lang_zh, st.transcribe, st.no_timestamps]
    clips = [base[b % 6] for b in range(120)]
    e.log_mel(clips, want_output=False)
    e.encode(120)
    opts = e.gen_opts(12, False, suppress_eot=True)
    got = e.generate([prompt] * 120, opts).tokens
    Wb = R.to_torch(sd, round_bf16=True)
    rules = R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin,
                    suppress=default_suppress(st, dims.vocab), begin_suppress=[220, st.eot], timestamps=False)
    rules.suppress_eot = True
    enc = encode_chunked(np.stack([R.log_mel(c, pd.n_mels) for c in base]), Wb, dims)
    worst = 0.0
    for b in (97, 103, 110, 119):                                      # fourth row group
        toks = got[b]
        xkv = R.cross_kv(enc[b % 6:b % 6 + 1], Wb, dims)
        cache = R.SelfCache.empty(dims.dec_layers)
        logits = None
        for t in prompt:
            logits = R.decoder_forward(torch.full((1, 1), t), cache, xkv, Wb, dims)[:, 0]
        for i, t in enumerate(toks):
            s = R.apply_rules(logits[0], toks[:i], rules)
            gap = float(s.max() - s[t])
            worst = max(worst, gap)
            assert s[t] > -np.inf and gap < 0.15, (b, i, gap)
            logits = R.decoder_forward(torch.full((1, 1), t), cache, xkv, Wb, dims)[:, 0]
    opts_b = e.gen_opts(12, False)
    e.log_mel(clips[:24], want_output=False)
    e.encode(24)
    wide = e.generate_beam([prompt] * 24, 5, opts_b)
    e.log_mel(base, want_output=False)
    e.encode(6)
    narrow = e.generate_beam([prompt] * 6, 5, opts_b)
    agree = [wide.tokens[a] == narrow.tokens[a % 6] for a in range(24)]
    assert np.mean(agree) >= 0.8, np.mean(agree)
    for a in range(24):
        if agree[a]:
            assert abs(float(wide.sum_logprob[a]) - float(narrow.sum_logprob[a % 6])) < 0.05 * max(1, len(wide.tokens[a])), a
    e.close()


@pytest.mark.parametrize("preset", ["tiny", "large-v3-w2"])
def test_lds_staged_activations_are_bit_identical_at_every_width(preset):
    """`dec_x_lds` (round 6): the decode GEMMs request their activation tile coalesced, park it in LDS and read the MFMA fragments
    back - the same fragments in the same MFMA order as the fragment loads straight from memory.  Held bit for bit on the step
    logits and on greedy / beam results for 1 ... 4 row groups (also a ragged last group), at two geometries (K = 384 / 1536 and
    1280 / 5120: different k-steps per wave, K splits, waves per workgroup)."""
    from taiwan_tongues_asr_ce_amd.engine import Engine
    pd = PRESETS[preset]
    e = Engine(pd, COMPUTE_BF16, 128)
    e.load_weights(synth.iter_weights(pd))
    st = e.special
    base = [synth.noise_clip(0), synth.tonal_clip(1), synth.burst_clip(2), synth.noise_clip(3), synth.tonal_clip(4)]
    prompt = [st.sot, st.lang_zh, st.transcribe, st.no_timestamps]
    for B in (1, 5, 32, 40, 70, 97, 128):
        e.log_mel([base[b % 5] for b in range(B)], want_output=False)
        e.encode(B)
        outs = []
        for on in (1, 0):
            e.set_option("dec_x_lds", on)
            e.decode_reset(B)
            lg = [e.decode_step([t] * B).copy() for t in prompt + [1234]]
            r = e.generate([prompt] * B, e.gen_opts(6, False, suppress_eot=True))
            outs.append((lg, r.tokens, r.sum_logprob.copy()))
        assert all(np.array_equal(a, b) for a, b in zip(outs[0][0], outs[1][0])), (preset, B)
        assert outs[0][1] == outs[1][1] and np.array_equal(outs[0][2], outs[1][2]), (preset, B)
    e.log_mel(base * 4, want_output=False)
    e.encode(20)
    beams = []
    for on in (1, 0):
        e.set_option("dec_x_lds", on)
        r = e.generate_beam([prompt] * 20, 5, e.gen_opts(8, False))
        beams.append((r.tokens, r.sum_logprob.copy()))
    assert beams[0][0] == beams[1][0] and np.array_equal(beams[0][1], beams[1][1]), preset
    e.set_option("dec_x_lds", 1)
    e.close()
