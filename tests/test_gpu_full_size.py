"""GPU, BASELINE.json's full configuration (whisper-large-v3 geometry, 32+32 layers, B = 32, bf16, 128 new tokens).
The oracle cannot run all of it in test time, so: (i) size-independent properties of the whole batch - the
logits-processor invariants on every emitted token, finite scores, bit-identical replays, clip-order equivariance;
(ii) eight ROWS of the B = 32 batch recomputed by the CPU oracle at full depth over 24 tokens (prompt logits within 0.08,
teacher-forced token equality under margin), also for a clip decoded alone and by beam(1); (iii) one clip end to end in f32 (1e-3) and bf16."""
import numpy as np
import pytest

from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, PRESETS

pytestmark = pytest.mark.gpu
B, N_NEW = 32, 128


@pytest.fixture(scope="module")
def eng():
    from taiwan_tongues_asr_ce_amd.engine import Engine
    dims = PRESETS["large-v3"]
    e = Engine(dims, COMPUTE_BF16, B)
    e.load_weights(synth.iter_weights(dims))
    yield e
    e.close()


def test_full_size_invariants(eng):
    e = eng
    st = e.special
    clips = [synth.noise_clip(i) if i % 3 else synth.tonal_clip(i) for i in range(B)]
    e.log_mel(clips, want_output=False)
    enc = e.encode(B, want_output=True)
    assert np.isfinite(enc).all() and enc.shape == (B, 1500, 1280)
    assert 0.5 < float(np.abs(enc).mean()) < 2.0                       # LayerNorm-scale output
    prompt = [st.sot, st.lang_zh, st.transcribe]
    opts = e.gen_opts(N_NEW, True, suppress_eot=True, check_interval=1 << 20)      # timestamps on: exercises every rule
    res = e.generate([prompt] * B, opts)
    sup = {opts.suppress[i] for i in range(opts.n_suppress)}
    assert all(len(t) == N_NEW for t in res.tokens)
    assert np.isfinite(res.sum_logprob).all() and (res.sum_logprob < 0).all()
    assert ((res.no_speech_prob >= 0) & (res.no_speech_prob <= 1)).all()
    for toks in res.tokens:
        assert all(0 <= t < e.dims.vocab for t in toks)
        assert not sup.intersection(toks) and st.eot not in toks and st.no_timestamps not in toks
        assert toks[0] >= st.timestamp_begin and toks[0] - st.timestamp_begin <= 50      # first token: timestamp <= 1.0 s
        last_ts = -1
        for i, t in enumerate(toks):
            if t >= st.timestamp_begin:
                assert t >= last_ts, "timestamps must not decrease"
                last_ts = t
            if i >= 2 and toks[i - 1] >= st.timestamp_begin and toks[i - 2] >= st.timestamp_begin:
                assert t < st.timestamp_begin, "a timestamp pair must be followed by text"
    # a replay is bit-identical: tokens, scores and no-speech probabilities (no float atomics in the bf16 path)
    res2 = e.generate([prompt] * B, opts)
    assert res2.tokens == res.tokens
    assert np.array_equal(res2.sum_logprob, res.sum_logprob) and np.array_equal(res2.no_speech_prob, res.no_speech_prob)
    # clip-order equivariance, exactly: every kernel of the path computes a row independently of its batch position
    e.log_mel(clips[::-1], want_output=False)
    e.encode(B)
    rev = e.generate([prompt] * B, opts)
    assert rev.tokens[::-1] == res.tokens
    assert np.array_equal(rev.sum_logprob[::-1], res.sum_logprob)
    assert np.array_equal(rev.no_speech_prob[::-1], res.no_speech_prob)
    # a clip alone (B = 1: split cross-attention, 256x128 GEMM tiles) and beam search with ONE hypothesis take other kernels
    # than the batch of 32, so the last bits of their logits differ and a near-tie of the random-init weights may flip: all
    # three routes are graded against the ORACLE below (test_full_depth_b32_rows_teacher_forced_against_the_oracle), not
    # against each other; here only their structural properties
    for b in (5, 17, 30):
        e.log_mel([clips[b]], want_output=False)
        e.encode(1)
        solo = e.generate([prompt], e.gen_opts(16, True, suppress_eot=True)).tokens[0]
        assert len(solo) == 16 and solo[0] >= st.timestamp_begin and not sup.intersection(solo)


@pytest.fixture(scope="module")
def oracle_full():
    """Full-depth large-v3 weights for the oracle, bf16-rounded (what the bf16 engine holds): one generation per module."""
    import torch
    from oracle import whisper_ref as R
    torch.set_grad_enabled(False)
    dims = PRESETS["large-v3"]
    return R.Dims(**dims.as_dict()), R.to_torch(synth.state_dict(dims), round_bf16=True)


def test_full_depth_b32_rows_teacher_forced_against_the_oracle(eng, oracle_full):
    """The MEASURED configuration held to the oracle (deepened in round 5 - VERDICT round 4, next #1):
    whisper-large-v3 geometry, 32 + 32 layers, B = 32 different clips, bf16 - i.e. the single-pass cross-attention kernel, the
    identity-page self-attention and the 32-row decode GEMMs exactly as bench.py runs them (EOT suppressed, no host poll).  EIGHT
    rows of the batch (first, last, six inside) are recomputed by the CPU oracle (bf16-rounded weights) as one batch of 8: the
    step-API logits of the four prompt positions are within 0.08, and under teacher forcing each of the first 24 greedy tokens of
    the B = 32 generate() - positions 4...27: across the KV-page boundary at 16 and three replays of the 8-step graph - is within
    0.15 of the oracle's best allowed logit and IS the oracle's token wherever its top-2 margin exceeds 0.16.  Two of the rows
    decoded ALONE (B = 1: frame-split cross-attention, other GEMM tiles) and by beam search with one hypothesis pass the same
    grading."""
    import torch
    from oracle import whisper_ref as R
    from oracle_checks import Graded, teacher_forced
    from taiwan_tongues_asr_ce_amd.engine import default_suppress
    e = eng
    st = e.special
    rd, Wb = oracle_full
    clips = [synth.noise_clip(i) if i % 3 else synth.tonal_clip(i) for i in range(B)]
    rows = (0, 3, 5, 11, 17, 22, 27, 31)
    n_graded = 24
    prompt = [st.sot, st.lang_zh, st.transcribe, st.no_timestamps]
    e.log_mel(clips, want_output=False)
    e.encode(B)
    e.decode_reset(B)
    step_logits = [e.decode_step([t] * B) for t in prompt]               # B = 32 rows through the step API (same kernels)
    opts = e.gen_opts(n_graded, False, suppress_eot=True, check_interval=1 << 20)
    res = e.generate([prompt] * B, opts)                                 # the benchmark's route: graph replay, K-split slabs
    assert all(len(t) == n_graded for t in res.tokens)
    rules = R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin,
                    suppress=default_suppress(st, rd.vocab), begin_suppress=[220, st.eot], timestamps=False, suppress_eot=True)
    rules_solo = R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin,
                         suppress=default_suppress(st, rd.vocab), begin_suppress=[220, st.eot], timestamps=False)
    mel = np.stack([R.log_mel(clips[b], e.dims.n_mels) for b in rows])
    total, solo_total = Graded(), Graded()
    from oracle_checks import prompt_state
    # the eight rows go through the oracle as ONE batch (the 3.6 GB of decoder weights are read once per position instead of
    # once per position and row); row k's slice of the batched state serves its own gradings
    from oracle_checks import encode_chunked
    enc_all = encode_chunked(mel, Wb, rd)
    start_all = prompt_state(prompt, enc_all, Wb, rd)

    def row_state(k):
        xkv, cache, last, per_pos = start_all
        xk = [tuple(t[k:k + 1] for t in layer) for layer in xkv]
        ck = R.SelfCache([t[k:k + 1] for t in cache.k], [t[k:k + 1] for t in cache.v])
        return xk, ck, last[k:k + 1], [p[k:k + 1] for p in per_pos]
    for k, b in enumerate(rows):
        for t, lg, want in zip(prompt, step_logits, start_all[3]):
            assert float(np.abs(lg[b] - want[k].numpy()).max()) < 0.08, (b, t)
    total.add(teacher_forced([res.tokens[b] for b in rows], prompt, enc_all, Wb, rd, rules, tol=0.15, margin=0.16,
                             n_check=n_graded, start=start_all))
    for k, b in enumerate(rows):
        if b in (5, 17):   # the same clip alone, greedy and beam(1)
            e.log_mel([clips[b]], want_output=False)
            e.encode(1)
            solo = e.generate([prompt], e.gen_opts(4, False)).tokens[0]
            beam1 = [t for t in e.generate_beam([prompt], 1, e.gen_opts(4, False)).tokens[0] if t != st.eot]
            start = row_state(k)
            n_both = min(len(solo), len(beam1))
            if n_both == len(solo) == len(beam1):   # the usual case: both routes graded in one batched pass of the oracle
                enc2 = torch.cat([enc_all[k:k + 1]] * 2)
                st2 = ([tuple(torch.cat([t, t]) for t in layer) for layer in start[0]],
                       R.SelfCache([torch.cat([t, t]) for t in start[1].k], [torch.cat([t, t]) for t in start[1].v]),
                       torch.cat([start[2]] * 2), None)
                solo_total.add(teacher_forced([solo, beam1], prompt, enc2, Wb, rd, rules_solo, tol=0.15, margin=0.16, start=st2))
            else:
                solo_total.add(teacher_forced([solo], prompt, enc_all[k:k + 1], Wb, rd, rules_solo, tol=0.15, margin=0.16, start=start))
                solo_total.add(teacher_forced([beam1], prompt, enc_all[k:k + 1], Wb, rd, rules_solo, tol=0.15, margin=0.16, start=start))
    e.log_mel(clips, want_output=False)                                   # leave the module engine with the batch resident
    e.encode(B)
    assert total.n_steps == len(rows) * n_graded and total.n_clear >= 0.6 * total.n_steps, total     # not vacuous
    assert solo_total.n_clear >= 4, solo_total


def test_full_depth_b32_whole_measured_decode_length_against_the_oracle(eng, oracle_full):
    """EXACTLY the measured configuration over its WHOLE decode length (round 5): whisper-large-v3 geometry, 32 + 32 layers, B = 32
    different clips, bf16, 4-token prompt + 128 greedy tokens with the benchmark's options (EOT suppressed, no host poll: 16 replays
    of the 8-step graph, KV pages crossed at 16 ... 128).  Eight rows of the batch are graded at EVERY one of their 128 positions
    with one causal pass of the full-depth CPU oracle per row (bf16-rounded weights): every choice within 0.15 of the oracle's
    best allowed logit and equal to the oracle's token wherever its top-2 margin exceeds 0.16; at least 60 % of the 1 024
    graded steps carry such a margin."""
    import torch
    from oracle import whisper_ref as R
    from oracle_checks import encode_chunked, teacher_forced_causal
    from taiwan_tongues_asr_ce_amd.engine import default_suppress
    e = eng
    st = e.special
    rd, Wb = oracle_full
    clips = [synth.noise_clip(i) if i % 3 else synth.tonal_clip(i) for i in range(B)]
    rows = (1, 5, 9, 14, 18, 23, 27, 30)
    prompt = [st.sot, st.lang_zh, st.transcribe, st.no_timestamps]
    e.log_mel(clips, want_output=False)
    e.encode(B)
    res = e.generate([prompt] * B, e.gen_opts(N_NEW, False, suppress_eot=True, check_interval=1 << 20))
    assert all(len(t) == N_NEW for t in res.tokens)
    rules = R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin,
                    suppress=default_suppress(st, rd.vocab), begin_suppress=[220, st.eot], timestamps=False, suppress_eot=True)
    enc_ref = encode_chunked(np.stack([R.log_mel(clips[b], e.dims.n_mels) for b in rows]), Wb, rd)
    g = teacher_forced_causal([res.tokens[b] for b in rows], prompt, enc_ref, Wb, rd, rules, tol=0.15, margin=0.16, rows_per_pass=4)
    assert g.n_steps == len(rows) * N_NEW and g.n_clear >= 0.6 * g.n_steps, g


def test_full_depth_f32_parity_one_clip():
    """The north-star tolerance at the FULL model: whisper-large-v3 geometry, all 32 + 32 layers, f32 compute mode,
    one 30-s clip — encoder output and the logits of the prompt positions within 1e-3 of the f32 CPU oracle, greedy
    tokens identical.  (~3 TFLOP on the host cores for the oracle, hence a single clip and three new tokens.)"""
    import torch
    from oracle import whisper_ref as R
    from taiwan_tongues_asr_ce_amd.config import COMPUTE_F32
    from taiwan_tongues_asr_ce_amd.engine import Engine, default_suppress
    torch.set_grad_enabled(False)
    dims = PRESETS["large-v3"]
    rd = R.Dims(**dims.as_dict())
    sd = synth.state_dict(dims)
    e = Engine(dims, COMPUTE_F32, 1)
    e.load_weights(sd.items())
    st = e.special
    clip = synth.tonal_clip(2)
    mel = e.log_mel([clip])
    enc = e.encode(1, want_output=True)
    prompt = [st.sot, st.lang_zh, st.transcribe, st.no_timestamps]
    e.decode_reset(1)
    step_logits = [e.decode_step([t]) for t in prompt]
    opts = e.gen_opts(3, False)
    res = e.generate([prompt], opts)
    e.close()
    W = R.to_torch(sd)
    del sd
    mel_ref = R.log_mel(clip, dims.n_mels)[None]
    np.testing.assert_allclose(mel, mel_ref, atol=2e-4)
    enc_ref = R.encoder_forward(torch.from_numpy(mel_ref), W, rd)
    assert float(np.abs(enc - enc_ref.numpy()).max()) < 1e-3
    xkv = R.cross_kv(enc_ref, W, rd)
    cache = R.SelfCache.empty(rd.dec_layers)
    for t, lg in zip(prompt, step_logits):
        want = R.decoder_forward(torch.full((1, 1), t), cache, xkv, W, rd)[:, 0].numpy()
        assert float(np.abs(lg - want).max()) < 1e-3
    rules = R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin,
                    suppress=default_suppress(st, rd.vocab), begin_suppress=[220, st.eot], timestamps=False)
    ref = R.greedy_decode(enc_ref, prompt, W, rd, rules, 3)
    assert res.tokens == ref.tokens


def test_full_depth_bf16_against_oracle_with_rounded_weights(oracle_full):
    """Same full model in the MEASURED mode (bf16): against the f32 oracle holding the bf16-rounded weights the encoder
    output stays within 0.06 (measured 0.023 max, 0.0026 mean on LayerNorm-scale values) and the prompt logits within
    0.08 (measured 0.032 on logits of std 1.8); under teacher forcing every greedy choice is within 0.15 of the oracle's
    best allowed logit, and IDENTICAL to the oracle's choice wherever the oracle's top-2 margin exceeds 2 x that tolerance
    (measured: identical argmax at all positions)."""
    import torch
    from oracle import whisper_ref as R
    from taiwan_tongues_asr_ce_amd.engine import Engine, default_suppress
    torch.set_grad_enabled(False)
    dims = PRESETS["large-v3"]
    rd, Wb = oracle_full
    e = Engine(dims, COMPUTE_BF16, 1)
    e.load_weights(synth.iter_weights(dims))
    st = e.special
    clip = synth.tonal_clip(2)
    e.log_mel([clip], want_output=False)
    enc = e.encode(1, want_output=True)
    prompt = [st.sot, st.lang_zh, st.transcribe, st.no_timestamps]
    e.decode_reset(1)
    step_logits = [e.decode_step([t]) for t in prompt]
    toks = e.generate([prompt], e.gen_opts(8, False)).tokens[0]
    e.close()
    enc_ref = R.encoder_forward(torch.from_numpy(R.log_mel(clip, dims.n_mels)[None]), Wb, rd)
    err = np.abs(enc - enc_ref.numpy())
    assert err.max() < 0.06 and err.mean() < 0.006
    xkv = R.cross_kv(enc_ref, Wb, rd)
    cache = R.SelfCache.empty(rd.dec_layers)
    logits = None
    for t, lg in zip(prompt, step_logits):
        logits = R.decoder_forward(torch.full((1, 1), t), cache, xkv, Wb, rd)[:, 0]
        assert float(np.abs(lg - logits.numpy()).max()) < 0.08
    rules = R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin,
                    suppress=default_suppress(st, rd.vocab), begin_suppress=[220, st.eot], timestamps=False)
    assert len(toks) == 8
    n_clear = 0
    for i, t in enumerate(toks):
        s = R.apply_rules(logits[0], toks[:i], rules)
        top2 = np.sort(np.asarray(s))[-2:]
        # token equality wherever the oracle's own top-2 margin exceeds 2 x the logit tolerance stated above (0.08)
        if top2[1] - top2[0] > 0.16:
            assert int(np.argmax(s)) == t, (i, t, int(np.argmax(s)))
            n_clear += 1
        assert s[t] > -np.inf and float(s.max() - s[t]) < 0.15, i
        logits = R.decoder_forward(torch.full((1, 1), t), cache, xkv, Wb, rd)[:, 0]
    assert n_clear >= 4, n_clear      # at least half of the 8 steps were held to token equality (measured: 7-8)


def test_full_depth_beam5_eight_clips_properties_and_oracle_scores(oracle_full):
    """Config C5's per-GPU share at FULL depth: whisper-large-v3, 32 + 32 layers, bf16, 8 clips x
    beam 5 = 40 decode rows (the shared-clip cross-attention kernel, re-indexed page tables, two row groups per weight stream),
    at Whisper's 30-s window and at the opt-in short window (150 positions = 3-s utterances, the streaming path).
    Properties: rule invariants on every returned hypothesis, finite scores, bit-identical replay, clip-order equivariance
    (a clip's 5 rows never interact with another clip's).  Oracle: the returned hypothesis of one clip is teacher-forced through
    the CPU oracle (bf16-rounded weights) - the engine's cumulative log-probability matches the oracle's within 0.1 per token."""
    import torch
    from oracle import whisper_ref as R
    from taiwan_tongues_asr_ce_amd.engine import Engine, default_suppress
    dims = PRESETS["large-v3"]
    rd, Wb = oracle_full
    A, beam, n_new = 8, 5, 10
    e = Engine(dims, COMPUTE_BF16, A * beam)
    e.load_weights(synth.iter_weights(dims))
    st = e.special
    prompt = [st.sot, st.lang_zh, st.transcribe]
    full = [synth.noise_clip(200 + i) if i % 2 else synth.tonal_clip(200 + i) for i in range(A)]
    short = [synth.noise_clip(300 + i, 48000) for i in range(A)]
    for n_ctx, clips in ((0, full), (150, short)):
        e.set_audio_ctx(n_ctx)
        e.log_mel(clips, want_output=False)
        e.encode(A)
        opts = e.gen_opts(n_new, True)
        sup = {opts.suppress[i] for i in range(opts.n_suppress)}
        res = e.generate_beam([prompt] * A, beam, opts)
        assert len(res.tokens) == A and np.isfinite(res.sum_logprob).all() and (res.sum_logprob <= 0).all()
        assert ((res.no_speech_prob >= 0) & (res.no_speech_prob <= 1)).all()
        for toks in res.tokens:
            assert 0 < len(toks) <= n_new and all(0 <= t < dims.vocab for t in toks)
            assert not sup.intersection(toks) and st.no_timestamps not in toks
            assert st.timestamp_begin <= toks[0] <= st.timestamp_begin + 50
            last = -1
            for i, t in enumerate(toks):
                if t >= st.timestamp_begin:
                    assert t >= last
                    last = t
                if i >= 2 and toks[i - 1] >= st.timestamp_begin and toks[i - 2] >= st.timestamp_begin:
                    assert t < st.timestamp_begin
        again = e.generate_beam([prompt] * A, beam, opts)
        assert again.tokens == res.tokens and np.array_equal(again.sum_logprob, res.sum_logprob)
        e.log_mel(clips[::-1], want_output=False)
        e.encode(A)
        rev = e.generate_beam([prompt] * A, beam, opts)
        assert rev.tokens[::-1] == res.tokens and np.array_equal(rev.sum_logprob[::-1], res.sum_logprob)
        if n_ctx == 0:     # oracle score of clip 3's hypothesis (one full-depth encoder pass on the host)
            a = 3
            rules = R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin,
                            suppress=default_suppress(st, rd.vocab), begin_suppress=[220, st.eot], timestamps=True)
            enc_ref = R.encoder_forward(torch.from_numpy(R.log_mel(clips[a], dims.n_mels)[None]), Wb, rd)
            xkv = R.cross_kv(enc_ref, Wb, rd)
            cache = R.SelfCache.empty(rd.dec_layers)
            logits = None
            for t in prompt:
                logits = R.decoder_forward(torch.full((1, 1), t), cache, xkv, Wb, rd)[:, 0]
            total = 0.0
            for i, t in enumerate(res.tokens[a]):
                s = R.apply_rules(logits[0], res.tokens[a][:i], rules)
                assert s[t] > -np.inf, (i, t)
                total += float(torch.log_softmax(s, dim=-1)[t])
                logits = R.decoder_forward(torch.full((1, 1), t), cache, xkv, Wb, rd)[:, 0]
            n_scored = len(res.tokens[a])
            # the engine's sum may include the final EOT's log-probability when the hypothesis ended (EOT is stripped from tokens)
            slack = 0.1 * (n_scored + 1)
            s_eot = float(torch.log_softmax(R.apply_rules(logits[0], res.tokens[a], rules), dim=-1)[st.eot])
            got = float(res.sum_logprob[a])
            assert min(abs(got - total), abs(got - (total + s_eot))) < slack, (got, total, s_eot)
    e.set_audio_ctx(0)
    e.close()
