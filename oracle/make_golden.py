"""Generate tests/golden/*.npz from the reference's HF-Transformers Whisper path.

Run ONLY in the build container (needs `transformers`; the GPU box never runs this):
    python oracle/make_golden.py
Inputs are the seeded synthetic weights/clips of taiwan_tongues_asr_ce_amd.synth; outputs are what the
HF classes the reference trains/evaluates with (train_asr.py:33-43,518-545) compute on them:
WhisperFeatureExtractor, WhisperForConditionalGeneration (encoder, cached decoder), and the
Suppress*/WhisperTimeStamp logits processors.  Fixtures are data only (no reference source text).
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from transformers import WhisperConfig, WhisperFeatureExtractor, WhisperForConditionalGeneration  # noqa: E402
from transformers.generation.logits_process import (  # noqa: E402
    SuppressTokensAtBeginLogitsProcessor, SuppressTokensLogitsProcessor, WhisperTimeStampLogitsProcessor)

from taiwan_tongues_asr_ce_amd import synth  # noqa: E402
from taiwan_tongues_asr_ce_amd.config import PRESETS, SpecialTokens, NON_SPEECH_TOKENS_MULTI  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.manual_seed(0)
torch.set_grad_enabled(False)


def hf_model(dims, seed=0, dtype=torch.float32, profile="gauss"):
    st = SpecialTokens.for_vocab(dims.vocab)
    cfg = WhisperConfig(
        vocab_size=dims.vocab, num_mel_bins=dims.n_mels, d_model=dims.d_model,
        encoder_layers=dims.enc_layers, encoder_attention_heads=dims.n_heads, encoder_ffn_dim=dims.ffn_dim,
        decoder_layers=dims.dec_layers, decoder_attention_heads=dims.n_heads, decoder_ffn_dim=dims.ffn_dim,
        max_source_positions=dims.n_audio_ctx, max_target_positions=dims.n_text_ctx,
        pad_token_id=st.eot, bos_token_id=st.eot, eos_token_id=st.eot, decoder_start_token_id=st.sot,
        activation_function="gelu", attn_implementation="eager")
    model = WhisperForConditionalGeneration(cfg).eval()
    sd = {k: torch.from_numpy(v) for k, v in synth.iter_weights(dims, seed, profile)}
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("proj_out" in m for m in missing), missing
    return model.to(dtype), st


def hf_processors(st, begin_index, timestamps, suppress, begin_suppress, max_initial=50):
    procs = [SuppressTokensAtBeginLogitsProcessor(begin_suppress, begin_index=begin_index),
             SuppressTokensLogitsProcessor(suppress)]
    if timestamps:
        gc = types.SimpleNamespace(no_timestamps_token_id=st.no_timestamps, eos_token_id=st.eot,
                                   bos_token_id=st.eot, max_initial_timestamp_index=max_initial,
                                   _detect_timestamp_from_logprob=True)
        procs.append(WhisperTimeStampLogitsProcessor(gc, begin_index=begin_index))
    return procs


def hf_greedy(model, st, enc, prompt, n_steps, timestamps, suppress, begin_suppress):
    """Greedy loop over HF's cached decoder + HF's processor classes; stops rows at EOT."""
    B = enc.shape[0]
    procs = hf_processors(st, len(prompt), timestamps, suppress, begin_suppress)
    ids = torch.tensor([prompt] * B, dtype=torch.long)
    out = model(encoder_outputs=(enc,), decoder_input_ids=ids, use_cache=True)
    past = out.past_key_values
    prompt_logits = out.logits.float().clone()
    logits = out.logits[:, -1].float()
    raw, toks = [], []
    done = torch.zeros(B, dtype=torch.bool)
    for _ in range(n_steps):
        raw.append(logits.clone())
        s = logits
        for p in procs:
            s = p(ids, s)
        nxt = s.argmax(-1)
        nxt = torch.where(done, torch.full_like(nxt, st.eot), nxt)
        toks.append(nxt.clone())
        done |= nxt == st.eot
        ids = torch.cat([ids, nxt[:, None]], dim=1)
        if ids.shape[1] >= model.config.max_target_positions or bool(done.all()):
            break
        out = model(encoder_outputs=(enc,), decoder_input_ids=nxt[:, None], past_key_values=past, use_cache=True)
        past = out.past_key_values
        logits = out.logits[:, -1].float()
    return torch.stack(raw).numpy(), torch.stack(toks).numpy(), prompt_logits.numpy(), past


def golden_mel():
    clips = {"noise": synth.noise_clip(0), "tonal": synth.tonal_clip(0), "burst": synth.burst_clip(0),
             "short": synth.noise_clip(5, 176102)}  # warm_up.wav length after resampling
    out = {}
    for M in (80, 128):
        fe = WhisperFeatureExtractor(feature_size=M)
        out[f"filters_{M}"] = np.asarray(fe.mel_filters, dtype=np.float32)
        for name, pcm in clips.items():
            mel = fe(pcm, sampling_rate=16000, return_tensors="np")["input_features"][0]
            assert mel.shape == (M, 3000)
            out[f"{name}_{M}_stride7"] = mel[:, ::7].astype(np.float32)
            out[f"{name}_{M}_head"] = mel[:, :16].astype(np.float32)
            out[f"{name}_{M}_tail"] = mel[:, -16:].astype(np.float32)
            out[f"{name}_{M}_sum"] = np.array([mel.astype(np.float64).sum(), np.abs(mel).astype(np.float64).sum()])
    np.savez_compressed(os.path.join(OUT, "mel.npz"), **out)
    print("mel.npz", {k: v.shape for k, v in list(out.items())[:4]})


def golden_micro(profile="gauss", fname="micro.npz"):
    dims = PRESETS["micro"]
    model, st = hf_model(dims, profile=profile)
    fe = WhisperFeatureExtractor(feature_size=dims.n_mels, chunk_length=1)
    n = dims.n_frames * 160
    pcm = np.stack([synth.noise_clip(i, n) for i in range(3)])
    pcm[2, n // 3:] = 0.0
    mel = np.stack([fe(p, sampling_rate=16000, return_tensors="np")["input_features"][0] for p in pcm])
    eo = model.model.encoder(torch.from_numpy(mel), output_hidden_states=True)
    enc = eo.last_hidden_state
    prompt = [st.sot, st.lang_zh, st.transcribe]
    suppress = [1, 2, 7, st.sot, st.sot_prev, st.no_speech, st.translate, st.transcribe, st.lang_zh]
    begin_suppress = [5, st.eot]
    out = dict(pcm=pcm, mel=mel, enc=enc.numpy(), prompt=np.array(prompt), suppress=np.array(suppress),
               begin_suppress=np.array(begin_suppress))
    for i, h in enumerate(eo.hidden_states):
        out[f"enc_hidden_{i}"] = h.numpy()
    raw, toks, plog, past = hf_greedy(model, st, enc, prompt, 24, True, suppress, begin_suppress)
    out["ts_logits"], out["ts_tokens"], out["prompt_logits"] = raw, toks, plog
    cc = past.cross_attention_cache
    out["cross_k0"] = cc.layers[0].keys.numpy()
    out["cross_v0"] = cc.layers[0].values.numpy()
    out["cross_k1"] = cc.layers[1].keys.numpy()
    raw, toks, _, _ = hf_greedy(model, st, enc, prompt + [st.no_timestamps], 24, False, suppress, begin_suppress)
    out["nots_logits"], out["nots_tokens"] = raw, toks
    if profile != "gauss":   # the waveforms / features are micro.npz's; what this profile adds is how peaked the attention is
        out.pop("pcm")
        o = model.model.decoder(input_ids=torch.tensor([prompt + [7, 9, 11, 13]] * 3), encoder_hidden_states=enc, output_attentions=True)
        out["self_attn_pos0_share"] = np.array([float(a[:, :, 1:, 0].mean()) for a in o.attentions])    # mean weight on position 0
        out["cross_attn_max_share"] = np.array([float(a.max(-1).values.mean()) for a in o.cross_attentions])
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "tokens(ts)", toks.T.tolist()[0][:12], "enc", enc.shape, {k: v for k, v in out.items() if "share" in k})


def golden_tiny(profile="gauss", fname="tiny.npz"):
    dims = PRESETS["tiny"]
    model, st = hf_model(dims, profile=profile)
    fe = WhisperFeatureExtractor(feature_size=dims.n_mels)
    pcm = [synth.noise_clip(0), synth.tonal_clip(1)]
    mel = np.stack([fe(p, sampling_rate=16000, return_tensors="np")["input_features"][0] for p in pcm])
    enc = model.model.encoder(torch.from_numpy(mel)).last_hidden_state
    suppress = list(NON_SPEECH_TOKENS_MULTI) + [st.translate, st.transcribe, st.sot, st.sot_prev, st.no_speech]
    begin_suppress = [220, st.eot]
    out = dict(enc_stride=enc[:, ::25, ::3].numpy(), enc_mean=enc.mean((1, 2)).numpy(),
               enc_std=enc.std((1, 2)).numpy(), suppress=np.array(suppress),
               begin_suppress=np.array(begin_suppress))
    for tag, prompt, ts in (("ts", [st.sot, st.lang_zh, st.transcribe], True),
                            ("nots", [st.sot, st.lang_zh, st.transcribe, st.no_timestamps], False)):
        raw, toks, plog, _ = hf_greedy(model, st, enc, prompt, 20, ts, suppress, begin_suppress)
        top = torch.from_numpy(raw[0]).topk(32, dim=-1)
        out[f"{tag}_prompt"] = np.array(prompt)
        out[f"{tag}_tokens"] = toks
        out[f"{tag}_top_ids"] = top.indices.numpy()
        out[f"{tag}_top_vals"] = top.values.numpy()
        out[f"{tag}_logits_stride"] = raw[:, :, ::97]
        out[f"{tag}_no_speech"] = torch.softmax(torch.from_numpy(plog[:, 0]), -1)[:, st.no_speech].numpy()
        print("tiny", tag, toks.T.tolist())
    np.savez_compressed(os.path.join(OUT, fname), **out)


def golden_rules():
    """Processor known-answer vectors: random rows + hand-built histories -> HF-processed rows."""
    st = SpecialTokens.for_vocab(512)
    V = 512
    g = torch.Generator().manual_seed(7)
    tb, eot = st.timestamp_begin, st.eot
    histories = [[], [tb + 3], [tb + 3, 17], [17, 18], [tb + 1, 20, tb + 5], [tb + 1, 20, tb + 5, tb + 5],
                 [tb, tb], [tb + 2, 9, 10, tb + 40, tb + 40, 33], [tb + 63], [11, tb + 62, tb + 62]]
    suppress = [1, 2, 7, st.sot, st.no_speech]
    begin_suppress = [5, eot]
    rows, outs_ts, outs_nots, hist_pad = [], [], [], []
    for k, h in enumerate(histories):
        for variant in range(3):
            row = torch.randn(V, generator=g) * 3.0
            if variant == 1:
                row[tb:] += 4.0  # timestamp mass dominates -> force-timestamp branch
            if variant == 2:
                row[:tb] += 6.0
            prompt = [st.sot, st.lang_zh, st.transcribe]
            ids = torch.tensor([prompt + h], dtype=torch.long)
            for ts, store in ((True, outs_ts), (False, outs_nots)):
                s = row[None].clone()
                for p in hf_processors(st, len(prompt), ts, suppress, begin_suppress):
                    s = p(ids, s)
                store.append(s[0].numpy())
            rows.append(row.numpy())
            hist_pad.append(np.array(h + [-1] * (8 - len(h))))
    np.savez_compressed(os.path.join(OUT, "rules.npz"), rows=np.stack(rows), out_ts=np.stack(outs_ts),
                        out_nots=np.stack(outs_nots), hist=np.stack(hist_pad), suppress=np.array(suppress),
                        begin_suppress=np.array(begin_suppress))
    print("rules.npz", len(rows))


def golden_align():
    """Token-level timestamps (the `.words` path): HF cross-attentions of chosen alignment heads for a fixed token
    sequence and what `WhisperGenerationMixin._extract_token_timestamps` (median filter 7, DTW) makes of them."""
    from transformers.utils import ModelOutput
    out = {}
    for name, n_text in (("micro", 14), ("tiny", 40)):
        dims = PRESETS[name]
        model, st = hf_model(dims)
        fe = WhisperFeatureExtractor(feature_size=dims.n_mels, chunk_length=dims.n_frames // 100)
        n = dims.n_frames * 160
        pcm = synth.noise_clip(7, n) if name == "micro" else synth.tonal_clip(7)
        mel = fe(pcm, sampling_rate=16000, return_tensors="np")["input_features"]
        enc = model.model.encoder(torch.from_numpy(mel)).last_hidden_state
        rng = np.random.default_rng(11)
        hi = min(dims.vocab, st.eot) if st.eot > 300 else dims.vocab
        text = rng.integers(10, max(11, min(hi, 20000)), size=n_text).tolist()
        prefix = [st.sot, st.lang_zh, st.transcribe, st.no_timestamps]
        tokens = prefix + text + [st.eot]
        heads = [[dims.dec_layers - 1, 0], [dims.dec_layers - 1, dims.n_heads - 1], [dims.dec_layers // 2, 1 % dims.n_heads]]
        o = model.model.decoder(input_ids=torch.tensor([tokens]), encoder_hidden_states=enc, output_attentions=True)
        cross = o.cross_attentions                                   # tuple(L) of [1, H, n_tok, T]
        w = torch.stack([cross[l][0, h] for l, h in heads])
        lp = torch.log_softmax(model.proj_out(o.last_hidden_state)[0, :-1], dim=-1)
        tok_lp = lp[torch.arange(len(tokens) - 1), torch.tensor(tokens[1:])]
        go = ModelOutput(cross_attentions=[tuple(cross)], sequences=torch.tensor([tokens]))
        for tag, nf in (("full", dims.n_frames), ("short", (dims.n_frames * 2 // 3) // 2 * 2)):
            ts = model._extract_token_timestamps(go, heads, num_frames=nf, num_input_ids=len(prefix))
            out[f"{name}_ts_{tag}"] = ts[0].numpy()
            out[f"{name}_nf_{tag}"] = np.array(nf)
        out[f"{name}_pcm_seed"] = np.array(7)
        out[f"{name}_tokens"] = np.array(tokens)
        out[f"{name}_n_prefix"] = np.array(len(prefix))
        out[f"{name}_heads"] = np.array(heads)
        out[f"{name}_weights"] = w.numpy().astype(np.float16 if name == "tiny" else np.float32)
        out[f"{name}_token_logprob"] = tok_lp.numpy()
    np.savez_compressed(os.path.join(OUT, "align.npz"), **out)
    print("align.npz", {k: v.shape for k, v in out.items() if "ts_" in k or "weights" in k})


def golden_bf16():
    _golden_lowp(torch.bfloat16, "tiny_bf16.npz", "tiny bf16")


def golden_f16():
    """Golden set G7: as G5 with every parameter cast to float16 and HF's own fp16 arithmetic - exactly the precision regime the
    reference's GPU call sites ask for (compute_type="float16": asr_core.py:141, api/config.py:12, faster_whisper_asr.py:95).
    Gate for the engine's TTASR_COMPUTE_F16 mode."""
    _golden_lowp(torch.float16, "tiny_f16.npz", "tiny f16")


def _golden_lowp(dtype, fname, label, profile="gauss"):
    """Golden set G5: the SAME tiny-geometry model with every parameter cast to bfloat16 and HF's
    own bf16 arithmetic (residual stream, LayerNorm and logits all in bf16) - the precision regime of the reference's
    GPU path (asr_core.py:141 float16; this build measures in bf16).  Stored per greedy step and clip: the token HF picks,
    and the top-2 margin of the PROCESSED scores, so a bf16 engine can be held to token equality wherever the reference
    itself is not within rounding distance of a tie."""
    dims = PRESETS["tiny"]
    model, st = hf_model(dims, dtype=dtype, profile=profile)
    fe = WhisperFeatureExtractor(feature_size=dims.n_mels)
    pcm = [synth.noise_clip(0), synth.tonal_clip(1), synth.noise_clip(2), synth.burst_clip(3)]
    mel = np.stack([fe(p, sampling_rate=16000, return_tensors="np")["input_features"][0] for p in pcm])
    enc = model.model.encoder(torch.from_numpy(mel).to(dtype)).last_hidden_state
    suppress = list(NON_SPEECH_TOKENS_MULTI) + [st.translate, st.transcribe, st.sot, st.sot_prev, st.no_speech]
    begin_suppress = [220, st.eot]
    out = dict(clips=np.array(["noise0", "tonal1", "noise2", "burst3"]), suppress=np.array(suppress),
               begin_suppress=np.array(begin_suppress), enc_stride=enc[:, ::25, ::3].float().numpy())
    for tag, prompt, ts in (("ts", [st.sot, st.lang_zh, st.transcribe], True),
                            ("nots", [st.sot, st.lang_zh, st.transcribe, st.no_timestamps], False)):
        raw, toks, _, _ = hf_greedy(model, st, enc, prompt, 24, ts, suppress, begin_suppress)
        procs = hf_processors(st, len(prompt), ts, suppress, begin_suppress)
        ids = torch.tensor([prompt] * len(pcm), dtype=torch.long)
        margins, second = [], []
        for i in range(toks.shape[0]):
            s = torch.from_numpy(raw[i]).clone()
            for p in procs:
                s = p(ids, s)
            chosen = torch.from_numpy(toks[i])                        # bf16 logits hold exact ties: argmax = first maximum
            s_chosen = s.gather(1, chosen[:, None])[:, 0]
            assert (s_chosen == s.max(-1).values).all()
            others = s.scatter(1, chosen[:, None], float("-inf"))
            margins.append((s_chosen - others.max(-1).values).numpy())
            second.append(others.argmax(-1).numpy())
            ids = torch.cat([ids, torch.from_numpy(toks[i])[:, None]], dim=1)
        out[f"{tag}_prompt"] = np.array(prompt)
        out[f"{tag}_tokens"] = toks
        out[f"{tag}_margin"] = np.stack(margins).astype(np.float32)
        out[f"{tag}_runner_up"] = np.stack(second)
        out[f"{tag}_logits_stride"] = raw[:, :, ::97]
        print(label, tag, toks.T.tolist()[0][:10], "min margin", float(np.stack(margins).min()))
    np.savez_compressed(os.path.join(OUT, fname), **out)


def golden_beam():
    """Golden set G6: HF `generate(num_beams=5)` on the seeded tiny model (2 clips, 10 new tokens, the default suppress list,
    with and without the timestamp rules).  No hypothesis reaches <|endoftext|> in this regime, so what it pins is the
    evolution of the live beams and the final choice (highest cumulative log-probability); HF takes log_softmax BEFORE
    masking (no renormalisation over the allowed tokens), openai-whisper / CTranslate2 after, hence `avg_score` is
    compared with a tolerance and the finished-pool logic is pinned by the known-answer test in tests/."""
    from transformers import LogitsProcessorList
    from transformers.modeling_outputs import BaseModelOutput
    dims = PRESETS["tiny"]
    model, st = hf_model(dims)
    model.generation_config.suppress_tokens = None
    model.generation_config.begin_suppress_tokens = None
    fe = WhisperFeatureExtractor(feature_size=dims.n_mels)
    pcm = [synth.noise_clip(0), synth.tonal_clip(1)]
    mel = np.stack([fe(p, sampling_rate=16000, return_tensors="np")["input_features"][0] for p in pcm])
    enc = model.model.encoder(torch.from_numpy(mel)).last_hidden_state
    suppress = sorted(set(list(NON_SPEECH_TOKENS_MULTI) + [st.translate, st.transcribe, st.sot, st.sot_prev, st.no_speech]))
    out = dict(suppress=np.array(suppress), begin_suppress=np.array([220, st.eot]), beam=np.array(5), n_new=np.array(10))
    for tag, prompt, ts in (("nots", [st.sot, st.lang_zh, st.transcribe, st.no_timestamps], False),
                            ("ts", [st.sot, st.lang_zh, st.transcribe], True)):
        procs = LogitsProcessorList(hf_processors(st, len(prompt), ts, suppress, [220, st.eot]))
        r = model.generate(encoder_outputs=BaseModelOutput(last_hidden_state=enc), decoder_input_ids=torch.tensor([prompt] * 2),
                           num_beams=5, do_sample=False, max_new_tokens=10, length_penalty=1.0, early_stopping=False,
                           logits_processor=procs, return_dict_in_generate=True, output_scores=True, eos_token_id=st.eot,
                           pad_token_id=st.eot)
        out[f"{tag}_prompt"] = np.array(prompt)
        out[f"{tag}_tokens"] = r.sequences[:, len(prompt):].numpy()
        out[f"{tag}_avg_score"] = r.sequences_scores.numpy()
        print("beam", tag, r.sequences[:, len(prompt):].tolist(), r.sequences_scores.tolist())
    np.savez_compressed(os.path.join(OUT, "beam_hf.npz"), **out)


def golden_trained():
    """Round 6: the micro / tiny fixtures once more on the "trained" weight profile of synth.py
    (heavy-tailed matrices, LayerNorm outlier channels, two massive residual channels, an attention sink on decoder position 0):
    HF in f32 (micro: every intermediate; tiny: encoder slice, step logits, greedy tokens) and HF's own bf16 / fp16 arithmetic on
    the cast tiny model (tokens + top-2 margins per step, like G5 / G7)."""
    golden_micro("trained", "micro_trained.npz")
    golden_tiny("trained", "tiny_trained.npz")
    _golden_lowp(torch.bfloat16, "tiny_trained_bf16.npz", "tiny trained bf16", "trained")
    _golden_lowp(torch.float16, "tiny_trained_f16.npz", "tiny trained f16", "trained")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if "--trained-only" in sys.argv:   # round 6: adds the *_trained*.npz fixtures without touching the older ones
        golden_trained()
        sys.exit(0)
    if "--f16-only" in sys.argv:      # round 3: adds tests/golden/tiny_f16.npz without touching the older fixtures
        golden_f16()
        sys.exit(0)
    golden_mel()
    golden_rules()
    golden_micro()
    golden_tiny()
    golden_align()
    golden_bf16()
    golden_f16()
    golden_beam()
