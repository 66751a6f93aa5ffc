"""GPU: the reference-shaped surfaces (WhisperModel.transcribe, ASRInterface adapter) over the HIP engine."""
import asyncio
import types
import warnings

import numpy as np
import pytest
import torch

from oracle import whisper_ref as R
from taiwan_tongues_asr_ce_amd import synth
from taiwan_tongues_asr_ce_amd.config import PRESETS

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def model():
    from taiwan_tongues_asr_ce_amd.model import WhisperModel
    return WhisperModel("synthetic:tiny", device="cuda", compute_type="float32", max_batch=8)


def test_transcribe_signature_of_the_reference_call_sites(model):
    audio = synth.noise_clip(0)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        # exactly the kwargs of asr_core.py:159-167
        segments, info = model.transcribe(audio, language="zh", word_timestamps=False, vad_filter=True, beam_size=5,
                                          condition_on_previous_text=True, initial_prompt="")
        assert any("vad" in str(x.message).lower() for x in w) and not any("beam" in str(x.message) for x in w)
    assert info.language == "zh" and info.language_probability == 1.0 and abs(info.duration - 30.0) < 1e-6
    assert hasattr(segments, "__next__")  # lazy, like faster-whisper's generator
    segs = list(segments)
    assert all(hasattr(s, a) for s in segs for a in ("text", "start", "end", "words", "tokens", "avg_logprob"))
    assert all(0.0 <= s.start <= s.end <= 30.0 for s in segs)
    # first-window tokens == the oracle's greedy tokens for the same prompt / rules
    dims = R.Dims(**PRESETS["tiny"].as_dict())
    W = R.to_torch(synth.state_dict(PRESETS["tiny"]))
    st = model.special
    enc = R.encoder_forward(torch.from_numpy(R.log_mel(audio, 80))[None], W, dims)
    from taiwan_tongues_asr_ce_amd.engine import default_suppress
    rules = R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin,
                    suppress=default_suppress(st, dims.vocab), begin_suppress=[220, st.eot], timestamps=True)
    # beam_size=5 as the reference passes it: first window == the oracle's beam search (limit 40 tokens for CPU time)
    segments, _ = model.transcribe(audio, language="zh", beam_size=5, max_new_tokens=40, temperature=0.0)
    segs = list(segments)
    ref = R.beam_decode(enc, [st.sot, st.lang_zh, st.transcribe], W, dims, rules, 5, 40)
    got = [t for s in segs if s.seek == 0 for t in s.tokens]  # first 30-s window
    ref_toks = [t for t in ref.tokens[0] if t != st.eot]
    assert got == ref_toks[: len(got)] and len(got) > 0


def test_transcribe_errors_and_inputs(model, tmp_path):
    with pytest.raises(ValueError):
        model.transcribe(np.zeros((2, 16000), np.float32), language="zh", beam_size=1)
    segs, info = model.transcribe(np.zeros(0, np.float32), language="zh", beam_size=1, temperature=0.0)
    assert list(segs) == [] and info.duration == 0.0
    import wave
    p = str(tmp_path / "x.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes((synth.noise_clip(2, 48000) * 32767).astype("<i2").tobytes())
    segs, info = model.transcribe(p, beam_size=1, temperature=0.0)  # path input + language detection
    assert info.language in ("en", "zh") or len(info.language) <= 3
    assert 0.0 < info.language_probability <= 1.0 and abs(info.duration - 3.0) < 1e-3
    list(segs)


def test_long_audio_windows_advance(model):
    audio = np.concatenate([synth.noise_clip(7), synth.noise_clip(8, 160000)])  # 40 s -> >= 2 windows
    segs, info = model.transcribe(audio, language="zh", beam_size=1, without_timestamps=True, temperature=0.0)
    segs = list(segs)
    assert len({s.seek for s in segs}) >= 2 and segs[-1].end <= 40.0 + 1e-6


def test_batch_path_equals_single(model):
    clips = [synth.noise_clip(i) for i in range(3)] + [synth.noise_clip(9, 50000)]
    batch = model.transcribe_batch(clips, max_new_tokens=12)
    for c, want in zip(clips, batch):
        assert model.transcribe_batch([c], max_new_tokens=12)[0] == want


def test_asr_adapter_result_dict():
    from taiwan_tongues_asr_ce_amd.asr import ASRFactory
    asr = ASRFactory.create_asr_pipeline("faster_whisper", model_size="synthetic:tiny", compute_type="float16")
    assert (asr.device, asr.compute_type, asr.model_size) == ("cuda", "float16", "synthetic:tiny")
    pcm = (synth.noise_clip(4, 48000) * 32767).astype("<i2").tobytes()
    client = types.SimpleNamespace(scratch_buffer=bytearray(pcm), client_id="c1", last_start_time=1.5,
                                   get_file_name=lambda: "c1_0.wav")
    out = asyncio.run(asr.transcribe(client))
    assert out is None or set(out) == {"language", "language_probability", "final", "text", "duration", "words"}
    # the adapter really decodes with the reference's options (beam 5, faster_whisper_asr.py:139-149): its text is the text
    # of a direct transcribe(beam_size=5) call on the same model, and the model has the rows that beam needs
    assert asr.asr_pipeline.max_batch >= 5
    audio = np.frombuffer(pcm, dtype="<i2").astype(np.float32) / 32768.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        segs, _ = asr.asr_pipeline.transcribe(audio, language="zh", **asr.default_transcribe_kwargs)
        direct = " ".join(s.text.strip() for s in segs)
    assert (out["text"] if out else "") == direct
    asr.warm_up()


def test_beam_that_does_not_fit_is_refused_not_degraded():
    from taiwan_tongues_asr_ce_amd.model import WhisperModel
    m = WhisperModel("synthetic:micro", device="cuda", compute_type="float32", max_batch=2)
    with pytest.raises(ValueError, match="max_batch"):
        m.transcribe(np.zeros(1600, np.float32), language="zh", beam_size=5)
    with pytest.raises(ValueError, match="at most 7"):
        m.transcribe(np.zeros(1600, np.float32), language="zh", beam_size=9)
    from taiwan_tongues_asr_ce_amd.config import COMPUTE_BF16, COMPUTE_F16
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m16 = WhisperModel("synthetic:micro", device="cuda", compute_type="float16", max_batch=1)
        m8 = WhisperModel("synthetic:micro", device="cuda", compute_type="int8_bfloat16", max_batch=1)
    # "float16" - the reference's GPU setting - is a real fp16 mode (no substitution, no warning); the int8 variants say
    # that their weights stay 16-bit
    assert m16.engine.compute_type == COMPUTE_F16 and m8.engine.compute_type == COMPUTE_BF16
    assert not any("compute_type='float16'" in str(x.message) for x in w)
    assert any("int8 weights are not implemented" in str(x.message) for x in w)


def test_fallback_ladder_runs_and_reports_temperature(model):
    """Synthetic weights always fail the avg-logprob threshold, so the default ladder walks to a sampled result."""
    audio = synth.noise_clip(11, 160000)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        segs, _ = model.transcribe(audio, language="zh", beam_size=1, max_new_tokens=24)
        segs = list(segs)
        greedy, _ = model.transcribe(audio, language="zh", beam_size=1, max_new_tokens=24, temperature=0.0)
        greedy = list(greedy)
    assert all(s.temperature == 0.0 for s in greedy)
    assert segs and all(0.0 <= s.temperature <= 1.0 for s in segs)
    # with a relaxed threshold the first attempt is accepted
    ok, _ = model.transcribe(audio, language="zh", beam_size=1, max_new_tokens=24, log_prob_threshold=-100.0,
                             compression_ratio_threshold=None)
    assert [s.tokens for s in ok] == [s.tokens for s in greedy]


def test_batched_streaming_backend_coalesces_requests():
    from taiwan_tongues_asr_ce_amd.asr import ASRFactory
    asr = ASRFactory.create_asr_pipeline("mi355x_whisper_batched", model_size="synthetic:tiny", compute_type="float16",
                                         max_clips=4, max_wait_ms=50.0, beam_size=2)
    clients = [types.SimpleNamespace(scratch_buffer=bytearray((synth.noise_clip(20 + i, 48000) * 32767).astype("<i2").tobytes()),
                                     client_id=f"c{i}", last_start_time=0.0) for i in range(6)]

    async def run():
        res = await asyncio.gather(*[asr.transcribe(c) for c in clients])
        await asr.aclose()
        return res

    res = asyncio.run(run())
    assert len(res) == 6 and sum(asr.batches_run) == 6 and max(asr.batches_run) > 1  # requests shared engine passes
    for r in res:
        assert r is None or (set(r) == {"language", "language_probability", "final", "text", "duration", "words"}
                             and 0.0 <= r["duration"] <= 3.0 + 1e-6)
    # same audio alone or inside a batch gives the same text
    single = asr.asr_pipeline.transcribe_windows([pcm for pcm in [synth.noise_clip(20, 48000)]], beam_size=2,
                                                 initial_prompt="繁體中文")
    if res[0] is not None:
        quant = (synth.noise_clip(20, 48000) * 32767).astype("<i2").astype(np.float32) / 32768.0
        alone = asr.asr_pipeline.transcribe_windows([quant], beam_size=2, initial_prompt="繁體中文")
        assert alone[0][0] == res[0]["text"]


def test_ct2_directory_loads_like_the_same_weights_in_memory(tmp_path):
    """WhisperModel("models") on a CTranslate2 directory (fp16 model.bin, as `compute_type="float16"` deployments
    ship) produces the tokens of the identical weights loaded directly, and the folder tool runs on top of it."""
    import wave
    from taiwan_tongues_asr_ce_amd import batch_cli, ct2
    from taiwan_tongues_asr_ce_amd.model import WhisperModel
    dims = PRESETS["micro"]
    hf = {k: v.astype(np.float16).astype(np.float32) for k, v in synth.iter_weights(dims)}       # fp16-representable
    variables, aliases = ct2.hf_to_ct2(hf.items(), dims, dtype=np.float16)
    ct2.write_model_bin(str(tmp_path / "model.bin"), variables, aliases)
    (tmp_path / "config.json").write_text("{}", encoding="utf-8")
    m = WhisperModel(str(tmp_path), device="cuda", compute_type="float32", max_batch=8)   # the folder tool decodes with beam 5
    assert (m.dims.d_model, m.dims.enc_layers, m.dims.vocab) == (dims.d_model, dims.enc_layers, dims.vocab)
    eng = m.engine
    clip = synth.noise_clip(3)[: dims.n_frames * 160]
    eng.log_mel([clip], want_output=False)
    enc_ct2 = eng.encode(1, want_output=True)
    W = R.to_torch(hf)
    rd = R.Dims(**dims.as_dict())
    enc_ref = R.encoder_forward(torch.from_numpy(R.log_mel(clip, dims.n_mels, n_samples=dims.n_frames * 160))[None], W, rd).numpy()
    assert np.abs(enc_ct2 - enc_ref).max() < 1e-3
    # folder tool end to end on the real engine
    folder = tmp_path / "audio"; folder.mkdir()
    with wave.open(str(folder / "a.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes((clip * 32767).astype("<i2").tobytes())
    (folder / "a.txt").write_text("測試", encoding="utf-8")
    final = batch_cli.process_audio_folder(str(folder), model=m, output_json=str(tmp_path / "out.json"), log=lambda *_: None)
    assert final["summary"]["total_files"] == 1 and "error" not in final["detailed_results"][0]
    assert (folder / "a_asr.txt").exists()


def test_vad_filter_pipeline_with_the_energy_stand_in(model):
    """vad_filter=True + a speech-probability source: only the speech chunks reach the engine and every time comes
    back on the original time line (two bursts 10 s apart -> the second segment group is shifted by the removed pause)."""
    from taiwan_tongues_asr_ce_amd import vad
    a = np.zeros(30 * 16000, np.float32)
    a[: 4 * 16000] = synth.tonal_clip(0)[: 4 * 16000]
    a[14 * 16000: 18 * 16000] = synth.tonal_clip(1)[: 4 * 16000]
    chunks = vad.get_speech_timestamps(a, vad.VadOptions())
    assert len(chunks) == 2 and chunks[1]["start"] > 13 * 16000
    kw = dict(language="zh", beam_size=1, temperature=0.0, max_new_tokens=24)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        segs, info = model.transcribe(a, vad_filter=True, vad_parameters={"backend": "energy"}, **kw)
        segs = list(segs)
        assert any("stand-in" in str(x.message) for x in w)
    cut = vad.collect_chunks(a, chunks)
    ref, ref_info = model.transcribe(cut, vad_filter=False, **kw)
    ref = list(ref)
    assert abs(info.duration - 30.0) < 1e-6 and abs(info.duration_after_vad - len(cut) / 16000) < 1e-6 < info.duration_after_vad < 10
    assert [s.tokens for s in segs] == [s.tokens for s in ref] and len(segs) > 0
    m = vad.SpeechTimestampsMap(chunks)
    for s, r in zip(segs, ref):
        assert s.start == m.get_original_time(r.start) and s.end == m.get_original_time(r.end)
    assert all(s.end <= 18.5 for s in segs) and all(not (4.5 < s.start < 13.5) for s in segs)
    # an operator-supplied probability function takes precedence and needs no opt-in
    model.vad_speech_prob_fn = lambda audio: np.zeros(int(np.ceil(len(audio) / vad.WINDOW)), np.float32)
    try:
        segs, info = model.transcribe(a, vad_filter=True, **kw)
        assert list(segs) == [] and info.duration_after_vad == 0.0
    finally:
        model.vad_speech_prob_fn = None


def test_hf_directory_with_generation_config_alignment_heads(tmp_path):
    """WhisperModel on an HF-format directory (fp16 safetensors, generation_config.json with curated alignment heads):
    same tokens as the identical weights loaded in memory; word timestamps use the directory's heads."""
    import json
    from safetensors.numpy import save_file
    from taiwan_tongues_asr_ce_amd.model import WhisperModel
    d = PRESETS["micro"]
    hf = {k: v.astype(np.float16) for k, v in synth.iter_weights(d)}
    (tmp_path / "config.json").write_text(json.dumps(dict(
        num_mel_bins=d.n_mels, max_source_positions=d.n_audio_ctx, d_model=d.d_model, encoder_attention_heads=d.n_heads,
        encoder_ffn_dim=d.ffn_dim, encoder_layers=d.enc_layers, decoder_layers=d.dec_layers, vocab_size=d.vocab,
        max_target_positions=d.n_text_ctx)), encoding="utf-8")
    (tmp_path / "generation_config.json").write_text(json.dumps({"alignment_heads": [[1, 0], [1, 1]]}), encoding="utf-8")
    save_file(hf, str(tmp_path / "model.safetensors"))
    m = WhisperModel(str(tmp_path), device="cuda", compute_type="float32", max_batch=2)
    assert m.alignment_heads == [(1, 0), (1, 1)]
    clip = synth.noise_clip(4)[: d.n_frames * 160]
    eng = m.engine
    eng.log_mel([clip], want_output=False)
    enc = eng.encode(1, want_output=True)
    W = R.to_torch({k: v.astype(np.float32) for k, v in hf.items()})
    rd = R.Dims(**d.as_dict())
    enc_ref = R.encoder_forward(torch.from_numpy(R.log_mel(clip, d.n_mels, n_samples=d.n_frames * 160))[None], W, rd).numpy()
    assert np.abs(enc - enc_ref).max() < 1e-3
    st = m.special
    toks = [st.sot, st.lang_zh, st.transcribe, st.no_timestamps, 7, 9, 11, 13, st.eot]
    w, lp = eng.align(0, toks, m.alignment_heads)
    rw, rlp = R.alignment_weights(torch.from_numpy(enc_ref), toks, W, rd, m.alignment_heads, return_logprobs=True)
    assert np.abs(w - rw.numpy()).max() < 2e-5 and np.abs(lp - rlp.numpy()).max() < 2e-3


def test_transcribe_many_equals_file_by_file(model):
    """Several files in lock step through one engine pass (ragged-prompt beam search) == the same files transcribed
    one after the other: same segments, tokens and times; also through the fallback ladder and with word timestamps."""
    files = [np.concatenate([synth.tonal_clip(0), synth.noise_clip(1)[: 10 * 16000]]),      # 40 s: two windows
             np.concatenate([synth.noise_clip(2), synth.tonal_clip(3), synth.burst_clip(4)[: 15 * 16000]]),   # 75 s
             synth.tonal_clip(5)[: 20 * 16000]]                                               # 20 s: one short window
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for kw in (dict(beam_size=2, temperature=0.0, log_prob_threshold=None, max_new_tokens=20),
                   dict(beam_size=1, temperature=0.0, log_prob_threshold=None, max_new_tokens=16, word_timestamps=True),
                   dict(beam_size=2, temperature=(0.0, 0.4), best_of=2, max_new_tokens=12)):        # default thresholds: ladder
            many = model.transcribe_many(files, language="zh", **kw)
            assert len(many) == 3
            for audio, (segs, info) in zip(files, many):
                ref, ref_info = model.transcribe(audio, language="zh", **kw)
                ref = list(ref)
                assert abs(info.duration - ref_info.duration) < 1e-9
                assert [s.tokens for s in segs] == [s.tokens for s in ref], kw
                assert [(s.seek, s.start, s.end, s.temperature) for s in segs] == [(s.seek, s.start, s.end, s.temperature) for s in ref]
                if kw.get("word_timestamps"):
                    assert [[(w.word, w.start, w.end) for w in s.words] for s in segs] == \
                           [[(w.word, w.start, w.end) for w in s.words] for s in ref]


def test_unsupported_options_warn_and_unknown_ones_raise(model):
    audio = synth.noise_clip(0)[: 5 * 16000]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        segs, _ = model.transcribe(audio, language="zh", beam_size=1, temperature=0.0, max_new_tokens=4,
                                   no_repeat_ngram_size=3, repetition_penalty=1.2, length_penalty=1, patience=1.0)
        list(segs)
    msgs = [str(x.message) for x in w]
    assert any("no_repeat_ngram_size" in m for m in msgs) and any("repetition_penalty" in m for m in msgs)
    assert not any("length_penalty" in m or "patience" in m for m in msgs)      # neutral values are fine
    with pytest.raises(TypeError):
        model.transcribe(audio, language="zh", beam=5)


def test_hotwords_and_prefix_reach_the_engine(model):
    """hotwords / prefix are prompt-side options (faster-whisper get_prompt): the decode must equal the oracle's greedy
    search on the prompt `_prompt` builds, and differ from the plain run."""
    audio = synth.tonal_clip(6)[: 12 * 16000]
    st = model.special
    dims = R.Dims(**PRESETS["tiny"].as_dict())
    W = R.to_torch(synth.state_dict(PRESETS["tiny"]))
    # a 12-s recording: the features of the recording, zero-padded in FEATURE space to one window (faster-whisper pad_or_trim)
    enc = R.encoder_forward(torch.from_numpy(R.file_window(R.log_mel_file(audio, 80), 0))[None], W, dims)
    from taiwan_tongues_asr_ce_amd.engine import default_suppress
    rules = R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin,
                    suppress=default_suppress(st, dims.vocab), begin_suppress=[220, st.eot], timestamps=True)
    kw = dict(language="zh", beam_size=1, temperature=0.0, max_new_tokens=10, log_prob_threshold=None)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        plain = [t for s in model.transcribe(audio, **kw)[0] for t in s.tokens]
        for extra in (dict(hotwords="台灣 語音"), dict(prefix="今天")):
            got = [t for s in model.transcribe(audio, **kw, **extra)[0] if s.seek == 0 for t in s.tokens]
            hot = model.tokenizer.encode(" " + extra["hotwords"]) if "hotwords" in extra else None
            pre = model.tokenizer.encode(" " + extra["prefix"]) if "prefix" in extra else None
            prompt, _ = model._prompt(st.lang_zh, "transcribe", False, [], hot, pre)
            ref = [t for t in R.greedy_decode(enc, prompt, W, dims, rules, 10).tokens[0] if t != st.eot]
            assert got == ref[: len(got)] and len(got) > 0 and got != plain
