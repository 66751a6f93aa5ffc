"""CPU-only checks: the C-ABI library builds/loads and exports every symbol include/ttasr.h declares,
fails loudly without a GPU, and the host-side logic around it (prompt building, segment splitting,
sharding, tokenizer stub, adapter registry)."""
import ctypes
import os
import re

import numpy as np
import pytest

from taiwan_tongues_asr_ce_amd import _lib
from taiwan_tongues_asr_ce_amd.config import PRESETS, SpecialTokens

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "ttasr.h")).read()
    declared = sorted(set(re.findall(r"\b(ttasr_[a-z_0-9]+)\s*\(", hdr)))
    assert declared == sorted(_lib.SYMBOLS)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared:
        assert getattr(raw, s) is not None
    assert b"gfx950" in lib.ttasr_version()


def test_library_exports_exactly_the_header(lib):
    """The dynamic symbol table of libttasr.so defines the entry points of include/ttasr.h and NOTHING else: no C++ internal
    (kernel launchers, `__device_stub__*`, helpers), no HIP kernel handle.  The library is loaded into processes that also hold
    torch, RCCL and an integrator's other extensions; generic names in the global namespace would be a collision waiting for a
    host.  Built with -fvisibility=hidden + a linker version script (csrc/ttasr.map)."""
    import shutil
    import subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    defined = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert defined == sorted(_lib.SYMBOLS), sorted(set(defined) ^ set(_lib.SYMBOLS))
    kinds = {line.split()[-2] for line in out.splitlines() if line.strip()}
    assert kinds == {"T"}, kinds


def test_release_library_has_no_environment_switch(lib):
    """Every kernel-selection override is an explicit ttasr_set_option call; the shipped library holds no TTASR_* switch name
    (`strings libttasr.so | grep TTASR_` is empty).  -DTTASR_EXPERIMENTS builds bring the lab switches back."""
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"TTASR_" not in blob
    assert lib.ttasr_set_option(None, b"flash", 0) == -1       # NULL context: refused, no crash


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_lib.Config) == 12 * 4
    assert ctypes.sizeof(_lib.GenOpts) == 12 * 4 + 2 * ctypes.sizeof(ctypes.c_void_p)


def test_create_rejects_bad_geometry_and_no_gpu(lib):
    import torch
    h = ctypes.c_void_p()
    bad = _lib.Config(80, 1500, 384, 5, 1536, 4, 4, 51865, 448, 1, 1, 0)  # head_dim != 64
    assert lib.ttasr_create(ctypes.byref(bad), 0, ctypes.byref(h)) == -1
    assert b"head_dim" in lib.ttasr_last_error(None)
    if not torch.cuda.is_available():
        d = PRESETS["micro"]
        ok = _lib.Config(d.n_mels, d.n_audio_ctx, d.d_model, d.n_heads, d.ffn_dim, d.enc_layers, d.dec_layers, d.vocab,
                         d.n_text_ctx, 1, 1, 0)
        rc = lib.ttasr_create(ctypes.byref(ok), 0, ctypes.byref(h))
        assert rc == -2 and not h.value  # TTASR_E_HIP: fails loudly, no CPU fallback
        assert b"no CPU fallback" in lib.ttasr_last_error(None) or b"HIP" in lib.ttasr_last_error(None)
        from taiwan_tongues_asr_ce_amd.model import WhisperModel
        with pytest.raises(RuntimeError):
            WhisperModel("synthetic:micro", device="cuda")
        with pytest.raises(RuntimeError):
            WhisperModel("synthetic:micro", device="cpu")


def _bare_model(preset="tiny"):
    from taiwan_tongues_asr_ce_amd.model import WhisperModel
    m = object.__new__(WhisperModel)
    m.dims = PRESETS[preset]
    m.special = SpecialTokens.for_vocab(m.dims.vocab)
    m.is_multilingual = True
    return m


def test_prompt_building():
    m = _bare_model()
    st = m.special
    p, sot = m._prompt(st.lang_zh, "transcribe", False, [])
    assert p == [st.sot, st.lang_zh, st.transcribe] and sot == 0
    p, sot = m._prompt(st.lang_zh, "transcribe", True, list(range(1000, 1300)))
    assert p[0] == st.sot_prev and sot == 1 + 223 and p[sot] == st.sot and p[-1] == st.no_timestamps
    assert p[1:sot] == list(range(1300 - 223, 1300))  # at most n_text_ctx/2 - 1 previous tokens
    assert m._lang_token("zh") == 50260 and m._lang_token("en") == 50259


def test_split_segments_on_timestamp_pairs():
    m = _bare_model()
    tb, eot = m.special.timestamp_begin, m.special.eot
    toks = [tb + 0, 11, 12, tb + 100, tb + 100, 13, tb + 250, tb + 250, eot]
    segs, adv = m._split_segments(toks, 0, 3000, 0.0, False)
    assert [(round(a, 2), round(b, 2), t) for a, b, t in segs] == [
        (0.0, 2.0, [tb, 11, 12, tb + 100]), (2.0, 5.0, [tb + 100, 13, tb + 250])]
    assert adv == 500  # 5.0 s -> 500 frames
    # single trailing timestamp: the window is consumed whole
    segs, adv = m._split_segments([tb, 11, tb + 50, tb + 50, 12, tb + 700, eot], 0, 3000, 30.0, False)
    assert adv == 3000 and len(segs) == 2 and round(segs[1][1], 2) == 44.0
    # no timestamps at all
    segs, adv = m._split_segments([11, 12, eot], 0, 1000, 0.0, True)
    assert adv == 1000 and segs == [(0.0, 10.0, [11, 12])]
    assert m._split_segments([eot], 0, 3000, 0.0, False) == ([], 3000)


def test_shard_range_covers_everything():
    from taiwan_tongues_asr_ce_amd.dist import shard_range
    for n in (0, 1, 7, 32, 256, 257):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def test_tokenizer_stub_and_pcm_conversion():
    from taiwan_tongues_asr_ce_amd.asr import ASRFactory, pcm16_bytes_to_float
    from taiwan_tongues_asr_ce_amd.tokenizer import load_tokenizer
    tk = load_tokenizer(None, 51865)
    assert tk.decode(tk.encode("繁體中文 abc")) == "繁體中文 abc"
    assert tk.decode([50257, 50364]) == ""
    x = pcm16_bytes_to_float(np.array([0, 16384, -32768], dtype="<i2").tobytes())
    np.testing.assert_array_equal(x, np.array([0.0, 0.5, -1.0], dtype=np.float32))
    with pytest.raises(ValueError):
        ASRFactory.create_asr_pipeline("whisper_cpp")


def test_decode_audio_wav(tmp_path):
    import wave
    from taiwan_tongues_asr_ce_amd.model import decode_audio
    sr = 44100
    t = np.arange(sr) / sr
    st = np.stack([np.sin(2 * np.pi * 440 * t), np.sin(2 * np.pi * 440 * t)], axis=1)
    p = str(tmp_path / "a.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(sr)
        w.writeframes((st * 20000).astype("<i2").tobytes())
    x = decode_audio(p)
    assert x.dtype == np.float32 and abs(len(x) - 16000) <= 1
    f = np.fft.rfft(x[:16000] * np.hanning(len(x[:16000])))
    assert abs(int(np.argmax(np.abs(f))) - 440) <= 1


def test_decode_audio_wav_sample_widths(tmp_path):
    """8- / 16- / 24- / 32-bit PCM carry the same signal (24-bit studio files are common in the field; asr_core.py:118 globs *.wav)."""
    import wave
    from taiwan_tongues_asr_ce_amd.model import decode_audio
    sig = 0.6 * np.sin(2 * np.pi * 200 * np.arange(16000) / 16000) - 0.2
    ref = None
    for width in (2, 3, 4, 1):
        q = np.round(sig * (2 ** (8 * width - 1) - 1)).astype(np.int64)
        if width == 1:
            raw = (q + 128).astype(np.uint8).tobytes()
        elif width == 3:
            u = (q & 0xFFFFFF).astype(np.uint32)
            raw = np.stack([u & 255, (u >> 8) & 255, (u >> 16) & 255], axis=1).astype(np.uint8).tobytes()
        else:
            raw = q.astype("<i%d" % width).tobytes()
        p = str(tmp_path / f"w{width}.wav")
        with wave.open(p, "wb") as w:
            w.setnchannels(1); w.setsampwidth(width); w.setframerate(16000)
            w.writeframes(raw)
        x = decode_audio(p)
        assert x.dtype == np.float32 and len(x) == 16000
        np.testing.assert_allclose(x, sig, atol=2.0 ** -(8 * width - 2) + 1e-6)
        if ref is None:
            ref = x
        assert x.min() < -0.79 and x.max() > 0.39                                   # the negative half survived the sign extension


def test_decode_audio_float_and_extensible_wav(tmp_path):
    """IEEE-float (tag 3) and WAVE_FORMAT_EXTENSIBLE (tag 0xFFFE) files - what DAWs and multichannel writers emit, and what the
    stdlib `wave` reader of this Python refuses - decode to the same samples as their 16-bit twin."""
    import struct
    from taiwan_tongues_asr_ce_amd.model import decode_audio
    sig = (0.5 * np.sin(2 * np.pi * 300 * np.arange(8000) / 8000)).astype(np.float32)           # 8 kHz: also resampled
    st = np.stack([sig, sig], axis=1)

    def riff(fmt, data, junk=b""):
        chunks = junk + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(data)) + data
        return b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks
    f32 = riff(struct.pack("<HHIIHH", 3, 2, 8000, 8000 * 8, 8, 32), st.astype("<f4").tobytes(),
               junk=b"LIST" + struct.pack("<I", 5) + b"abcde\0")                                  # an odd-sized chunk before fmt
    guid_pcm = struct.pack("<H", 1) + bytes.fromhex("000000001000800000aa00389b71")
    ext = riff(struct.pack("<HHIIHHHHI", 0xFFFE, 2, 8000, 8000 * 4, 4, 16, 22, 16, 3) + guid_pcm,
               np.round(st * 32767).astype("<i2").tobytes())
    outs = []
    for name, blob in (("f32.wav", f32), ("ext.wav", ext)):
        p = tmp_path / name
        p.write_bytes(blob)
        x = decode_audio(str(p))
        assert x.dtype == np.float32 and abs(len(x) - 16000) <= 1
        outs.append(x)
    np.testing.assert_allclose(outs[0], outs[1], atol=2e-4)
    f = np.abs(np.fft.rfft(outs[0][:16000] * np.hanning(16000)))
    assert abs(int(np.argmax(f)) - 300) <= 1
    (tmp_path / "bad.wav").write_bytes(b"RIFFxxxxWAVEjunk")
    with pytest.raises(ValueError):
        decode_audio(str(tmp_path / "bad.wav"))


def test_synth_weights_are_order_independent():
    from taiwan_tongues_asr_ce_amd import synth
    d = PRESETS["micro"]
    specs = synth.tensor_specs(d)
    # HF's state dict has 90 entries / 818432 parameters for this geometry: ours omits the tied proj_out.weight
    assert len(specs) == 89 and sum(int(np.prod(s)) for _, s, _ in specs) == 818432
    a = synth.make_tensor(*specs[7])
    b = dict(synth.iter_weights(d))[specs[7][0]]
    np.testing.assert_array_equal(a, b)
    assert synth.noise_clip(3)[:4].tolist() == synth.noise_clip(3, 100)[:4].tolist()


def test_streaming_trigger_rule_table():
    """SilenceAtEndOfChunk (buffering_strategies.py:66-71,118-126) restated; the table is read off the reference
    code: with SimpleVAD (one segment [0, duration]) the first clause never fires, so the trigger is 'more than
    2 s after the offset', i.e. every second 1.5-s chunk at the default settings."""
    from taiwan_tongues_asr_ce_amd.streaming import chunk_ready, should_transcribe
    bps = 16000 * 2
    assert not chunk_ready(int(1.5 * bps), 1.5) and chunk_ready(int(1.5 * bps) + 2, 1.5)
    off = 0.1
    for seconds, vad_end, want in [(1.5, 1.5, False), (2.0, 2.0, False), (2.1, 2.1, False), (2.11, 2.11, True),
                                   (3.0, 3.0, True), (1.5, 1.0, True), (1.5, 1.39, True), (1.5, 1.41, False),
                                   (0.5, 0.0, True), (0.05, 0.0, False)]:
        assert should_transcribe(int(round(seconds * bps)), vad_end, off) == want, (seconds, vad_end)


def test_hf_directory_reader_safetensors_and_bin(tmp_path):
    """The HF-format `models/` directory (config.json + model.safetensors | pytorch_model.bin): geometry from the config,
    every tensor back as float32 under its HF name, the tied proj_out dropped."""
    import json
    import torch
    from safetensors.numpy import save_file
    from taiwan_tongues_asr_ce_amd import synth
    from taiwan_tongues_asr_ce_amd.model import _read_hf_dir
    d = PRESETS["micro"]
    sd = synth.state_dict(d)
    cfg = dict(num_mel_bins=d.n_mels, max_source_positions=d.n_audio_ctx, d_model=d.d_model, encoder_attention_heads=d.n_heads,
               encoder_ffn_dim=d.ffn_dim, encoder_layers=d.enc_layers, decoder_layers=d.dec_layers, vocab_size=d.vocab,
               max_target_positions=d.n_text_ctx)
    for kind in ("safetensors", "bin"):
        p = tmp_path / kind
        p.mkdir()
        (p / "config.json").write_text(json.dumps(cfg), encoding="utf-8")
        full = dict(sd)
        full["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
        if kind == "safetensors":
            save_file({k: v.astype(np.float16) for k, v in full.items()}, str(p / "model.safetensors"))   # fp16 checkpoint
        else:
            torch.save({k: torch.from_numpy(v) for k, v in full.items()}, str(p / "pytorch_model.bin"))
        dims, tensors = _read_hf_dir(str(p))
        assert (dims.n_mels, dims.n_audio_ctx, dims.d_model, dims.n_heads, dims.ffn_dim, dims.enc_layers, dims.dec_layers,
                dims.vocab, dims.n_text_ctx) == (d.n_mels, d.n_audio_ctx, d.d_model, d.n_heads, d.ffn_dim, d.enc_layers,
                                                 d.dec_layers, d.vocab, d.n_text_ctx)
        got = dict(tensors)
        assert set(got) == set(sd)
        for k, v in sd.items():
            assert got[k].dtype == np.float32
            ref = v.astype(np.float16).astype(np.float32) if kind == "safetensors" else v
            assert np.array_equal(got[k], ref), k
    with pytest.raises(FileNotFoundError):
        (tmp_path / "empty").mkdir()
        (tmp_path / "empty" / "config.json").write_text(json.dumps(cfg), encoding="utf-8")
        list(_read_hf_dir(str(tmp_path / "empty"))[1])


def test_prompt_with_hotwords_and_prefix():
    m = _bare_model()
    st = m.special
    hot, pre, prev = [900, 901], [700, 701, 702], [500, 501]
    p, sot = m._prompt(st.lang_zh, "transcribe", False, prev, hot, None)
    assert p == [st.sot_prev, 900, 901, 500, 501, st.sot, st.lang_zh, st.transcribe] and sot == 5
    p, sot = m._prompt(st.lang_zh, "transcribe", False, [], hot, None)          # hotwords alone still open with <|startofprev|>
    assert p == [st.sot_prev, 900, 901, st.sot, st.lang_zh, st.transcribe] and sot == 3
    p, sot = m._prompt(st.lang_zh, "transcribe", False, prev, hot, pre)         # a prefix drops the hotwords
    assert p == [st.sot_prev, 500, 501, st.sot, st.lang_zh, st.transcribe, st.timestamp_begin, 700, 701, 702] and sot == 3
    p, sot = m._prompt(st.lang_zh, "transcribe", True, [], None, pre)
    assert p == [st.sot, st.lang_zh, st.transcribe, st.no_timestamps, 700, 701, 702] and sot == 0
    p, _ = m._prompt(st.lang_zh, "transcribe", False, list(range(1000, 1400)), list(range(2000, 2400)), None)
    assert len(p) == 448 - 32 and p[1:224] == list(range(2000, 2223)) and p[224] == 1400 - (448 - 32 - 227)   # previous text gives way


def test_fallback_rule_matches_faster_whisper_generate_with_fallback():
    """faster-whisper retries a window at the next temperature when it is too repetitive or too unlikely, and cancels the
    retry only when the window is BOTH probably silent and unlikely (a silent-looking window with a fine
    log-probability but a high compression ratio must still be retried)."""
    from taiwan_tongues_asr_ce_amd.model import WhisperModel
    p = dict(no_speech_threshold=0.6, log_prob_threshold=-1.0, compression_ratio_threshold=2.4)
    f = WhisperModel._needs_fallback
    assert f(-0.5, 0.1, 1.0, p) is False                      # fine
    assert f(-0.5, 0.1, 3.0, p) is True                       # repetitive
    assert f(-1.5, 0.1, 1.0, p) is True                       # unlikely
    assert f(-1.5, 0.9, 1.0, p) is False                      # silent AND unlikely: accepted (and skipped later)
    assert f(-0.5, 0.9, 3.0, p) is True                       # silent-looking but likely and repetitive: retried
    assert f(-1.5, 0.9, 3.0, p) is False
    assert f(-1.5, 0.9, 3.0, dict(p, log_prob_threshold=None)) is True      # no log-prob threshold: nothing cancels
    assert f(-9.0, 0.0, 9.0, dict(no_speech_threshold=None, log_prob_threshold=None, compression_ratio_threshold=None)) is False


def test_default_suppress_list_has_startoflm_like_the_reference():
    from taiwan_tongues_asr_ce_amd.config import SpecialTokens
    from taiwan_tongues_asr_ce_amd.engine import default_suppress
    for vocab, sot_lm in ((51865, 50360), (51866, 50361)):
        st = SpecialTokens.for_vocab(vocab)
        assert st.sot_lm == sot_lm == st.sot_prev - 1
        ids = default_suppress(st, vocab)
        assert {st.transcribe, st.translate, st.sot, st.sot_prev, st.sot_lm} <= set(ids)
        assert st.eot not in ids and st.no_timestamps not in ids and all(i < st.timestamp_begin for i in ids)
    st = SpecialTokens.for_vocab(512)                          # synthetic vocabularies have no <|startoflm|>
    assert st.sot_lm == -1 and -1 not in default_suppress(st, 512)


class _CtxEngine:
    """Engine double that records the audio-context calls and answers the feature calls of the file-level path."""

    def __init__(self, full):
        self.full, self.ctx, self.calls = full, full, []

    def set_audio_ctx(self, n=0):
        self.ctx = n or self.full
        self.calls.append(("ctx", self.ctx))

    def log_mel_windows(self, audio, seeks, floor_max=None, want_output=False, want_max=False):
        self.calls.append(("windows", self.ctx, list(seeks)))
        return None, np.zeros(len(seeks), np.float32)


def test_file_feature_max_never_runs_on_a_reduced_audio_context():
    """transcribe_windows(audio_ctx=...) used to leave the engine on a short window; the next
    file-level call then took the whole-file maximum over a fraction of every 30-s window.  _file_feature_max restores the
    model's window first, and transcribe_windows restores it on the way out (also when a pass raises)."""
    m = _bare_model()
    m.max_batch = 4
    m.engine = _CtxEngine(m.dims.n_audio_ctx)
    m.engine.set_audio_ctx(150)                                   # what an earlier short-window pass left behind
    m._file_feature_max(np.zeros(16000 * 70, np.float32))
    feature_calls = [c for c in m.engine.calls if c[0] == "windows"]
    assert feature_calls and all(c[1] == m.dims.n_audio_ctx for c in feature_calls)
    assert feature_calls[0][2] == [0, 3000, 6000]

    class Boom(_CtxEngine):
        def log_mel(self, *a, **k):
            raise RuntimeError("boom")
    m.engine = Boom(m.dims.n_audio_ctx)
    m.n_window = 480000
    m.tokenizer = type("T", (), {"encode": staticmethod(lambda s: []), "decode": staticmethod(lambda t: "")})()
    with pytest.raises(RuntimeError):
        m.transcribe_windows([np.zeros(48000, np.float32)], audio_ctx=150, beam_size=1)
    assert m.engine.ctx == m.dims.n_audio_ctx and m.engine.calls[-1] == ("ctx", m.dims.n_audio_ctx)


def test_adapter_returns_none_for_a_vad_emptied_chunk_and_retries_only_on_opt_in():
    """faster_whisper_asr.py:186-198: the reference's retry re-opens a temp file it deleted at :179, always falls into
    `except: pass` and returns None - so a chunk the VAD empties yields None.  That is the default here; the
    retry is an explicit opt-in (`retry_without_vad=True`), live only when a VAD source exists, and its segments are dropped when
    the decoder itself marks them silent (no_speech_prob) or unlikely (avg_logprob)."""
    import asyncio
    from taiwan_tongues_asr_ce_amd.asr import MI355XWhisperASR

    class Seg:
        def __init__(self, text, no_speech_prob=0.0, avg_logprob=-0.1):
            self.text, self.start, self.end, self.words = text, 0.0, 1.0, None
            self.no_speech_prob, self.avg_logprob = no_speech_prob, avg_logprob

    class Pipe:
        def __init__(self, prob_fn, retry_segs):
            self.vad_speech_prob_fn, self.calls, self.retry_segs = prob_fn, [], retry_segs

        def transcribe(self, audio, **kw):
            self.calls.append(kw["vad_filter"])
            return iter([] if kw["vad_filter"] else self.retry_segs), type("I", (), {"language": "zh", "language_probability": 1.0})()

    client = type("C", (), {"scratch_buffer": (np.zeros(1600, "<i2")).tobytes(), "last_start_time": 0})()
    fn = lambda a: np.zeros(4)  # noqa: E731
    cases = (  # (VAD source, opt-in, segments of the retry) -> (transcribe calls, text)
        (fn, False, [Seg("重試成功")], [True], None),                      # default: the reference's effective behaviour
        (fn, True, [Seg("重試成功")], [True, False], "重試成功"),          # opt-in retry
        (None, True, [Seg("重試成功")], [True], None),                     # no VAD source: vad_filter=True kept the whole clip already
        (fn, True, [Seg("幻覺", no_speech_prob=0.9)], [True, False], None),    # the decoder calls the chunk silence
        (fn, True, [Seg("幻覺", avg_logprob=-1.7)], [True, False], None),      # a low-confidence guess
    )
    for prob_fn, opt_in, retry_segs, want_calls, want_text in cases:
        a = object.__new__(MI355XWhisperASR)
        a.asr_pipeline = Pipe(prob_fn, retry_segs)
        a.default_transcribe_kwargs = {"word_timestamps": False, "vad_filter": True, "beam_size": 5,
                                       "condition_on_previous_text": True, "initial_prompt": "繁體中文"}
        a.text_filter = None
        a.retry_without_vad, a.retry_no_speech_threshold, a.retry_logprob_threshold = opt_in, 0.6, -1.0
        out = asyncio.run(a.transcribe(client))
        assert a.asr_pipeline.calls == want_calls
        assert (out["text"] if out else None) == want_text
    # a speech-probability source in the adapter's default kwargs counts as a VAD source too
    a.asr_pipeline = Pipe(None, [Seg("x")])
    a.default_transcribe_kwargs["vad_speech_prob_fn"] = fn
    assert a._vad_is_active()


def test_every_set_option_key_is_documented_in_the_header():
    """include/ttasr.h lists the ttasr_set_option keys; the list must be exactly what the library accepts (engine_search.hip)."""
    src = open(os.path.join(ROOT, "taiwan_tongues_asr_ce_amd", "csrc", "engine_search.hip")).read()
    body = src[src.index("int set_option(ttasr_ctx* c, const std::string& key, int v) {"):]
    body = body[:body.index("\n}\n")]
    accepted = set(re.findall(r'key == "([a-z_0-9]+)"', body))
    hdr = open(os.path.join(ROOT, "include", "ttasr.h")).read()
    doc = hdr[hdr.index("kernel-selection overrides"):hdr.index("int ttasr_set_option")]
    documented = set(re.findall(r'"([a-z_0-9]+)"', doc))
    assert accepted and accepted == documented, (sorted(accepted - documented), sorted(documented - accepted))


def test_bench_reports_pmc_numbers_only_for_the_kernel_signature_they_were_taken_on():
    """a committed counter profile is quoted by bench.py only when it lists the signature (kernel name,
    template arguments, grid) of the kernel the run launched; a stale profile yields null plus a reason."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sig = "cross_attn_pipe_kernel<unsigned short, true, true, 3> grid 163840"
    xp = {"kernel": "cross_attn_pipe_kernel grid 163840", "signatures": [sig], "traffic_bytes_per_32row_launch": 246679059}
    tr, note = bench.xattn_traffic_from_profile(xp, sig, 32)
    assert tr == 246679059 and sig in note
    assert bench.xattn_traffic_from_profile(xp, sig, 16)[0] == round(246679059 / 2)
    tr, note = bench.xattn_traffic_from_profile(xp, "cross_attn_decode_kernel<unsigned short, false, 4, 8, true, 1> grid 163840", 32)
    assert tr is None and note.startswith("stale profile")
    old = {"kernel": "cross_attn_decode_kernel grid 163840", "traffic_bytes_per_32row_launch": 1}   # a round-3 file: no signatures
    assert bench.xattn_traffic_from_profile(old, sig, 32)[0] is None
    g0, g1 = "gemm_bf16_v3_kernel<unsigned short, 0> grid 1443840", "gemm_bf16_v3_kernel<unsigned short, 1> grid 1925120"
    kern = {"a": {"signatures": [g0], "mfma_busy_frac": 0.5}, "b": {"signatures": [g1], "mfma_busy_frac": 0.25}}
    busy, note = bench.pmc_busy_from_profile(kern, [g0, g1], [1.0, 1.0])
    assert note is None and abs(busy - 2.0 / (1 / 0.5 + 1 / 0.25)) < 1e-4
    busy, note = bench.pmc_busy_from_profile(kern, [g0, "gemm_bf16_v4_kernel<unsigned short, 1> grid 1"], [1.0, 1.0])
    assert busy is None and "stale profile" in note
    # the committed profile must carry signatures (refreshed whenever the dominant kernel changes)
    import json
    xp_now = json.load(open(os.path.join(ROOT, "profiles", "xattn_pmc.json")))
    assert xp_now.get("signatures"), "profiles/xattn_pmc.json predates the signature check: refresh it (tools/gpu_session.sh)"


def test_transcribe_groups_lanes_threads_and_errors():
    """Round 6: `WhisperModel.transcribe_groups` runs the groups on `pipeline_depth` worker threads, each bound to its own
    engine lane (lane 0 owns the weights, the others are created with share_weights_with=lane 0 on first use); results come back in
    group order, an exception inside a lane surfaces on the caller's thread, and depth 1 touches neither threads nor lanes."""
    import threading
    import time

    class _LaneEngine:
        made = []

        def __init__(self, *args, share_weights_with=None):
            self.owner = share_weights_with
            self.closed = False
            _LaneEngine.made.append(self)

        def close(self):
            self.closed = True

    m = _bare_model()
    m._engine_ctor, m._engine_args, m.pipeline_depth = _LaneEngine, (), 2
    m.engine = _LaneEngine()                       # lane 0 through the test setter
    seen = []

    def fake_many(group, **kw):
        seen.append((threading.current_thread().name, m.engine, tuple(group)))
        time.sleep(0.05 if group[0] % 2 == 0 else 0.01)         # groups finish out of order
        if group[0] < 0:
            raise RuntimeError("lane fault")
        return [("segments of", g) for g in group]
    m.transcribe_many = fake_many
    groups = [[0, 1], [2, 3], [4], [6, 7], [9]]
    out = m.transcribe_groups(groups, pipeline_depth=1)
    assert out == [[("segments of", g) for g in grp] for grp in groups]
    assert len(m._lanes) == 1 and all(t == threading.current_thread().name and e is m._lanes[0] for t, e, _ in seen)
    seen.clear()
    out2 = m.transcribe_groups(groups)                                       # the model's depth: 2
    assert out2 == out                                                       # group order, whatever finished first
    assert len(m._lanes) == 2 and m._lanes[1].owner is m._lanes[0] and m._lanes[0].owner is None
    by_thread = {}
    for t, e, g in seen:
        by_thread.setdefault(t, set()).add(id(e))
    assert set(by_thread) == {"ttasr-lane0", "ttasr-lane1"} and all(len(v) == 1 for v in by_thread.values())   # one engine per thread
    assert {e for t, e, g in seen if t == "ttasr-lane1"} == {m._lanes[1]}
    assert m.engine is m._lanes[0]                                            # the caller's thread still drives lane 0
    with pytest.raises(RuntimeError, match="lane fault"):
        m.transcribe_groups([[0], [-1], [2], [4]])
    assert m.transcribe_groups(groups[:2]) == out[:2]                         # usable afterwards
    m.close()
    assert all(e.closed for e in _LaneEngine.made[:2]) and m._lanes == []

    class _NoShare:
        def __init__(self, *args):
            pass
    m2 = _bare_model()
    m2._engine_ctor, m2._engine_args, m2.pipeline_depth = _NoShare, (), 2
    m2.engine = _NoShare()
    m2.transcribe_many = lambda group, **kw: list(group)
    with pytest.raises(RuntimeError, match="needs the HIP engine"):
        m2.transcribe_groups([[1], [2]])
