"""CTranslate2 model.bin reader.  Unpinned against a real file (none exists here): these tests
prove the reader inverts the writer, survives the format's variants (fp16 / bf16 / int8+scale, aliases, old
binary versions) and rejects damaged files loudly."""
import os
import struct

import numpy as np
import pytest

from taiwan_tongues_asr_ce_amd import ct2, synth
from taiwan_tongues_asr_ce_amd.config import PRESETS

DIMS = PRESETS["micro"]


def _hf():
    return {k: v for k, v in synth.iter_weights(DIMS)}


def test_round_trip_f32(tmp_path):
    hf = _hf()
    variables, aliases = ct2.hf_to_ct2(hf.items(), DIMS)
    ct2.write_model_bin(str(tmp_path / "model.bin"), variables, aliases)
    (tmp_path / "config.json").write_text('{"suppress_ids": [1, 2], "lang_ids": [5]}', encoding="utf-8")
    assert ct2.is_ct2_dir(str(tmp_path))
    dims, tensors, cfg = ct2.read_ct2_dir(str(tmp_path))
    assert (dims.n_mels, dims.n_audio_ctx, dims.d_model, dims.n_heads, dims.ffn_dim, dims.enc_layers, dims.dec_layers,
            dims.vocab, dims.n_text_ctx) == (DIMS.n_mels, DIMS.n_audio_ctx, DIMS.d_model, DIMS.n_heads, DIMS.ffn_dim,
                                             DIMS.enc_layers, DIMS.dec_layers, DIMS.vocab, DIMS.n_text_ctx)
    got = dict(tensors)
    assert set(got) == set(hf)                       # exactly the tensors ttasr_load_tensor expects
    for k in hf:
        assert got[k].dtype == np.float32 and np.array_equal(got[k], hf[k]), k
    assert cfg["suppress_ids"] == [1, 2]


def test_header_layout_is_the_published_one(tmp_path):
    p = str(tmp_path / "m.bin")
    ct2.write_model_bin(p, {"a/b": np.arange(6, dtype=np.float32).reshape(2, 3), "n": np.asarray(7, dtype=np.int16)}, {"c": "a/b"})
    raw = open(p, "rb").read()
    want = struct.pack("<I", 6) + struct.pack("<H", 12) + b"WhisperSpec\0" + struct.pack("<II", 3, 2)
    want += struct.pack("<H", 4) + b"a/b\0" + struct.pack("<BII", 2, 2, 3) + struct.pack("<BI", 0, 24) + np.arange(6, dtype="<f4").tobytes()
    want += struct.pack("<H", 2) + b"n\0" + struct.pack("<B", 0) + struct.pack("<BI", 2, 2) + struct.pack("<h", 7)
    want += struct.pack("<I", 1) + struct.pack("<H", 2) + b"c\0" + struct.pack("<H", 4) + b"a/b\0"
    assert raw == want
    spec, rev, v, al = ct2.read_model_bin(p)
    assert (spec, rev, al) == ("WhisperSpec", 3, {"c": "a/b"}) and v["n"].shape == () and int(v["n"]) == 7


def test_half_bf16_and_int8_variants(tmp_path):
    hf = _hf()
    variables, aliases = ct2.hf_to_ct2(hf.items(), DIMS, dtype=np.float16)
    # one weight stored as int8 with per-row scale, the way CTranslate2 quantises linear layers
    name = "decoder/layer_0/ffn/linear_0/weight"
    w = hf["model.decoder.layers.0.fc1.weight"]
    scale = (127.0 / np.abs(w).max(axis=1)).astype(np.float32)
    variables[name] = np.round(w * scale[:, None]).astype(np.int8)
    variables[name + "_scale"] = scale
    ct2.write_model_bin(str(tmp_path / "model.bin"), variables, aliases)
    # splice a bfloat16 variable in by hand (numpy has no bf16): dtype id 5
    bf = (hf["model.encoder.layer_norm.weight"].view(np.uint32) >> 16).astype("<u2")
    with open(tmp_path / "model.bin", "rb") as f:
        raw = bytearray(f.read())
    key = b"encoder/layer_norm/gamma\0"
    at = raw.index(key) + len(key)
    n = bf.size
    assert raw[at] == 1 and struct.unpack_from("<I", raw, at + 1)[0] == n and raw[at + 5] == 4      # rank 1, fp16
    raw[at + 5] = 5
    raw[at + 10:at + 10 + 2 * n] = bf.tobytes()
    with open(tmp_path / "model.bin", "wb") as f:
        f.write(raw)
    dims, tensors, _ = ct2.read_ct2_dir(str(tmp_path))
    got = dict(tensors)
    for k, v in hf.items():
        if k == "model.encoder.layer_norm.weight":
            continue                                  # truncated to bf16 above; checked bit-exactly below
        tol = 2e-3 * max(1.0, float(np.abs(v).max()))
        assert np.allclose(got[k], v, atol=tol), k
    q = got["model.decoder.layers.0.fc1.weight"]
    assert np.abs(q - w).max() <= 0.5 / scale.min() + 1e-7
    g = got["model.encoder.layer_norm.weight"]
    assert np.array_equal(g.view(np.uint32), hf["model.encoder.layer_norm.weight"].view(np.uint32) & 0xFFFF0000)


def test_old_binary_version_and_damage(tmp_path):
    def s(t):
        return struct.pack("<H", len(t) + 1) + t.encode() + b"\0"
    data = np.arange(4, dtype="<f4")
    old = struct.pack("<I", 3) + s("WhisperSpec") + struct.pack("<II", 1, 1) + s("x") + struct.pack("<BI", 1, 4) + struct.pack("<BI", 4, 4) + data.tobytes() + struct.pack("<I", 0)
    p = tmp_path / "old.bin"
    p.write_bytes(old)
    _, _, v, _ = ct2.read_model_bin(str(p))
    assert np.array_equal(v["x"], data)
    p.write_bytes(old[:-10])
    with pytest.raises(ct2.CT2FormatError, match="truncated"):
        ct2.read_model_bin(str(p))
    p.write_bytes(struct.pack("<I", 0) + old[4:])
    with pytest.raises(ct2.CT2FormatError, match="binary version"):
        ct2.read_model_bin(str(p))
    d = tmp_path / "notwhisper"; d.mkdir()
    ct2.write_model_bin(str(d / "model.bin"), {"x": data}, spec="TransformerSpec")
    with pytest.raises(ct2.CT2FormatError, match="TransformerSpec"):
        ct2.read_ct2_dir(str(d))
    ct2.write_model_bin(str(d / "model.bin"), {"x": data})
    with pytest.raises(ct2.CT2FormatError, match="no variable"):
        ct2.read_ct2_dir(str(d))


def test_reader_against_the_independent_int8_writer(tmp_path):
    """A full Whisper model written by tests/ct2_fixture_writer.py (no code shared with ct2.py: int8 linears with per-row
    `_scale`, float16 elsewhere, aliased + quantised output projection, variables sorted by name) reads back to exactly the
    de-quantised tensors, with the geometry inferred from the shapes alone."""
    from ct2_fixture_writer import write_whisper_ct2_int8
    hf = _hf()
    expect = write_whisper_ct2_int8(str(tmp_path), hf, DIMS.n_heads, DIMS.enc_layers, DIMS.dec_layers,
                                    config={"suppress_ids": [1, 2, 7], "suppress_ids_begin": [220, 50257], "alignment_heads": [[1, 0]]})
    dims, tensors, cfg = ct2.read_ct2_dir(str(tmp_path))
    assert (dims.n_mels, dims.n_audio_ctx, dims.d_model, dims.n_heads, dims.ffn_dim, dims.enc_layers, dims.dec_layers,
            dims.vocab, dims.n_text_ctx) == (DIMS.n_mels, DIMS.n_audio_ctx, DIMS.d_model, DIMS.n_heads, DIMS.ffn_dim,
                                             DIMS.enc_layers, DIMS.dec_layers, DIMS.vocab, DIMS.n_text_ctx)
    got = dict(tensors)
    assert set(got) == set(hf) == set(expect)
    for k in hf:
        assert got[k].dtype == np.float32 and got[k].shape == hf[k].shape, k
        np.testing.assert_array_equal(got[k], expect[k], err_msg=k)                   # exactly the de-quantised values
        err = np.abs(got[k] - hf[k]).max()
        assert err <= np.abs(hf[k]).max() * (1.0 / 127 if hf[k].ndim == 2 else 2e-3) + 1e-6, (k, err)   # and close to the originals
    assert cfg["alignment_heads"] == [[1, 0]]
