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


# ---- malformed files: whatever is wrong with an operator's model.bin, the reader answers with
# CT2FormatError - never a crash with another exception type, never a silent mis-load --------------------------------------
def _s(t):
    raw = t if isinstance(t, bytes) else t.encode()
    return struct.pack("<H", len(raw) + 1) + raw + b"\0"


def _var(name, arr, dtype_id=None, n_bytes=None, shape=None):
    arr = np.ascontiguousarray(arr)
    shape = arr.shape if shape is None else shape
    data = arr.tobytes()
    ids = {"float32": 0, "int8": 1, "int16": 2, "int32": 3, "float16": 4}
    return (_s(name) + struct.pack("<B", len(shape)) + b"".join(struct.pack("<I", x) for x in shape)
            + struct.pack("<B", ids[arr.dtype.name] if dtype_id is None else dtype_id)
            + struct.pack("<I", len(data) if n_bytes is None else n_bytes) + data)


def _file(variables=(), aliases=(), version=6, spec="WhisperSpec", n_vars=None, n_aliases=None):
    out = struct.pack("<I", version) + _s(spec) + struct.pack("<I", 3) + struct.pack("<I", len(variables) if n_vars is None else n_vars)
    out += b"".join(variables)
    out += struct.pack("<I", len(aliases) if n_aliases is None else n_aliases) + b"".join(_s(a) + _s(t) for a, t in aliases)
    return out


_F4 = np.arange(6, dtype="<f4").reshape(2, 3)
MALFORMED_BIN = {
    "empty file": (b"", "empty|mmap|truncated"),
    "header cut inside the version": (b"\x06\x00", "truncated"),
    "header cut inside the spec name": (struct.pack("<I", 6) + struct.pack("<H", 12) + b"Whisp", "truncated"),
    "variable count says more than the file holds": (_file([_var("a", _F4)], n_vars=3), "truncated"),
    "absurd variable count": (_file([], n_vars=0xFFFFFFFF), "truncated"),
    "data cut short": (_file([_var("a", _F4)])[:-20], "truncated"),
    "unknown dtype id": (_file([_var("a", _F4, dtype_id=9)]), "unknown dtype id 9"),
    "shape x itemsize != byte count": (_file([_var("a", _F4, shape=(2, 4))]), r"shape \(2, 4\) x 4 B != 24 B"),
    "byte count larger than the data": (_file([_var("a", _F4, n_bytes=4000)]), "truncated|!="),
    "rank byte is garbage": (_file([_s("a") + struct.pack("<B", 200) + b"\0" * 64]), "rank 200"),
    "name is not UTF-8": (_file([_var(b"\xff\xfe\xfa", _F4)]), "not UTF-8"),
    "the same variable twice": (_file([_var("a", _F4), _var("a", _F4)]), "appears twice"),
    "alias table cut short": (_file([_var("a", _F4)], aliases=[("b", "a")], n_aliases=2), "truncated"),
    "future binary version": (_file([_var("a", _F4)], version=4096), "binary version 4096"),
}


@pytest.mark.parametrize("what", sorted(MALFORMED_BIN))
def test_malformed_model_bin_always_raises_ct2_format_error(tmp_path, what):
    raw, match = MALFORMED_BIN[what]
    p = tmp_path / "model.bin"
    p.write_bytes(raw)
    with pytest.raises(ct2.CT2FormatError, match=match):
        ct2.read_model_bin(str(p))
    with pytest.raises(ct2.CT2FormatError):
        ct2.read_ct2_dir(str(tmp_path))


def _whisper_vars():
    return ct2.hf_to_ct2(_hf().items(), DIMS)


def _drop(name):
    def f(v, a):
        del v[name]
    return f


def _set(name, value):
    def f(v, a):
        v[name] = value
    return f


def _alias(alias, target):
    def f(v, a):
        a[alias] = target
    return f


_D = DIMS.d_model
MALFORMED_WHISPER = {
    # (edit of a valid WhisperSpec variable set, expected message)
    "no conv stem": (_drop("encoder/conv1/weight"), "no variable 'encoder/conv1/weight'"),
    "conv weight of the wrong rank": (_set("encoder/conv1/weight", np.zeros((_D, DIMS.n_mels), np.float32)), "rank-3"),
    "conv kernel of the wrong width": (_set("encoder/conv1/weight", np.zeros((_D, DIMS.n_mels, 5), np.float32)), "kernel width 5"),
    "head count that does not give head_dim 64": (_set("encoder/num_heads", np.asarray(3, np.int16)), "head_dim 64"),
    "alias whose target is absent": (lambda v, a: (v.pop("decoder/layer_norm/gamma"), a.__setitem__("decoder/layer_norm/gamma", "nowhere/gamma")),
                                     "missing 'decoder/layer_norm/gamma'"),
    "alias cycle": (lambda v, a: (v.pop("decoder/layer_norm/beta"), a.__setitem__("decoder/layer_norm/beta", "x"), a.__setitem__("x", "decoder/layer_norm/beta")),
                    "alias cycle"),
    "self-attention not fused (separate q only)": (_set("encoder/layer_0/self_attention/linear_0/weight", np.zeros((_D, _D), np.float32)),
                                                   "expected fused q/k/v"),
    "fused q/k/v weight with an unfused bias": (_set("decoder/layer_1/self_attention/linear_0/bias", np.zeros(_D, np.float32)),
                                                 r"linear_0/bias: expected fused q/k/v"),
    "linear_0 missing altogether": (_drop("decoder/layer_0/self_attention/linear_0/weight"), "missing 'decoder/layer_0/self_attention/linear_0/weight'"),
    "cross-attention k/v not fused": (_set("decoder/layer_0/attention/linear_1/weight", np.zeros((_D, _D), np.float32)), "expected fused k/v"),
    "cross-attention k/v bias of the wrong length": (_set("decoder/layer_0/attention/linear_1/bias", np.zeros(_D, np.float32)), r"linear_1/bias: expected fused k/v"),
    "int8 weight without its scale": (_set("encoder/layer_0/ffn/linear_0/weight", np.zeros((DIMS.ffn_dim, _D), np.int8)), "without a .*_scale"),
    "int8 scale of the wrong length": (lambda v, a: (v.__setitem__("encoder/layer_0/ffn/linear_0/weight", np.ones((DIMS.ffn_dim, _D), np.int8)),
                                                    v.__setitem__("encoder/layer_0/ffn/linear_0/weight_scale", np.ones(7, np.float32))), "_scale has 7 entries"),
    "int8 scale holding a zero": (lambda v, a: (v.__setitem__("encoder/layer_0/ffn/linear_0/weight", np.ones((DIMS.ffn_dim, _D), np.int8)),
                                               v.__setitem__("encoder/layer_0/ffn/linear_0/weight_scale", np.zeros(DIMS.ffn_dim, np.float32))), "zeros / non-finite"),
    "a layer norm without its beta": (_drop("encoder/layer_1/ffn/layer_norm/beta"), "missing 'encoder/layer_1/ffn/layer_norm/beta'"),
    "no decoder layers": (lambda v, a: [v.pop(k) for k in list(v) if k.startswith("decoder/layer_")], "no decoder/layer_0"),
    "untied output projection": (lambda v, a: (a.pop("decoder/projection/weight"), v.__setitem__("decoder/projection/weight", v["decoder/embeddings/weight"] + 1)),
                                 "untied output projection"),
}


@pytest.mark.parametrize("what", sorted(MALFORMED_WHISPER))
def test_malformed_whisper_spec_always_raises_ct2_format_error(tmp_path, what):
    edit, match = MALFORMED_WHISPER[what]
    variables, aliases = _whisper_vars()
    edit(variables, aliases)
    ct2.write_model_bin(str(tmp_path / "model.bin"), variables, aliases)
    with pytest.raises(ct2.CT2FormatError, match=match):
        dims, tensors, _ = ct2.read_ct2_dir(str(tmp_path))
        dict(tensors)                    # the tensor stream is lazy: consume it


def test_random_byte_damage_never_escapes_as_another_exception(tmp_path):
    """400 random single-byte corruptions and 100 random truncations of a valid file: the reader returns or raises
    CT2FormatError - nothing else (IndexError, struct.error, MemoryError, UnicodeDecodeError ...)."""
    variables, aliases = _whisper_vars()
    ct2.write_model_bin(str(tmp_path / "good.bin"), variables, aliases)
    good = (tmp_path / "good.bin").read_bytes()
    # the header / name / shape bytes are where damage matters: pick offsets from the first variable records and the alias table
    rng = np.random.default_rng(5)
    d = tmp_path / "m"; d.mkdir()
    hot = np.concatenate([rng.integers(0, 4096, size=250), rng.integers(len(good) - 200, len(good), size=150)])
    n_raised = 0
    for i, off in enumerate(hot):
        bad = bytearray(good)
        bad[int(off)] ^= int(rng.integers(1, 256))
        (d / "model.bin").write_bytes(bytes(bad))
        try:
            _, tensors, _ = ct2.read_ct2_dir(str(d))
            for _ in tensors:
                pass
        except ct2.CT2FormatError:
            n_raised += 1
    for cut in rng.integers(1, len(good), size=100):
        (d / "model.bin").write_bytes(good[:int(cut)])
        with pytest.raises(ct2.CT2FormatError):
            _, tensors, _ = ct2.read_ct2_dir(str(d))
            for _ in tensors:
                pass
    assert n_raised > 20      # the corruption did hit structure, not only payload bytes
