"""CTranslate2 `models/` directory reader: the on-disk format the reference deploys
(`model.bin` + `config.json` + `vocabulary.json`/`tokenizer.json` + `preprocessor_config.json`, named at
faster_whisper_asr.py:38, main.py:107, README.md:64-68).

PARITY UNPINNED.  CTranslate2 is a third-party dependency that is neither vendored under /root/reference nor
installable offline, and no `model.bin` exists on this machine, so this module restates the published
serialisation (CTranslate2 `ModelSpec._serialize`, binary version 6) and the published Whisper converter's
variable naming from memory; it is exercised only by a round trip through the writer below.  It must be
checked against an operator-supplied file before being trusted.

Layout (little endian):
    u32 binary_version                (>= 2 supported; 6 is current)
    str spec_name ("WhisperSpec"), u32 spec_revision
    u32 n_variables, then per variable:
        str name, u8 rank, u32 dims[rank], u8 dtype_id, u32 n_bytes, raw data      (version >= 4)
        str name, u8 rank, u32 dims[rank], u8 item_size, u32 n_items, raw data     (version  < 4)
    u32 n_aliases, then per alias: str alias, str target                           (version >= 3)
    str := u16 length (including the terminating NUL) + bytes + NUL
dtype ids: 0 float32, 1 int8, 2 int16, 3 int32, 4 float16, 5 bfloat16.

int8 weights carry a per-output-row `<name>_scale` (w ≈ q / scale); int16 weights a scalar scale.  The converter fuses
q/k/v of self-attention into `linear_0` ([3d, d], k bias = zeros), keeps cross-attention q in `linear_0` and fuses k/v into
`linear_1`; queries are scaled at run time, so weights are unscaled like HF's.
"""
from __future__ import annotations

import json
import os
import struct
from typing import Dict, Iterable, Iterator, Optional, Tuple

import numpy as np

from .config import WhisperDims

_DTYPES = {0: np.dtype("<f4"), 1: np.dtype("i1"), 2: np.dtype("<i2"), 3: np.dtype("<i4"), 4: np.dtype("<f2"), 5: "bf16"}
_DTYPE_IDS = {np.dtype("float32"): 0, np.dtype("int8"): 1, np.dtype("int16"): 2, np.dtype("int32"): 3, np.dtype("float16"): 4}


class CT2FormatError(ValueError):
    pass


class _Cursor:
    def __init__(self, buf):
        self.buf, self.pos = buf, 0

    def take(self, n: int):
        if n < 0 or self.pos + n > len(self.buf):
            raise CT2FormatError(f"truncated model.bin: need {n} bytes at offset {self.pos}, file has {len(self.buf)}")
        out = self.buf[self.pos:self.pos + n]
        self.pos += n
        return out

    def u8(self) -> int:
        return struct.unpack("<B", self.take(1))[0]

    def u16(self) -> int:
        return struct.unpack("<H", self.take(2))[0]

    def u32(self) -> int:
        return struct.unpack("<I", self.take(4))[0]

    def string(self) -> str:
        at = self.pos
        n = self.u16()
        raw = bytes(self.take(n))
        try:
            return raw.rstrip(b"\0").decode("utf-8")
        except UnicodeDecodeError as ex:
            raise CT2FormatError(f"string at offset {at} is not UTF-8 ({ex.reason}): not a CTranslate2 model.bin?") from None


def _bf16_to_f32(raw: np.ndarray) -> np.ndarray:
    return (raw.astype(np.uint32) << 16).view(np.float32)


def read_model_bin(path: str) -> Tuple[str, int, Dict[str, np.ndarray], Dict[str, str]]:
    """-> (spec name, spec revision, {variable: array in its stored dtype (bf16 widened to f32)}, {alias: target})."""
    try:
        buf = np.memmap(path, dtype=np.uint8, mode="r")
    except ValueError as ex:          # an empty file cannot be mapped
        raise CT2FormatError(f"{path}: {ex}") from None
    cur = _Cursor(buf)
    version = cur.u32()
    if version < 2 or version > 64:
        raise CT2FormatError(f"{path}: binary version {version} is not a CTranslate2 model this reader understands")
    spec, revision = cur.string(), cur.u32()
    variables: Dict[str, np.ndarray] = {}
    for _ in range(cur.u32()):
        name = cur.string()
        if name in variables:
            raise CT2FormatError(f"variable {name!r} appears twice")
        rank = cur.u8()
        if rank > 8:
            raise CT2FormatError(f"{name}: rank {rank} (CTranslate2 tensors have at most a few dimensions: damaged header?)")
        shape = tuple(cur.u32() for _ in range(rank))
        if version >= 4:
            dtype_id, n_bytes = cur.u8(), cur.u32()
            if dtype_id not in _DTYPES:
                raise CT2FormatError(f"{name}: unknown dtype id {dtype_id}")
            dt = _DTYPES[dtype_id]
        else:
            item, n_items = cur.u8(), cur.u32()
            n_bytes = item * n_items
            dt = {4: np.dtype("<f4"), 2: np.dtype("<i2"), 1: np.dtype("i1")}.get(item)
            if dt is None:
                raise CT2FormatError(f"{name}: item size {item}")
        raw = cur.take(n_bytes)
        itemsize = 2 if isinstance(dt, str) else dt.itemsize
        count = int(np.prod(shape, dtype=np.int64)) if rank else 1
        if count * itemsize != n_bytes:
            raise CT2FormatError(f"{name}: shape {shape} x {itemsize} B != {n_bytes} B")
        if isinstance(dt, str):
            arr = _bf16_to_f32(np.frombuffer(raw, dtype="<u2")).reshape(shape)
        else:
            arr = np.frombuffer(raw, dtype=dt).reshape(shape)
        variables[name] = arr
    aliases: Dict[str, str] = {}
    if version >= 3:
        for _ in range(cur.u32()):
            alias = cur.string()
            aliases[alias] = cur.string()
    return spec, revision, variables, aliases


def write_model_bin(path: str, variables: Dict[str, np.ndarray], aliases: Optional[Dict[str, str]] = None,
                    spec: str = "WhisperSpec", revision: int = 3, version: int = 6) -> None:
    """The inverse of `read_model_bin` (tests, and exporting a synthetic checkpoint in the deployed format)."""
    def s(text: str) -> bytes:
        raw = text.encode("utf-8")
        return struct.pack("<H", len(raw) + 1) + raw + b"\0"
    with open(path, "wb") as f:
        f.write(struct.pack("<I", version) + s(spec) + struct.pack("<I", revision) + struct.pack("<I", len(variables)))
        for name, arr in variables.items():
            arr = np.asarray(arr)
            data = np.ascontiguousarray(arr).tobytes()
            f.write(s(name) + struct.pack("<B", arr.ndim) + b"".join(struct.pack("<I", d) for d in arr.shape))
            f.write(struct.pack("<B", _DTYPE_IDS[arr.dtype]) + struct.pack("<I", len(data)) + data)
        aliases = aliases or {}
        f.write(struct.pack("<I", len(aliases)))
        for a, t in aliases.items():
            f.write(s(a) + s(t))


# ---- CT2 Whisper variables -> the HF state-dict names ttasr_load_tensor takes ------------------------------------

def _dense(variables: Dict[str, np.ndarray], aliases: Dict[str, str], name: str) -> Optional[np.ndarray]:
    """Float32 value of a possibly aliased / quantised variable."""
    seen = set()
    while name in aliases:          # the format allows an alias of an alias; a cycle is damage
        if name in seen:
            raise CT2FormatError(f"alias cycle through {name!r}")
        seen.add(name)
        name = aliases[name]
    if name not in variables:
        return None
    v = variables[name]
    if v.dtype == np.int8 and name + "_scale" in variables:
        scale = np.asarray(variables[name + "_scale"], dtype=np.float32)
        if scale.size not in (1, v.shape[0] if v.ndim else 1):
            raise CT2FormatError(f"{name}_scale has {scale.size} entries for {v.shape[0] if v.ndim else 1} output rows")
        if not np.all(np.isfinite(scale)) or np.any(scale == 0):
            raise CT2FormatError(f"{name}_scale holds zeros / non-finite values")
        return v.astype(np.float32) / scale.reshape((-1,) + (1,) * (v.ndim - 1))
    if v.dtype in (np.int8, np.int16, np.int32) and v.ndim >= 2 and name + "_scale" not in variables:
        raise CT2FormatError(f"{name} is stored as {v.dtype} without a {name}_scale variable: cannot de-quantise")
    if v.dtype == np.int16 and name + "_scale" in variables:
        return v.astype(np.float32) / float(np.asarray(variables[name + "_scale"], dtype=np.float32).reshape(-1)[0])
    return np.asarray(v, dtype=np.float32)


def infer_dims(variables: Dict[str, np.ndarray], aliases: Dict[str, str], name: str = "ct2") -> WhisperDims:
    """CT2's config.json holds no geometry; it is all in the tensor shapes."""
    def shape(n):
        n = aliases.get(n, n)
        if n not in variables:
            raise CT2FormatError(f"model.bin has no variable {n!r}: not a Whisper model?")
        return variables[n].shape
    def shape_of_rank(n, rank):
        sh = shape(n)
        if len(sh) != rank or any(x <= 0 for x in sh):
            raise CT2FormatError(f"{n}: expected a rank-{rank} tensor with positive dimensions, got shape {sh}")
        return sh
    d_model, n_mels, _k = shape_of_rank("encoder/conv1/weight", 3)
    if _k != 3:
        raise CT2FormatError(f"encoder/conv1/weight: kernel width {_k}, Whisper's stem has 3")
    n_audio_ctx = shape_of_rank("encoder/position_encodings/encodings", 2)[0]
    vocab = shape_of_rank("decoder/embeddings/weight", 2)[0]
    n_text_ctx = shape_of_rank("decoder/position_encodings/encodings", 2)[0]
    ffn = shape_of_rank("encoder/layer_0/ffn/linear_0/weight", 2)[0]
    enc_layers = dec_layers = 0
    while f"encoder/layer_{enc_layers}/ffn/linear_0/weight" in variables:
        enc_layers += 1
    while f"decoder/layer_{dec_layers}/ffn/linear_0/weight" in variables:
        dec_layers += 1
    heads = variables.get("encoder/num_heads")
    n_heads = int(np.asarray(heads).reshape(-1)[0]) if heads is not None and np.asarray(heads).size else d_model // 64
    if n_heads <= 0 or d_model % n_heads or d_model // n_heads != 64:
        raise CT2FormatError(f"{n_heads} heads for d_model {d_model}: every Whisper checkpoint has head_dim 64")
    if dec_layers == 0:
        raise CT2FormatError("model.bin has no decoder/layer_0: not a Whisper model?")
    return WhisperDims(name, n_mels, n_audio_ctx, d_model, n_heads, ffn, enc_layers, dec_layers, vocab, n_text_ctx)


def iter_hf_tensors(variables: Dict[str, np.ndarray], aliases: Dict[str, str], dims: WhisperDims) -> Iterator[Tuple[str, np.ndarray]]:
    d = dims.d_model

    def get(n):
        v = _dense(variables, aliases, n)
        if v is None:
            raise CT2FormatError(f"model.bin is missing {n!r}")
        return v

    def linear(ct2, hf, bias=True):
        yield hf + ".weight", get(ct2 + "/weight")
        if bias:
            yield hf + ".bias", get(ct2 + "/bias")

    def norm(ct2, hf):
        yield hf + ".weight", get(ct2 + "/gamma")
        yield hf + ".bias", get(ct2 + "/beta")

    def self_attention(ct2, hf):
        w, b = get(ct2 + "/linear_0/weight"), get(ct2 + "/linear_0/bias")
        if w.shape != (3 * d, d):
            raise CT2FormatError(f"{ct2}/linear_0/weight: expected fused q/k/v {(3 * d, d)}, got {w.shape}")
        if b.shape != (3 * d,):
            raise CT2FormatError(f"{ct2}/linear_0/bias: expected fused q/k/v ({3 * d},), got {b.shape}")
        yield hf + ".q_proj.weight", w[:d]
        yield hf + ".q_proj.bias", b[:d]
        yield hf + ".k_proj.weight", w[d:2 * d]        # Whisper's k projection has no bias (the fused slot is zeros)
        yield hf + ".v_proj.weight", w[2 * d:]
        yield hf + ".v_proj.bias", b[2 * d:]
        yield from linear(ct2 + "/linear_1", hf + ".out_proj")

    def ffn(ct2, hf):
        yield from norm(ct2 + "/ffn/layer_norm", hf + ".final_layer_norm")
        yield from linear(ct2 + "/ffn/linear_0", hf + ".fc1")
        yield from linear(ct2 + "/ffn/linear_1", hf + ".fc2")

    yield from linear("encoder/conv1", "model.encoder.conv1")
    yield from linear("encoder/conv2", "model.encoder.conv2")
    yield "model.encoder.embed_positions.weight", get("encoder/position_encodings/encodings")
    for i in range(dims.enc_layers):
        c, h = f"encoder/layer_{i}", f"model.encoder.layers.{i}"
        yield from norm(c + "/self_attention/layer_norm", h + ".self_attn_layer_norm")
        yield from self_attention(c + "/self_attention", h + ".self_attn")
        yield from ffn(c, h)
    yield from norm("encoder/layer_norm", "model.encoder.layer_norm")

    yield "model.decoder.embed_tokens.weight", get("decoder/embeddings/weight")
    yield "model.decoder.embed_positions.weight", get("decoder/position_encodings/encodings")
    for i in range(dims.dec_layers):
        c, h = f"decoder/layer_{i}", f"model.decoder.layers.{i}"
        yield from norm(c + "/self_attention/layer_norm", h + ".self_attn_layer_norm")
        yield from self_attention(c + "/self_attention", h + ".self_attn")
        yield from norm(c + "/attention/layer_norm", h + ".encoder_attn_layer_norm")
        yield from linear(c + "/attention/linear_0", h + ".encoder_attn.q_proj")
        w, b = get(c + "/attention/linear_1/weight"), get(c + "/attention/linear_1/bias")
        if w.shape != (2 * d, d):
            raise CT2FormatError(f"{c}/attention/linear_1/weight: expected fused k/v {(2 * d, d)}, got {w.shape}")
        if b.shape != (2 * d,):
            raise CT2FormatError(f"{c}/attention/linear_1/bias: expected fused k/v ({2 * d},), got {b.shape}")
        yield h + ".encoder_attn.k_proj.weight", w[:d]
        yield h + ".encoder_attn.v_proj.weight", w[d:]
        yield h + ".encoder_attn.v_proj.bias", b[d:]
        yield from linear(c + "/attention/linear_2", h + ".encoder_attn.out_proj")
        yield from ffn(c, h)
    yield from norm("decoder/layer_norm", "model.decoder.layer_norm")
    proj = _dense(variables, aliases, "decoder/projection/weight")
    if proj is not None and not np.array_equal(proj, get("decoder/embeddings/weight")):
        raise CT2FormatError("decoder/projection/weight differs from the embeddings: untied output projection is not supported")


def is_ct2_dir(path: str) -> bool:
    return os.path.isfile(os.path.join(path, "model.bin"))


def read_ct2_dir(path: str) -> Tuple[WhisperDims, Iterable[Tuple[str, np.ndarray]], dict]:
    """-> (geometry, HF-named float32 tensors, the directory's config.json as a dict: suppress_ids, lang_ids, …)."""
    spec, _rev, variables, aliases = read_model_bin(os.path.join(path, "model.bin"))
    if spec and "Whisper" not in spec:
        raise CT2FormatError(f"{path}: model.bin holds a {spec}, not a WhisperSpec")
    dims = infer_dims(variables, aliases, os.path.basename(os.path.normpath(path)))
    cfg = {}
    cfg_path = os.path.join(path, "config.json")
    if os.path.exists(cfg_path):
        with open(cfg_path, "r", encoding="utf-8") as f:
            cfg = json.load(f)
    return dims, iter_hf_tensors(variables, aliases, dims), cfg


def hf_to_ct2(tensors: Iterable[Tuple[str, np.ndarray]], dims: WhisperDims, dtype=np.float32) -> Tuple[Dict[str, np.ndarray], Dict[str, str]]:
    """HF-named tensors → CT2 WhisperSpec variables (+ aliases), i.e. what the published converter writes."""
    sd = {k: np.asarray(v, dtype=np.float32) for k, v in tensors}
    d = dims.d_model
    out: Dict[str, np.ndarray] = {}

    def put(name, value):
        out[name] = np.ascontiguousarray(value.astype(dtype))

    def lin(ct2, hf):
        put(ct2 + "/weight", sd[hf + ".weight"])
        put(ct2 + "/bias", sd[hf + ".bias"])

    def nrm(ct2, hf):
        put(ct2 + "/gamma", sd[hf + ".weight"])
        put(ct2 + "/beta", sd[hf + ".bias"])

    def sa(ct2, hf):
        put(ct2 + "/linear_0/weight", np.concatenate([sd[hf + ".q_proj.weight"], sd[hf + ".k_proj.weight"], sd[hf + ".v_proj.weight"]]))
        put(ct2 + "/linear_0/bias", np.concatenate([sd[hf + ".q_proj.bias"], np.zeros(d, np.float32), sd[hf + ".v_proj.bias"]]))
        lin(ct2 + "/linear_1", hf + ".out_proj")

    out["encoder/num_heads"] = np.asarray(dims.n_heads, dtype=np.int16)
    lin("encoder/conv1", "model.encoder.conv1")
    lin("encoder/conv2", "model.encoder.conv2")
    put("encoder/position_encodings/encodings", sd["model.encoder.embed_positions.weight"])
    for i in range(dims.enc_layers):
        c, h = f"encoder/layer_{i}", f"model.encoder.layers.{i}"
        nrm(c + "/self_attention/layer_norm", h + ".self_attn_layer_norm")
        sa(c + "/self_attention", h + ".self_attn")
        nrm(c + "/ffn/layer_norm", h + ".final_layer_norm")
        lin(c + "/ffn/linear_0", h + ".fc1")
        lin(c + "/ffn/linear_1", h + ".fc2")
    nrm("encoder/layer_norm", "model.encoder.layer_norm")
    out["decoder/num_heads"] = np.asarray(dims.n_heads, dtype=np.int16)
    out["decoder/scale_embeddings"] = np.asarray(0, dtype=np.int8)
    put("decoder/embeddings/weight", sd["model.decoder.embed_tokens.weight"])
    put("decoder/position_encodings/encodings", sd["model.decoder.embed_positions.weight"])
    for i in range(dims.dec_layers):
        c, h = f"decoder/layer_{i}", f"model.decoder.layers.{i}"
        nrm(c + "/self_attention/layer_norm", h + ".self_attn_layer_norm")
        sa(c + "/self_attention", h + ".self_attn")
        nrm(c + "/attention/layer_norm", h + ".encoder_attn_layer_norm")
        lin(c + "/attention/linear_0", h + ".encoder_attn.q_proj")
        put(c + "/attention/linear_1/weight", np.concatenate([sd[h + ".encoder_attn.k_proj.weight"], sd[h + ".encoder_attn.v_proj.weight"]]))
        put(c + "/attention/linear_1/bias", np.concatenate([np.zeros(d, np.float32), sd[h + ".encoder_attn.v_proj.bias"]]))
        lin(c + "/attention/linear_2", h + ".encoder_attn.out_proj")
        nrm(c + "/ffn/layer_norm", h + ".final_layer_norm")
        lin(c + "/ffn/linear_0", h + ".fc1")
        lin(c + "/ffn/linear_1", h + ".fc2")
    nrm("decoder/layer_norm", "model.decoder.layer_norm")
    return out, {"decoder/projection/weight": "decoder/embeddings/weight"}
