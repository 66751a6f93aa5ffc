"""TEST infrastructure: an INDEPENDENT writer of CTranslate2 `model.bin` files for Whisper, written from the published
serialisation (CTranslate2 `ModelSpec.save` / `_serialize`, binary version 6) and the published converter's variable naming
(`ctranslate2/converters/transformers.py` WhisperLoader, `specs/whisper_spec.py`), sharing NO code with
taiwan_tongues_asr_ce_amd/ct2.py.  It writes what `ct2-transformers-converter --quantization int8` produces: linear weights
as int8 with a per-output-row float32 `<name>_scale`, everything else float16, the output projection as an ALIAS of the
embeddings, scalars (num_heads ...) as 0-d int16/int8 variables, variables sorted by name."""
import io
import json
import os

import numpy as np


def _str(buf, text):
    raw = text.encode("utf-8") + b"\x00"
    buf.write(np.uint16(len(raw)).tobytes())
    buf.write(raw)


def _var(buf, name, arr):
    arr = np.ascontiguousarray(arr)
    ids = {"float32": 0, "int8": 1, "int16": 2, "int32": 3, "float16": 4}
    _str(buf, name)
    buf.write(np.uint8(arr.ndim).tobytes())
    for dim in arr.shape:
        buf.write(np.uint32(dim).tobytes())
    buf.write(np.uint8(ids[arr.dtype.name]).tobytes())
    buf.write(np.uint32(arr.nbytes).tobytes())
    buf.write(arr.tobytes())


def write_whisper_ct2_int8(directory, hf, n_heads, enc_layers, dec_layers, config=None):
    """hf: {HF state-dict name: float32 array}.  Returns {HF name: the float32 value a correct reader must reconstruct}."""
    v, expect = {}, {}

    def quant(name, w):                                   # CTranslate2 int8: scale = 127 / max|row|, q = round(w * scale)
        amax = np.abs(w).max(axis=1)
        scale = (127.0 / np.where(amax == 0, 127.0, amax)).astype(np.float32)
        q = np.round(w * scale[:, None]).astype(np.int8)
        v[name] = q
        v[name + "_scale"] = scale
        return q.astype(np.float32) / scale[:, None]

    def half(name, x):
        v[name] = x.astype(np.float16)
        return v[name].astype(np.float32)

    def dense(ct, hf_name):
        expect[hf_name + ".weight"] = quant(ct + "/weight", hf[hf_name + ".weight"])
        expect[hf_name + ".bias"] = half(ct + "/bias", hf[hf_name + ".bias"])

    def ln(ct, hf_name):
        expect[hf_name + ".weight"] = half(ct + "/gamma", hf[hf_name + ".weight"])
        expect[hf_name + ".bias"] = half(ct + "/beta", hf[hf_name + ".bias"])

    def fused(ct, parts):                                  # rows of several HF projections in one CT2 linear; missing bias = zeros
        w = np.concatenate([hf[p + ".weight"] for p in parts])
        b = np.concatenate([hf.get(p + ".bias", np.zeros(hf[p + ".weight"].shape[0], np.float32)) for p in parts])
        wq, bh = quant(ct + "/weight", w), half(ct + "/bias", b)
        off = 0
        for p in parts:
            n = hf[p + ".weight"].shape[0]
            expect[p + ".weight"] = wq[off:off + n]
            if p + ".bias" in hf:
                expect[p + ".bias"] = bh[off:off + n]
            off += n

    for side, n_layers in (("encoder", enc_layers), ("decoder", dec_layers)):
        v[f"{side}/num_heads"] = np.asarray(n_heads, dtype=np.int16)
        for i in range(n_layers):
            c, h = f"{side}/layer_{i}", f"model.{side}.layers.{i}"
            ln(c + "/self_attention/layer_norm", h + ".self_attn_layer_norm")
            fused(c + "/self_attention/linear_0", [h + ".self_attn.q_proj", h + ".self_attn.k_proj", h + ".self_attn.v_proj"])
            dense(c + "/self_attention/linear_1", h + ".self_attn.out_proj")
            if side == "decoder":
                ln(c + "/attention/layer_norm", h + ".encoder_attn_layer_norm")
                dense(c + "/attention/linear_0", h + ".encoder_attn.q_proj")
                fused(c + "/attention/linear_1", [h + ".encoder_attn.k_proj", h + ".encoder_attn.v_proj"])
                dense(c + "/attention/linear_2", h + ".encoder_attn.out_proj")
            ln(c + "/ffn/layer_norm", h + ".final_layer_norm")
            dense(c + "/ffn/linear_0", h + ".fc1")
            dense(c + "/ffn/linear_1", h + ".fc2")
        ln(f"{side}/layer_norm", f"model.{side}.layer_norm")
    for k in ("conv1", "conv2"):                           # convolutions stay float16 (only Linear is quantised)
        expect[f"model.encoder.{k}.weight"] = half(f"encoder/{k}/weight", hf[f"model.encoder.{k}.weight"])
        expect[f"model.encoder.{k}.bias"] = half(f"encoder/{k}/bias", hf[f"model.encoder.{k}.bias"])
    expect["model.encoder.embed_positions.weight"] = half("encoder/position_encodings/encodings", hf["model.encoder.embed_positions.weight"])
    expect["model.decoder.embed_positions.weight"] = half("decoder/position_encodings/encodings", hf["model.decoder.embed_positions.weight"])
    expect["model.decoder.embed_tokens.weight"] = quant("decoder/embeddings/weight", hf["model.decoder.embed_tokens.weight"])
    v["decoder/scale_embeddings"] = np.asarray(0, dtype=np.int8)
    buf = io.BytesIO()
    buf.write(np.uint32(6).tobytes())
    _str(buf, "WhisperSpec")
    buf.write(np.uint32(3).tobytes())
    buf.write(np.uint32(len(v)).tobytes())
    for name in sorted(v):
        _var(buf, name, v[name])
    aliases = {"decoder/projection/weight": "decoder/embeddings/weight", "decoder/projection/weight_scale": "decoder/embeddings/weight_scale"}
    buf.write(np.uint32(len(aliases)).tobytes())
    for a in sorted(aliases):
        _str(buf, a)
        _str(buf, aliases[a])
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, "model.bin"), "wb") as f:
        f.write(buf.getvalue())
    with open(os.path.join(directory, "config.json"), "w", encoding="utf-8") as f:
        json.dump(config or {}, f)
    return expect
