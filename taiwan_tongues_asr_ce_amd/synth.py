"""Deterministic synthetic weights and audio clips.

No Whisper checkpoint exists on either box, so every parity and throughput run
uses weights produced here: one numpy Philox counter stream per tensor, keyed by (seed, crc32(name)), so
any tensor can be regenerated alone and the values do not depend on generation order or numpy version.
Tensor names are the HF state-dict names the reference's training path loads
(train_asr.py:518-545 -> WhisperForConditionalGeneration).
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Iterator, List, Tuple

import numpy as np

from .config import N_SAMPLES, WhisperDims


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, zlib.crc32(name.encode())]))


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> np.ndarray:
    """Encoder positional table, closed form of [HF] modeling_whisper.py:55-64 (float32 arithmetic)."""
    inc = np.float32(math.log(max_timescale) / (channels // 2 - 1))
    inv = np.exp(-inc * np.arange(channels // 2, dtype=np.float32)).astype(np.float32)
    t = np.arange(length, dtype=np.float32)[:, None] * inv[None, :]
    return np.concatenate([np.sin(t), np.cos(t)], axis=1).astype(np.float32)


def tensor_specs(dims: WhisperDims) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, kind) for every tensor of the model. kind selects the distribution."""
    d, f, m = dims.d_model, dims.ffn_dim, dims.n_mels
    out: List[Tuple[str, Tuple[int, ...], str]] = []

    def lin(prefix: str, n_out: int, n_in: int, bias: bool = True):
        out.append((prefix + ".weight", (n_out, n_in), "linear"))
        if bias:
            out.append((prefix + ".bias", (n_out,), "bias"))

    def ln(prefix: str):
        out.append((prefix + ".weight", (d,), "gamma"))
        out.append((prefix + ".bias", (d,), "beta"))

    def attn(prefix: str):
        lin(prefix + ".k_proj", d, d, bias=False)  # [HF] modeling_whisper.py:279 - no key bias
        lin(prefix + ".v_proj", d, d)
        lin(prefix + ".q_proj", d, d)
        lin(prefix + ".out_proj", d, d)

    out.append(("model.encoder.conv1.weight", (d, m, 3), "conv"))
    out.append(("model.encoder.conv1.bias", (d,), "bias"))
    out.append(("model.encoder.conv2.weight", (d, d, 3), "conv"))
    out.append(("model.encoder.conv2.bias", (d,), "bias"))
    out.append(("model.encoder.embed_positions.weight", (dims.n_audio_ctx, d), "sinusoid"))
    for i in range(dims.enc_layers):
        p = f"model.encoder.layers.{i}"
        attn(p + ".self_attn")
        ln(p + ".self_attn_layer_norm")
        lin(p + ".fc1", f, d)
        lin(p + ".fc2", d, f)
        ln(p + ".final_layer_norm")
    ln("model.encoder.layer_norm")
    out.append(("model.decoder.embed_tokens.weight", (dims.vocab, d), "embed"))
    out.append(("model.decoder.embed_positions.weight", (dims.n_text_ctx, d), "embed"))
    for i in range(dims.dec_layers):
        p = f"model.decoder.layers.{i}"
        attn(p + ".self_attn")
        ln(p + ".self_attn_layer_norm")
        attn(p + ".encoder_attn")
        ln(p + ".encoder_attn_layer_norm")
        lin(p + ".fc1", f, d)
        lin(p + ".fc2", d, f)
        ln(p + ".final_layer_norm")
    ln("model.decoder.layer_norm")
    return out


def make_tensor(name: str, shape: Tuple[int, ...], kind: str, seed: int = 0) -> np.ndarray:
    if kind == "sinusoid":
        return sinusoids(shape[0], shape[1])
    g = _rng(seed, name)
    x = g.standard_normal(shape, dtype=np.float32)
    if kind == "linear":
        x *= np.float32(1.0 / math.sqrt(shape[1]))
    elif kind == "conv":
        x *= np.float32(1.0 / math.sqrt(shape[1] * shape[2]))
    elif kind == "bias":
        x *= np.float32(0.02)
    elif kind == "gamma":
        x = np.float32(1.0) + np.float32(0.1) * x
    elif kind == "beta":
        x *= np.float32(0.1)
    elif kind == "embed":
        x *= np.float32(0.05)
    else:
        raise ValueError(kind)
    return x


def iter_weights(dims: WhisperDims, seed: int = 0) -> Iterator[Tuple[str, np.ndarray]]:
    for name, shape, kind in tensor_specs(dims):
        yield name, make_tensor(name, shape, kind, seed)


def state_dict(dims: WhisperDims, seed: int = 0) -> Dict[str, np.ndarray]:
    return dict(iter_weights(dims, seed))


def noise_clip(i: int, n_samples: int = N_SAMPLES) -> np.ndarray:
    """Clip i of the benchmark set: 0.1 * N(0,1), Philox(key=1234+i)."""
    g = np.random.Generator(np.random.Philox(key=1234 + i))
    return (np.float32(0.1) * g.standard_normal(n_samples, dtype=np.float32)).astype(np.float32)


def tonal_clip(i: int, n_samples: int = N_SAMPLES) -> np.ndarray:
    """Five sines 100-4000 Hz, amplitude 0.05 each; exercises the per-clip max-8 clamp."""
    t = np.arange(n_samples, dtype=np.float64) / 16000.0
    freqs = np.array([100.0, 440.0, 1000.0, 2500.0, 4000.0]) * (1.0 + 0.01 * i)
    x = sum(0.05 * np.sin(2 * np.pi * f * t + 0.3 * k) for k, f in enumerate(freqs))
    return x.astype(np.float32)


def burst_clip(i: int, n_samples: int = N_SAMPLES, burst_seconds: float = 3.0) -> np.ndarray:
    """Noise burst then digital silence: the streaming path pads ~3 s utterances to 30 s
    (buffering_strategies.py:118-126), and exact zeros hit the 1e-10 clamp."""
    x = np.zeros(n_samples, dtype=np.float32)
    n = min(n_samples, int(burst_seconds * 16000))
    x[:n] = noise_clip(1000 + i, n)
    return x
