"""CPU restatement of the Whisper inference hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
path (taiwan_tongues_asr_ce_amd) never does and fails loudly when its HIP library is missing.

What it restates.  The reference's hot path lives in un-vendored third-party code
(faster-whisper >= 0.9.0 / CTranslate2, requirements.txt:10; call sites asr_core.py:141,159-167,
api/file_asr.py:188,457-465, api/stt_streaming/src/asr/faster_whisper_asr.py:107,170).  Neither package is
installed or installable here, so the arithmetic is restated from the reference's *other* Whisper
implementation, HF Transformers (train_asr.py:33-43,518-545), citing
  [HF-FE]  transformers/models/whisper/feature_extraction_whisper.py
  [HF-M]   transformers/models/whisper/modeling_whisper.py
  [HF-LP]  transformers/generation/logits_process.py
  [HF-G]   transformers/models/whisper/generation_whisper.py
(transformers 5.15.0).  Pinning: tests/golden/*.npz hold outputs of that HF code run in the build
container on seeded synthetic weights (oracle/make_golden.py is the generating script);
tests/test_oracle_golden.py checks every function below against them.  The reference has no golden
vectors of its own for this path (api/tests/test_file_asr.py:40-60 mocks the model), and parity with the
CTranslate2 engine itself is unpinned (it cannot be run here).

Everything is float32 torch-CPU tensor algebra written from the formulas; no HF model class is used.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

NEG_INF = float("-inf")


# --------------------------------------------------------------------------------------------------
# a5: log-mel front-end
# --------------------------------------------------------------------------------------------------
def _hz_to_mel_slaney(f: np.ndarray) -> np.ndarray:
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3.0
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    with np.errstate(divide="ignore"):
        logpart = min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep
    return np.where(f >= min_log_hz, logpart, mels)


def _mel_to_hz_slaney(m: np.ndarray) -> np.ndarray:
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3.0
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filter_bank(n_mels: int, n_fft: int = 400, sr: int = 16000, fmin: float = 0.0,
                    fmax: float = 8000.0) -> np.ndarray:
    """[n_fft//2+1, n_mels] triangular slaney-scale, slaney-normalised filters.
    Same construction as [HF-FE]:95-103 (mel_filter_bank(norm="slaney", mel_scale="slaney"))."""
    n_freqs = n_fft // 2 + 1
    fft_freqs = np.linspace(0.0, sr / 2.0, n_freqs)
    mel_pts = np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2)
    hz_pts = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(hz_pts)
    slopes = hz_pts[None, :] - fft_freqs[:, None]  # [n_freqs, n_mels+2]
    down = -slopes[:, :-2] / fdiff[:-1]
    up = slopes[:, 2:] / fdiff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (hz_pts[2:n_mels + 2] - hz_pts[:n_mels])
    fb *= enorm[None, :]
    return fb.astype(np.float32)


def hann_window(n: int = 400) -> np.ndarray:
    """Periodic Hann (torch.hann_window default), [HF-FE]:143."""
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)).astype(np.float64)


def log_mel(pcm: np.ndarray, n_mels: int, n_samples: int = 480000, n_fft: int = 400, hop: int = 160
            ) -> np.ndarray:
    """f32[n] -> f32[n_mels, n_samples//hop].  Follows [HF-FE]:135-168: pad/trim to n_samples, centred
    STFT with reflect padding, drop the last frame (:154), power, mel matmul (:157),
    log10(clamp 1e-10) (:159), max(x, clip_max - 8) (:160-162), (x + 4) / 4 (:165).
    The DFT is done in float64 (ground truth for both the f32 HF path and the f32 HIP kernel)."""
    x = np.zeros(n_samples, dtype=np.float64)
    n = min(len(pcm), n_samples)
    x[:n] = np.asarray(pcm[:n], dtype=np.float64)
    pad = n_fft // 2
    xp = np.pad(x, (pad, pad), mode="reflect")
    n_frames = n_samples // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = xp[idx] * hann_window(n_fft)[None, :]
    spec = np.fft.rfft(frames, axis=1)
    power = spec.real ** 2 + spec.imag ** 2  # [frames, 201]
    mel = power @ mel_filter_bank(n_mels, n_fft).astype(np.float64)  # [frames, n_mels]
    logm = np.log10(np.maximum(mel, 1e-10))
    logm = np.maximum(logm, logm.max() - 8.0)
    logm = (logm + 4.0) / 4.0
    return np.ascontiguousarray(logm.T).astype(np.float32)


def log_mel_file(pcm: np.ndarray, n_mels: int) -> np.ndarray:
    """Features of a WHOLE recording, f32[n_mels, len(pcm) // 160]: log_mel without the pad / trim to one window, i.e.
    what WhisperFeatureExtractor(..., truncation=False, padding="longest") returns and what faster-whisper computes once per
    file before its 30-s window loop (reflection only at the two ends of the file, ONE dynamic-range floor for the file)."""
    return log_mel(pcm, n_mels, n_samples=len(pcm))


def file_window(features: np.ndarray, seek: int, n_frames: int = 3000) -> np.ndarray:
    """The encoder input of the window that starts at frame `seek`: features[:, seek:seek + n_frames], zero-padded in
    FEATURE space when the recording ends inside the window ([HF] generation_whisper.py long-form loop;
    faster-whisper pad_or_trim)."""
    seg = features[:, seek:seek + n_frames]
    out = np.zeros((features.shape[0], n_frames), dtype=np.float32)
    out[:, :seg.shape[1]] = seg
    return out


# --------------------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------------------
@dataclass
class Dims:
    n_mels: int
    n_audio_ctx: int
    d_model: int
    n_heads: int
    ffn_dim: int
    enc_layers: int
    dec_layers: int
    vocab: int
    n_text_ctx: int = 448


def to_torch(sd: Dict[str, np.ndarray], round_bf16: bool = False, round_f16: bool = False) -> Dict[str, torch.Tensor]:
    """numpy state dict -> torch f32.  round_bf16 / round_f16 round every matrix a 16-bit engine stores in that type
    (all >=2-D tensors) so the oracle sees the same weight values as the 16-bit kernels.  The engine keeps the encoder's
    sinusoid table in f32 in every mode; the f16 rounding leaves it alone (the bf16 rounding, pinned by round-1/2 tolerances,
    still rounds it: the difference is far inside the bf16 tolerances)."""
    out = {}
    for k, v in sd.items():
        t = torch.from_numpy(np.ascontiguousarray(v)).float()
        if round_bf16 and t.dim() >= 2:
            t = t.bfloat16().float()
        elif round_f16 and t.dim() >= 2 and k != "model.encoder.embed_positions.weight":
            t = t.half().float()
        out[k] = t
    return out


def _ln(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def _gelu(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


# Saturation of STORED activations (test-side switch, default off = the reference's f32 arithmetic).  The fp16 engine stores every
# projection output as fp16 and clamps it to +-65504 on the way (common.hpp N16<f16_t>::sat; HF clamps its fp16 hidden states for the
# same reason, [HF-M]:403-407); `with activation_clamp(65504.0):` makes this restatement clamp at the same places - the output of
# every linear layer (q after its 1/8 scaling: the engine folds the scaling into the weights) and of the conv stem's first layer.
# The decoder's out-proj / fc2 outputs are the exception when `decoder_residual=False`: the engine's decode STEP never stores them
# in 16 bits (their K-split f32 partial tiles are added straight into the f32 residual stream), so nothing clamps them there.
_ACT_CLAMP: Optional[float] = None
_ACT_CLAMP_DEC_RESIDUAL: bool = True


class activation_clamp:
    def __init__(self, limit: Optional[float], decoder_residual: bool = True):
        self.limit, self.dec_res = limit, decoder_residual

    def __enter__(self):
        global _ACT_CLAMP, _ACT_CLAMP_DEC_RESIDUAL
        self.prev = (_ACT_CLAMP, _ACT_CLAMP_DEC_RESIDUAL)
        _ACT_CLAMP, _ACT_CLAMP_DEC_RESIDUAL = self.limit, self.dec_res
        return self

    def __exit__(self, *exc):
        global _ACT_CLAMP, _ACT_CLAMP_DEC_RESIDUAL
        _ACT_CLAMP, _ACT_CLAMP_DEC_RESIDUAL = self.prev
        return False


def _sat(y: torch.Tensor) -> torch.Tensor:
    return y if _ACT_CLAMP is None else y.clamp(-_ACT_CLAMP, _ACT_CLAMP)


def _lin(x: torch.Tensor, W: Dict[str, torch.Tensor], p: str, scale: Optional[float] = None, dec_residual: bool = False
         ) -> torch.Tensor:
    """dec_residual: this is a decoder out-proj / fc2 (its output is added to the residual stream)."""
    y = x @ W[p + ".weight"].t()
    b = W.get(p + ".bias")
    if b is not None:
        y = y + b
    if scale is not None:
        y = y * scale
    return y if (dec_residual and not _ACT_CLAMP_DEC_RESIDUAL) else _sat(y)


def _split_heads(x: torch.Tensor, H: int) -> torch.Tensor:
    B, T, d = x.shape
    return x.view(B, T, H, d // H).permute(0, 2, 1, 3)


def _attend(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, mask: Optional[torch.Tensor] = None
            ) -> torch.Tensor:
    """q already scaled ([HF-M]:309: scaling is applied to q_proj output, attention called with 1.0)."""
    s = q @ k.transpose(-1, -2)
    if mask is not None:
        s = s + mask
    p = torch.softmax(s, dim=-1)
    o = p @ v  # [B,H,Tq,hd]
    B, H, Tq, hd = o.shape
    return o.permute(0, 2, 1, 3).reshape(B, Tq, H * hd)


# --------------------------------------------------------------------------------------------------
# a6 + a7: encoder
# --------------------------------------------------------------------------------------------------
def encoder_stem(mel: torch.Tensor, W: Dict[str, torch.Tensor]) -> torch.Tensor:
    """[B,M,2T] -> [B,T,d]: gelu(conv1 k3 s1 p1), gelu(conv2 k3 s2 p1), transpose, + positions
    ([HF-M]:618-624)."""
    x = _sat(_gelu(F.conv1d(mel, W["model.encoder.conv1.weight"], W["model.encoder.conv1.bias"], stride=1, padding=1)))
    x = _gelu(F.conv1d(x, W["model.encoder.conv2.weight"], W["model.encoder.conv2.bias"], stride=2, padding=1))
    x = x.permute(0, 2, 1)
    return x + W["model.encoder.embed_positions.weight"][None, : x.shape[1]]


def encoder_layer(x: torch.Tensor, W: Dict[str, torch.Tensor], p: str, H: int) -> torch.Tensor:
    """Pre-LN block, [HF-M]:380-413."""
    hd = x.shape[-1] // H
    h = _ln(x, W[p + ".self_attn_layer_norm.weight"], W[p + ".self_attn_layer_norm.bias"])
    q = _split_heads(_lin(h, W, p + ".self_attn.q_proj", scale=hd ** -0.5), H)
    k = _split_heads(_lin(h, W, p + ".self_attn.k_proj"), H)
    v = _split_heads(_lin(h, W, p + ".self_attn.v_proj"), H)
    x = x + _lin(_attend(q, k, v), W, p + ".self_attn.out_proj")
    h = _ln(x, W[p + ".final_layer_norm.weight"], W[p + ".final_layer_norm.bias"])
    return x + _lin(_gelu(_lin(h, W, p + ".fc1")), W, p + ".fc2")


def encoder_forward(mel: torch.Tensor, W: Dict[str, torch.Tensor], dims: Dims,
                    return_hidden: bool = False):
    x = encoder_stem(mel, W)
    hidden = [x]
    for i in range(dims.enc_layers):
        x = encoder_layer(x, W, f"model.encoder.layers.{i}", dims.n_heads)
        hidden.append(x)
    x = _ln(x, W["model.encoder.layer_norm.weight"], W["model.encoder.layer_norm.bias"])
    return (x, hidden) if return_hidden else x


# --------------------------------------------------------------------------------------------------
# a8: cross-attention K/V, a9: decoder step
# --------------------------------------------------------------------------------------------------
def cross_kv(enc: torch.Tensor, W: Dict[str, torch.Tensor], dims: Dims
             ) -> List[Tuple[torch.Tensor, torch.Tensor]]:
    """Per decoder layer (K, V) as [B,H,T,hd]; K has no bias ([HF-M]:279,331-335)."""
    out = []
    for i in range(dims.dec_layers):
        p = f"model.decoder.layers.{i}.encoder_attn"
        out.append((_split_heads(_lin(enc, W, p + ".k_proj"), dims.n_heads),
                    _split_heads(_lin(enc, W, p + ".v_proj"), dims.n_heads)))
    return out


@dataclass
class SelfCache:
    k: List[Optional[torch.Tensor]]
    v: List[Optional[torch.Tensor]]

    @staticmethod
    def empty(n_layers: int) -> "SelfCache":
        return SelfCache([None] * n_layers, [None] * n_layers)

    @property
    def length(self) -> int:
        return 0 if self.k[0] is None else self.k[0].shape[2]


def decoder_forward(tokens: torch.Tensor, cache: SelfCache, xkv, W: Dict[str, torch.Tensor], dims: Dims,
                    cross_probs: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """tokens [B,n] appended at positions cache.length.. ; returns logits [B,n,V] ([HF-M]:737-790, 1080).
    n > 1 applies the causal mask among the new tokens.  If `cross_probs` is a list, the cross-attention
    probabilities [B,H,n,T] of every layer are appended to it (HF `output_attentions`, [HF-M]:347-356)."""
    B, n = tokens.shape
    H = dims.n_heads
    hd = dims.d_model // H
    p0 = cache.length
    x = W["model.decoder.embed_tokens.weight"][tokens] + W["model.decoder.embed_positions.weight"][p0:p0 + n][None]
    mask = None
    if n > 1:
        mask = torch.full((n, p0 + n), NEG_INF)
        mask = torch.triu(mask, diagonal=p0 + 1)
    for i in range(dims.dec_layers):
        p = f"model.decoder.layers.{i}"
        h = _ln(x, W[p + ".self_attn_layer_norm.weight"], W[p + ".self_attn_layer_norm.bias"])
        q = _split_heads(_lin(h, W, p + ".self_attn.q_proj", scale=hd ** -0.5), H)
        k = _split_heads(_lin(h, W, p + ".self_attn.k_proj"), H)
        v = _split_heads(_lin(h, W, p + ".self_attn.v_proj"), H)
        if cache.k[i] is not None:
            k = torch.cat([cache.k[i], k], dim=2)
            v = torch.cat([cache.v[i], v], dim=2)
        cache.k[i], cache.v[i] = k, v
        x = x + _lin(_attend(q, k, v, mask), W, p + ".self_attn.out_proj", dec_residual=True)
        h = _ln(x, W[p + ".encoder_attn_layer_norm.weight"], W[p + ".encoder_attn_layer_norm.bias"])
        q = _split_heads(_lin(h, W, p + ".encoder_attn.q_proj", scale=hd ** -0.5), H)
        if cross_probs is not None:
            cross_probs.append(torch.softmax(q @ xkv[i][0].transpose(-1, -2), dim=-1))
        x = x + _lin(_attend(q, xkv[i][0], xkv[i][1]), W, p + ".encoder_attn.out_proj", dec_residual=True)
        h = _ln(x, W[p + ".final_layer_norm.weight"], W[p + ".final_layer_norm.bias"])
        x = x + _lin(_gelu(_lin(h, W, p + ".fc1")), W, p + ".fc2", dec_residual=True)
    x = _ln(x, W["model.decoder.layer_norm.weight"], W["model.decoder.layer_norm.bias"])
    return x @ W["model.decoder.embed_tokens.weight"].t()  # proj_out tied, [HF-M]:965,970


# --------------------------------------------------------------------------------------------------
# a10: logits processors + greedy selection
# --------------------------------------------------------------------------------------------------
@dataclass
class Rules:
    """Static description of the processor stack ([HF-G]:1774-1812 order: begin-suppress, suppress,
    timestamp rules)."""
    eot: int
    no_timestamps: int
    timestamp_begin: int
    suppress: Sequence[int] = ()
    begin_suppress: Sequence[int] = ()
    timestamps: bool = True            # WhisperTimeStampLogitsProcessor active
    max_initial_timestamp_index: Optional[int] = 50
    suppress_eot: bool = False         # benchmark mode: never stop


def apply_rules(logits: torch.Tensor, sampled: Sequence[int], rules: Rules, ts_prob_rule: Optional[bool] = None,
                ts_prob_margin: Optional[List[float]] = None) -> torch.Tensor:
    """One row f32[V] + the tokens sampled so far for that row (after the prompt) -> processed row.
    [HF-LP]:1816 (begin suppress), :1869 (suppress), :2000-2047 (timestamp rules).
    Test-side hooks for the one DISCONTINUOUS rule ("timestamp probability mass > best text probability => text is masked",
    [HF-LP]:2040-2047): `ts_prob_margin`, when a list, receives logsumexp(timestamp log-probs) - max(text log-prob) (the rule
    fires when > 0); `ts_prob_rule` = True / False forces the rule on / off instead of evaluating it (graders use it to grade a
    16-bit engine's token when that margin is inside the logit tolerance).  Defaults = the reference behaviour."""
    s = logits.clone().float()
    n = len(sampled)
    if n == 0 and len(rules.begin_suppress):
        s[list(rules.begin_suppress)] = NEG_INF
    if len(rules.suppress):
        s[list(rules.suppress)] = NEG_INF
    if rules.suppress_eot:
        s[rules.eot] = NEG_INF
    if rules.timestamps:
        tb = rules.timestamp_begin
        s[rules.no_timestamps] = NEG_INF
        last_ts = n >= 1 and sampled[-1] >= tb
        pen_ts = n < 2 or sampled[-2] >= tb
        if last_ts:
            if pen_ts:
                s[tb:] = NEG_INF
            else:
                s[: rules.eot] = NEG_INF
        ts = [t for t in sampled if t >= tb]
        if ts:
            last = ts[-1] if (last_ts and not pen_ts) else ts[-1] + 1
            s[tb:last] = NEG_INF
        if n == 0:
            s[:tb] = NEG_INF
            if rules.max_initial_timestamp_index is not None:
                s[tb + rules.max_initial_timestamp_index + 1:] = NEG_INF
        lp = torch.log_softmax(s, dim=-1)
        m = torch.logsumexp(lp[tb:], dim=-1) - lp[:tb].max()
        if ts_prob_margin is not None:
            ts_prob_margin.append(float(m))
        if (m > 0) if ts_prob_rule is None else ts_prob_rule:
            s[:tb] = NEG_INF
    return s


@dataclass
class GreedyResult:
    tokens: List[List[int]]              # sampled tokens per clip (EOT included when emitted)
    sum_logprob: List[float]
    no_speech_prob: List[float]
    step_logits: List[torch.Tensor] = field(default_factory=list)  # raw logits at each sampling step [B,V]


def greedy_decode(enc: torch.Tensor, prompt: Sequence[int], W: Dict[str, torch.Tensor], dims: Dims,
                  rules: Rules, max_new_tokens: int, no_speech_token: Optional[int] = None,
                  sot_index: int = 0, keep_logits: bool = False, token_by_token_prompt: bool = True
                  ) -> GreedyResult:
    """Greedy search with the processor stack; every clip shares `prompt`.  The prompt is fed one token at
    a time (exactly what the HIP engine does), which is arithmetically the same as a causal prefill.
    no-speech probability = softmax of the raw logits at the <|startoftranscript|> position
    ([HF-LP]:2050-2113)."""
    B = enc.shape[0]
    xkv = cross_kv(enc, W, dims)
    cache = SelfCache.empty(dims.dec_layers)
    no_speech = [0.0] * B
    logits = None
    if token_by_token_prompt:
        for j, t in enumerate(prompt):
            logits = decoder_forward(torch.full((B, 1), t, dtype=torch.long), cache, xkv, W, dims)[:, -1]
            if no_speech_token is not None and j == sot_index:
                no_speech = torch.softmax(logits.float(), dim=-1)[:, no_speech_token].tolist()
    else:
        full = decoder_forward(torch.tensor([list(prompt)] * B, dtype=torch.long), cache, xkv, W, dims)
        logits = full[:, -1]
        if no_speech_token is not None:
            no_speech = torch.softmax(full[:, sot_index].float(), dim=-1)[:, no_speech_token].tolist()
    sampled: List[List[int]] = [[] for _ in range(B)]
    done = [False] * B
    sum_lp = [0.0] * B
    res = GreedyResult(sampled, sum_lp, no_speech)
    for _ in range(max_new_tokens):
        if keep_logits:
            res.step_logits.append(logits.clone())
        nxt = []
        for b in range(B):
            if done[b]:
                nxt.append(rules.eot)
                continue
            s = apply_rules(logits[b], sampled[b], rules)
            t = int(torch.argmax(s))
            sum_lp[b] += float(torch.log_softmax(s, dim=-1)[t])
            sampled[b].append(t)
            if t == rules.eot:
                done[b] = True
            nxt.append(t)
        if all(done) or cache.length >= dims.n_text_ctx:
            break
        logits = decoder_forward(torch.tensor(nxt, dtype=torch.long)[:, None], cache, xkv, W, dims)[:, -1]
    return res


def transcribe_tokens(pcm_batch: Sequence[np.ndarray], W: Dict[str, torch.Tensor], dims: Dims,
                      prompt: Sequence[int], rules: Rules, max_new_tokens: int,
                      no_speech_token: Optional[int] = None, sot_index: int = 0) -> GreedyResult:
    """PCM -> token ids: the whole hot path (a5..a10) for one window per clip."""
    n_samples = dims.n_audio_ctx * 2 * 160
    mel = torch.from_numpy(np.stack([log_mel(p, dims.n_mels, n_samples) for p in pcm_batch]))
    enc = encoder_forward(mel, W, dims)
    return greedy_decode(enc, prompt, W, dims, rules, max_new_tokens, no_speech_token, sot_index)


# --------------------------------------------------------------------------------------------------
# a10 (beam): beam search as the reference call sites request it (beam_size=5 at asr_core.py:164,
# file_asr.py:462, faster_whisper_asr.py:144; patience 1, length_penalty 1 are faster-whisper's defaults).
# CTranslate2's implementation is un-vendored and unpinned; this restates the published Whisper algorithm
# (openai-whisper decoding.py BeamSearchDecoder + MaximumLikelihoodRanker) that CT2 re-implements:
# per audio, every live beam proposes its top (beam+1) tokens of log_softmax(processed logits); candidates
# are keyed by their full token sequence (duplicates collapse), taken in descending cumulative log-prob;
# those ending in EOT go to the finished pool (capped at round(beam * patience)), the first `beam` others
# become the new beams; decoding stops when every audio has a full finished pool or the length limit hits;
# the winner maximises sum_logprob / length (length_penalty = 1).
# --------------------------------------------------------------------------------------------------
@dataclass
class BeamResult:
    tokens: List[List[int]]          # best hypothesis per audio (EOT stripped)
    sum_logprob: List[float]
    no_speech_prob: List[float]


def beam_decode(enc: torch.Tensor, prompt: Sequence[int], W: Dict[str, torch.Tensor], dims: Dims, rules: Rules,
                beam: int, max_new_tokens: int, patience: float = 1.0, no_speech_token: Optional[int] = None,
                sot_index: int = 0) -> BeamResult:
    A = enc.shape[0]
    R = A * beam
    xkv_a = cross_kv(enc, W, dims)
    xkv = [(k.repeat_interleave(beam, dim=0), v.repeat_interleave(beam, dim=0)) for k, v in xkv_a]
    cache = SelfCache.empty(dims.dec_layers)
    no_speech = [0.0] * A
    logits = None
    for j, t in enumerate(prompt):
        logits = decoder_forward(torch.full((R, 1), t, dtype=torch.long), cache, xkv, W, dims)[:, -1]
        if no_speech_token is not None and j == sot_index:
            no_speech = torch.softmax(logits.float(), dim=-1)[::beam, no_speech_token].tolist()
    seqs: List[List[int]] = [[] for _ in range(R)]
    sums = [0.0] * R
    max_cand = round(beam * patience)
    finished: List[Dict[Tuple[int, ...], float]] = [dict() for _ in range(A)]
    for _ in range(max_new_tokens):
        lps = []
        for r in range(R):
            s = apply_rules(logits[r], seqs[r], rules)
            lps.append(torch.log_softmax(s, dim=-1))
        next_seqs, next_sums, src = [], [], []
        for a in range(A):
            scores: Dict[Tuple[int, ...], float] = {}
            sources: Dict[Tuple[int, ...], int] = {}
            for j in range(beam):
                r = a * beam + j
                top = torch.topk(lps[r], beam + 1)
                for lp, tok in zip(top.values.tolist(), top.indices.tolist()):
                    key = tuple(seqs[r] + [tok])
                    val = sums[r] + lp
                    if key not in scores or val > scores[key]:  # identical sequences collapse (dict semantics)
                        scores[key] = val
                        sources[key] = r
            saved = 0
            fin_new = {}
            for key in sorted(scores, key=lambda k: (-scores[k], k)):
                if key[-1] == rules.eot:
                    fin_new[key] = scores[key]
                else:
                    next_seqs.append(list(key)); next_sums.append(scores[key]); src.append(sources[key])
                    saved += 1
                    if saved == beam:
                        break
            for key in sorted(fin_new, key=lambda k: (-fin_new[k], k)):
                if len(finished[a]) >= max_cand:
                    break
                finished[a][key] = fin_new[key]
            while saved < beam:  # fewer than `beam` live candidates (tiny vocabularies): pad with the last one
                next_seqs.append(list(next_seqs[-1])); next_sums.append(-1e30); src.append(src[-1]); saved += 1
        seqs, sums = next_seqs, next_sums
        idx = torch.tensor(src, dtype=torch.long)
        for l in range(dims.dec_layers):
            cache.k[l] = cache.k[l].index_select(0, idx)
            cache.v[l] = cache.v[l].index_select(0, idx)
        if all(len(f) >= max_cand for f in finished) or cache.length >= dims.n_text_ctx:
            break
        nxt = torch.tensor([s[-1] for s in seqs], dtype=torch.long)[:, None]
        logits = decoder_forward(nxt, cache, xkv, W, dims)[:, -1]
    out_t, out_s = [], []
    for a in range(A):
        pool = dict(finished[a])
        if len(pool) < beam:  # not enough finished hypotheses: add the live beams, best first
            for j in sorted(range(beam), key=lambda j: -sums[a * beam + j]):
                if len(pool) >= beam:
                    break
                pool.setdefault(tuple(seqs[a * beam + j]), sums[a * beam + j])
        best = max(pool, key=lambda k: (pool[k] / max(len(k), 1), tuple(-t for t in k)))
        out_t.append([t for t in best if t != rules.eot])
        out_s.append(pool[best])
    return BeamResult(out_t, out_s, no_speech)


# --------------------------------------------------------------------------------------------------
# a11 (fallback ladder): temperature sampling.  faster-whisper's generate_with_fallback samples with
# best_of hypotheses at temperatures 0.2 ... 1.0 when the greedy/beam result fails the compression-ratio or
# log-prob thresholds (un-vendored; thresholds 2.4 / -1.0 / 0.6).  The draw is
# Gumbel-max over processed_logits / T with a counter-based uniform u(seed, row, position, token) - the same
# integer hash the HIP select kernel evaluates, so a sampled decode is reproducible on both sides.
# --------------------------------------------------------------------------------------------------
def _pcg_hash(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF
    x = (x * 747796405 + 2891336453) & 0xFFFFFFFF
    w = (((x >> ((x >> 28) + 4)) ^ x) * 277803737) & 0xFFFFFFFF
    return ((w >> 22) ^ w) & 0xFFFFFFFF


def sample_gumbel(seed: int, row: int, position: int, n: int) -> np.ndarray:
    key = _pcg_hash(np.uint64((seed & 0xFFFFFFFF) ^ int(_pcg_hash(np.uint64((row * 0x9E3779B9 + position) & 0xFFFFFFFF)))))
    h = _pcg_hash((int(key) + np.arange(n, dtype=np.uint64)) & 0xFFFFFFFF)
    u = (h >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0) + np.float32(0.5 / 16777216.0)
    return -np.log(-np.log(u, dtype=np.float32), dtype=np.float32)


def sample_decode(enc: torch.Tensor, prompt: Sequence[int], W: Dict[str, torch.Tensor], dims: Dims, rules: Rules,
                  best_of: int, temperature: float, seed: int, max_new_tokens: int) -> GreedyResult:
    """best_of independently sampled rows per clip; returns, per clip, the row with the best sum_logprob / length.
    sum_logprob accumulates log_softmax of the processed (untempered) logits at the drawn token."""
    A = enc.shape[0]
    R = A * best_of
    xkv_a = cross_kv(enc, W, dims)
    xkv = [(k.repeat_interleave(best_of, dim=0), v.repeat_interleave(best_of, dim=0)) for k, v in xkv_a]
    cache = SelfCache.empty(dims.dec_layers)
    logits = None
    for t in prompt:
        logits = decoder_forward(torch.full((R, 1), t, dtype=torch.long), cache, xkv, W, dims)[:, -1]
    sampled: List[List[int]] = [[] for _ in range(R)]
    done = [False] * R
    sum_lp = [0.0] * R
    for _ in range(max_new_tokens):
        position = cache.length - 1  # position of the token whose logits we hold
        nxt = []
        for r in range(R):
            if done[r]:
                nxt.append(rules.eot)
                continue
            s = apply_rules(logits[r], sampled[r], rules)
            z = (s / np.float32(temperature)).numpy() + sample_gumbel(seed, r, position, s.shape[0])
            t = int(np.argmax(z))
            sum_lp[r] += float(torch.log_softmax(s, dim=-1)[t])
            sampled[r].append(t)
            done[r] = t == rules.eot
            nxt.append(t)
        if all(done) or cache.length >= dims.n_text_ctx:
            break
        logits = decoder_forward(torch.tensor(nxt, dtype=torch.long)[:, None], cache, xkv, W, dims)[:, -1]
    toks, lps = [], []
    for a in range(A):
        rows = range(a * best_of, (a + 1) * best_of)
        best = max(rows, key=lambda r: (sum_lp[r] / max(len(sampled[r]), 1), -r))
        toks.append(sampled[best]); lps.append(sum_lp[best])
    return GreedyResult(toks, lps, [0.0] * A)


# --------------------------------------------------------------------------------------------------
# a2 (.words): token-level timestamps from cross-attention + dynamic time warping.  faster-whisper's
# `find_alignment` calls CTranslate2 `Whisper.align` (un-vendored); HF restates the same published algorithm in
# [HF-G] generation_whisper.py `_extract_token_timestamps` / `_median_filter` / `_dynamic_time_warping`, which is what
# is followed (and pinned) here: select the alignment heads, crop to num_frames // 2, drop the prompt rows, normalise
# each head over the TOKEN axis, median-filter along time (width 7, reflect), average the heads, DTW on the negated
# matrix, a token starts where the path first enters its row.
# --------------------------------------------------------------------------------------------------
def alignment_weights(enc: torch.Tensor, tokens: Sequence[int], W: Dict[str, torch.Tensor], dims: Dims,
                      heads: Sequence[Tuple[int, int]], return_logprobs: bool = False):
    """Teacher-forced pass over `tokens` for ONE clip: cross-attention probabilities of the (layer, head) pairs,
    [n_pairs, n_tok, T]; optionally log p(tokens[i+1] | tokens[:i+1]) from the raw logits, [n_tok - 1]."""
    xkv = cross_kv(enc, W, dims)
    cache = SelfCache.empty(dims.dec_layers)
    probs: List[torch.Tensor] = []
    logits = decoder_forward(torch.tensor([list(tokens)]), cache, xkv, W, dims, cross_probs=probs)
    w = torch.stack([probs[l][0, h] for l, h in heads])
    if not return_logprobs:
        return w
    lp = torch.log_softmax(logits[0, :-1], dim=-1)
    return w, lp[torch.arange(len(tokens) - 1), torch.tensor(list(tokens[1:]))]


def median_filter(x: torch.Tensor, width: int) -> torch.Tensor:
    pad = width // 2
    if x.shape[-1] <= pad:
        return x
    xp = F.pad(x, (pad, pad, 0, 0), mode="reflect")
    return xp.unfold(-1, width, 1).sort()[0][..., pad]


def dtw_path(cost: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Monotone alignment path of minimum total cost through an [n_tok, n_frames] matrix; on ties the horizontal
    move wins, then the vertical one (the branch order of the published implementation)."""
    n, m = cost.shape
    acc = np.full((n + 1, m + 1), np.inf, dtype=np.float32)
    trace = -np.ones((n + 1, m + 1), dtype=np.int8)
    acc[0, 0] = 0
    for j in range(1, m + 1):
        for i in range(1, n + 1):
            c0, c1, c2 = acc[i - 1, j - 1], acc[i - 1, j], acc[i, j - 1]
            if c0 < c1 and c0 < c2:
                c, t = c0, 0
            elif c1 < c0 and c1 < c2:
                c, t = c1, 1
            else:
                c, t = c2, 2
            acc[i, j] = np.float32(cost[i - 1, j - 1]) + c
            trace[i, j] = t
    trace[0, :] = 2
    trace[:, 0] = 1
    i, j, ti, tj = n, m, [], []
    while i > 0 or j > 0:
        ti.append(i - 1)
        tj.append(j - 1)
        t = trace[i, j]
        if t == 0:
            i, j = i - 1, j - 1
        elif t == 1:
            i -= 1
        else:
            j -= 1
    return np.array(ti)[::-1], np.array(tj)[::-1]


def token_timestamps(weights: torch.Tensor, n_prefix: int, num_frames: Optional[int] = None, medfilt: int = 7,
                     time_precision: float = 0.02) -> np.ndarray:
    """weights [n_pairs, n_tok, T] -> start time of every token after the first n_prefix (seconds)."""
    w = weights if num_frames is None else weights[..., : num_frames // 2]
    w = w[:, n_prefix:, :]
    std = torch.std(w, dim=-2, keepdim=True, unbiased=False)
    mean = torch.mean(w, dim=-2, keepdim=True)
    w = median_filter((w - mean) / std, medfilt).mean(dim=0)
    ti, tj = dtw_path(-w.double().numpy())
    jumps = np.pad(np.diff(ti), (1, 0), constant_values=1).astype(bool)
    return tj[jumps] * time_precision
