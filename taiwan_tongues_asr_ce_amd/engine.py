"""Thin Python owner of one libttasr context (one GPU, one stream).  numpy in / numpy out; every compute
method is a single C-ABI call into the HIP library."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .config import COMPUTE_BF16, COMPUTE_F16, COMPUTE_F32, NON_SPEECH_TOKENS_MULTI, SpecialTokens, WhisperDims


class TtasrError(RuntimeError):
    pass


@dataclass
class DeviceTensor:
    """A tensor in device memory of the engine's GPU: address, TTASR_DTYPE_* (0 float32, 1 bfloat16 bits, 2 float16 bits), shape."""
    ptr: int
    dtype: int
    shape: Tuple[int, ...]


@dataclass
class GenResult:
    tokens: List[List[int]]
    sum_logprob: np.ndarray
    no_speech_prob: np.ndarray


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def default_suppress(st: SpecialTokens, vocab: int) -> List[int]:
    """suppress_tokens=[-1] of faster-whisper (get_suppressed_tokens): the non-speech symbols plus <|transcribe|>,
    <|translate|>, <|startoftranscript|>, <|startofprev|>, <|startoflm|>; <|nospeech|> is masked as well, as in the
    generation_config.json of the released checkpoints and openai-whisper (its probability is read from the raw logits
    before the mask, and it is never a legitimate output)."""
    ids = [t for t in NON_SPEECH_TOKENS_MULTI if t < min(vocab, st.eot)]
    ids += [st.translate, st.transcribe, st.sot, st.sot_prev, st.no_speech]
    if st.sot_lm >= 0:
        ids.append(st.sot_lm)
    return sorted(set(ids))


class Engine:
    def __init__(self, dims: WhisperDims, compute_type: int = COMPUTE_BF16, max_batch: int = 1, device: int = 0,
                 share_weights_with: Optional["Engine"] = None):
        """share_weights_with: another Engine on the same GPU whose (finalized) device weights this one reads instead of loading
        its own copy (ttasr_create_shared): a second context for keeping two batches in flight costs workspaces only."""
        self.lib = _lib.load()
        self.dims = dims
        self.compute_type = compute_type
        self.max_batch = max_batch
        self.device = device
        h = C.c_void_p()
        if share_weights_with is not None:
            o = share_weights_with
            if o.dims != dims or o.compute_type != compute_type or o.device != device:
                raise ValueError("a weight-sharing engine must have its owner's geometry, compute type and device")
            rc = self.lib.ttasr_create_shared(o.h, max_batch, C.byref(h))
            what = "ttasr_create_shared"
        else:
            cfg = _lib.Config(dims.n_mels, dims.n_audio_ctx, dims.d_model, dims.n_heads, dims.ffn_dim, dims.enc_layers,
                              dims.dec_layers, dims.vocab, dims.n_text_ctx, compute_type, max_batch, 0)
            rc = self.lib.ttasr_create(C.byref(cfg), device, C.byref(h))
            what = "ttasr_create"
        if rc != 0:
            raise TtasrError(f"{what} failed ({rc}): {self.lib.ttasr_last_error(None).decode()}")
        self.h = h
        self.shares_weights = share_weights_with is not None
        self.special = SpecialTokens.for_vocab(dims.vocab)
        self.audio_ctx = dims.n_audio_ctx

    # -- plumbing ------------------------------------------------------------------------------
    def _check(self, rc: int, what: str):
        if rc != 0:
            raise TtasrError(f"{what} failed ({rc}): {self.lib.ttasr_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.ttasr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights -------------------------------------------------------------------------------
    def load_weights(self, tensors: Iterable[Tuple[str, "np.ndarray | DeviceTensor"]]):
        """(name, float32 host array) pairs, or (name, DeviceTensor) for tensors already resident in this GPU's memory
        (the multi-GPU start-up: dist.broadcast_weights hands over the RCCL buckets without a host round trip)."""
        for name, arr in tensors:
            if isinstance(arr, DeviceTensor):
                dims = (C.c_int64 * len(arr.shape))(*arr.shape)
                self._check(self.lib.ttasr_load_tensor_device(self.h, name.encode(), C.c_void_p(arr.ptr), arr.dtype, dims,
                                                              len(arr.shape)), f"load_tensor_device({name})")
                continue
            a = np.ascontiguousarray(arr, dtype=np.float32)
            dims = (C.c_int64 * a.ndim)(*a.shape)
            self._check(self.lib.ttasr_load_tensor(self.h, name.encode(), _ptr(a), dims, a.ndim), f"load_tensor({name})")
        self._check(self.lib.ttasr_finalize_weights(self.h), "finalize_weights")

    # -- a5 ------------------------------------------------------------------------------------
    def set_audio_ctx(self, n_ctx: int = 0):
        """Opt-in short window (ttasr_set_audio_ctx): the next log_mel/encode/generate use n_ctx encoder positions
        (n_ctx * 320 samples); 0 restores the model's 30-s window."""
        n = int(n_ctx) or self.dims.n_audio_ctx
        self._check(self.lib.ttasr_set_audio_ctx(self.h, n), "set_audio_ctx")
        self.audio_ctx = n

    def log_mel(self, clips: Sequence[np.ndarray], want_output: bool = True) -> Optional[np.ndarray]:
        B = len(clips)
        n_win = 2 * self.audio_ctx * 160
        stride = max(1, min(n_win, max((len(c) for c in clips), default=1)))
        pcm = np.zeros((B, stride), dtype=np.float32)
        ns = np.zeros(B, dtype=np.int64)
        for b, c in enumerate(clips):
            n = min(len(c), stride)
            pcm[b, :n] = np.asarray(c[:n], dtype=np.float32)
            ns[b] = n
        out = np.empty((B, self.dims.n_mels, 2 * self.audio_ctx), dtype=np.float32) if want_output else None
        self._check(self.lib.ttasr_log_mel(self.h, _ptr(pcm), stride, ns.ctypes.data_as(C.POINTER(C.c_int64)), B, 0,
                                           _ptr(out) if want_output else None), "log_mel")
        return out

    def log_mel_windows(self, audio, seeks: Sequence[int], floor_max: Optional[Sequence[float]] = None,
                        want_output: bool = False, want_max: bool = False):
        """Windows of recordings (ttasr_log_mel_windows): window b starts at 10-ms frame seeks[b] of `audio` (one float32
        array = all windows belong to that recording) or of audio[b] (a sequence: one recording per window, several files in
        lock step); frames are those of the whole-file STFT, frames past the end of the recording are 0 in feature space,
        and the dynamic-range floor comes from floor_max[b] (the whole-file log-mel maximum) when given.
        Returns (mel or None, per-window maxima or None)."""
        sk = np.ascontiguousarray(seeks, dtype=np.int64)
        B = len(sk)
        files = [audio] * B if isinstance(audio, np.ndarray) else list(audio)
        files = [np.ascontiguousarray(a, dtype=np.float32) for a in files]
        ptrs = (C.c_void_p * B)(*[a.ctypes.data for a in files])
        lens = np.asarray([len(a) for a in files], dtype=np.int64)
        fm = None if floor_max is None else np.ascontiguousarray(floor_max, dtype=np.float32)
        out = np.empty((B, self.dims.n_mels, 2 * self.audio_ctx), dtype=np.float32) if want_output else None
        mx = np.empty(B, dtype=np.float32) if want_max else None
        i64p = C.POINTER(C.c_int64)
        self._check(self.lib.ttasr_log_mel_windows(self.h, ptrs, lens.ctypes.data_as(i64p), sk.ctypes.data_as(i64p), B,
                                                   _ptr(fm) if fm is not None else None, _ptr(mx) if want_max else None,
                                                   _ptr(out) if want_output else None), "log_mel_windows")
        return out, mx

    def log_mel_device(self, dev_ptr: int, stride: int, n_samples: Sequence[int]):
        """PCM already resident in HBM (bench): dev_ptr = device address of float32 [B][stride]."""
        ns = np.asarray(n_samples, dtype=np.int64)
        self._check(self.lib.ttasr_log_mel(self.h, C.c_void_p(dev_ptr), stride, ns.ctypes.data_as(C.POINTER(C.c_int64)),
                                           len(ns), 1, None), "log_mel(device)")

    def log_mel_host_ptr(self, host_ptr: int, stride: int, n_samples: Sequence[int]):
        """PCM in a caller-owned (ideally pinned) host buffer: the call includes the H2D copy."""
        ns = np.asarray(n_samples, dtype=np.int64)
        self._check(self.lib.ttasr_log_mel(self.h, C.c_void_p(host_ptr), stride, ns.ctypes.data_as(C.POINTER(C.c_int64)),
                                           len(ns), 0, None), "log_mel(host)")

    def set_mel(self, mel: np.ndarray):
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        assert mel.shape[1:] == (self.dims.n_mels, 2 * self.audio_ctx), mel.shape
        self._check(self.lib.ttasr_set_mel(self.h, _ptr(mel), mel.shape[0]), "set_mel")

    def set_option(self, key: str, value: int):
        """Kernel-selection override for tests / A-B measurements (ttasr_set_option; keys in include/ttasr.h).  The library
        reads no environment variable: this call is the only way to leave the measured configuration."""
        self._check(self.lib.ttasr_set_option(self.h, key.encode(), int(value)), f"set_option({key})")

    # -- a6..a8 --------------------------------------------------------------------------------
    def encode(self, B: int, want_output: bool = False) -> Optional[np.ndarray]:
        out = np.empty((B, self.audio_ctx, self.dims.d_model), dtype=np.float32) if want_output else None
        self._check(self.lib.ttasr_encode(self.h, B, _ptr(out) if want_output else None), "encode")
        return out

    def set_encoder_output(self, enc: np.ndarray):
        enc = np.ascontiguousarray(enc, dtype=np.float32)
        self._check(self.lib.ttasr_set_encoder_output(self.h, _ptr(enc), enc.shape[0]), "set_encoder_output")

    def cross_kv(self, layer: int, which: int, B: int) -> np.ndarray:
        out = np.empty((B, self.dims.n_heads, self.audio_ctx, 64), dtype=np.float32)
        self._check(self.lib.ttasr_get_cross_kv(self.h, layer, which, B, _ptr(out)), "get_cross_kv")
        return out

    # -- a9, a10 -------------------------------------------------------------------------------
    def gen_opts(self, max_new_tokens: int, timestamps: bool, suppress: Optional[Sequence[int]] = None,
                 begin_suppress: Optional[Sequence[int]] = None, suppress_eot: bool = False, no_speech: bool = True,
                 sot_index: int = 0, max_initial_timestamp_index: Optional[int] = 50, check_interval: int = 8):
        st = self.special
        sup = np.asarray(default_suppress(st, self.dims.vocab) if suppress is None else list(suppress), dtype=np.int32)
        bsup = np.asarray([220, st.eot] if begin_suppress is None else list(begin_suppress), dtype=np.int32)
        o = _lib.GenOpts()
        o.max_new_tokens = max_new_tokens
        o.eot, o.no_timestamps, o.timestamp_begin = st.eot, st.no_timestamps, st.timestamp_begin
        o.no_speech = st.no_speech if no_speech else -1
        o.sot_index = sot_index
        o.timestamps = int(timestamps)
        o.max_initial_timestamp_index = -1 if max_initial_timestamp_index is None else max_initial_timestamp_index
        o.suppress_eot = int(suppress_eot)
        o.n_suppress, o.n_begin_suppress = len(sup), len(bsup)
        o.check_interval = check_interval
        o.suppress = sup.ctypes.data_as(C.POINTER(C.c_int32))
        o.begin_suppress = bsup.ctypes.data_as(C.POINTER(C.c_int32))
        o._keep = (sup, bsup)  # keep the arrays alive
        return o

    def generate(self, prompts: Sequence[Sequence[int]], opts, row_max_new: Optional[Sequence[int]] = None) -> GenResult:
        """Greedy search.  row_max_new (optional, one entry per row, each in [1, opts.max_new_tokens]) = per-row token budgets
        (ttasr_generate_capped): a row is finished at its budget or at EOT, and finished rows leave the decode step's attention
        kernels - the rows that go on are bit-identical to a run without budgets."""
        B = len(prompts)
        max_prompt = max(len(p) for p in prompts)
        pr = np.zeros((B, max_prompt), dtype=np.int32)
        pl = np.zeros(B, dtype=np.int32)
        for b, p in enumerate(prompts):
            pr[b, :len(p)] = p
            pl[b] = len(p)
        toks = np.zeros((B, opts.max_new_tokens), dtype=np.int32)
        lens = np.zeros(B, dtype=np.int32)
        lp = np.zeros(B, dtype=np.float32)
        ns = np.zeros(B, dtype=np.float32)
        i32p, f32p = C.POINTER(C.c_int32), C.POINTER(C.c_float)
        if row_max_new is not None:
            caps = np.ascontiguousarray(row_max_new, dtype=np.int32)
            if caps.shape != (B,):
                raise ValueError(f"row_max_new needs one entry per row ({B}), got shape {caps.shape}")
            self._check(self.lib.ttasr_generate_capped(self.h, B, pr.ctypes.data_as(i32p), pl.ctypes.data_as(i32p), max_prompt,
                                                       C.byref(opts), caps.ctypes.data_as(i32p), toks.ctypes.data_as(i32p),
                                                       lens.ctypes.data_as(i32p), lp.ctypes.data_as(f32p),
                                                       ns.ctypes.data_as(f32p)), "generate_capped")
            return GenResult([toks[b, :lens[b]].tolist() for b in range(B)], lp, ns)
        self._check(self.lib.ttasr_generate(self.h, B, pr.ctypes.data_as(i32p), pl.ctypes.data_as(i32p), max_prompt,
                                            C.byref(opts), toks.ctypes.data_as(i32p), lens.ctypes.data_as(i32p),
                                            lp.ctypes.data_as(f32p), ns.ctypes.data_as(f32p)), "generate")
        return GenResult([toks[b, :lens[b]].tolist() for b in range(B)], lp, ns)

    def generate_beam(self, prompts: Sequence[Sequence[int]], beam: int, opts, patience: float = 1.0,
                      sot_index: Optional[Sequence[int]] = None) -> GenResult:
        """Beam search over len(prompts) clips; rows = clips * beam <= max_batch.  Prompts may differ in length (one
        previous-text prompt per file); sot_index then gives each prompt's <|startoftranscript|> position."""
        A = len(prompts)
        i32p, f32p = C.POINTER(C.c_int32), C.POINTER(C.c_float)
        toks = np.zeros((A, opts.max_new_tokens), dtype=np.int32)
        lens = np.zeros(A, dtype=np.int32)
        lp = np.zeros(A, dtype=np.float32)
        ns = np.zeros(A, dtype=np.float32)
        plen = len(prompts[0])
        if all(len(p) == plen for p in prompts) and sot_index is None:
            pr = np.asarray(prompts, dtype=np.int32).reshape(A, plen)
            self._check(self.lib.ttasr_generate_beam(self.h, A, beam, pr.ctypes.data_as(i32p), plen, C.byref(opts),
                                                     C.c_float(patience), toks.ctypes.data_as(i32p), lens.ctypes.data_as(i32p),
                                                     lp.ctypes.data_as(f32p), ns.ctypes.data_as(f32p)), "generate_beam")
        else:
            max_prompt = max(len(p) for p in prompts)
            pr = np.zeros((A, max_prompt), dtype=np.int32)
            pl = np.zeros(A, dtype=np.int32)
            for a, p in enumerate(prompts):
                pr[a, :len(p)] = p
                pl[a] = len(p)
            so = None if sot_index is None else np.ascontiguousarray(sot_index, dtype=np.int32)
            self._check(self.lib.ttasr_generate_beam_ragged(
                self.h, A, beam, pr.ctypes.data_as(i32p), pl.ctypes.data_as(i32p),
                so.ctypes.data_as(i32p) if so is not None else None, max_prompt, C.byref(opts), C.c_float(patience),
                toks.ctypes.data_as(i32p), lens.ctypes.data_as(i32p), lp.ctypes.data_as(f32p), ns.ctypes.data_as(f32p)),
                "generate_beam_ragged")
        return GenResult([toks[a, :lens[a]].tolist() for a in range(A)], lp, ns)

    def generate_sample(self, prompts: Sequence[Sequence[int]], best_of: int, opts, temperature: float, seed: int = 0
                        ) -> GenResult:
        """Temperature sampling: best_of rows per clip (shared cross-KV), best average log-prob per clip returned."""
        A = len(prompts)
        plen = len(prompts[0])
        assert all(len(p) == plen for p in prompts), "sampling needs equal-length prompts"
        pr = np.asarray(prompts, dtype=np.int32).reshape(A, plen)
        toks = np.zeros((A, opts.max_new_tokens), dtype=np.int32)
        lens = np.zeros(A, dtype=np.int32)
        lp = np.zeros(A, dtype=np.float32)
        ns = np.zeros(A, dtype=np.float32)
        i32p, f32p = C.POINTER(C.c_int32), C.POINTER(C.c_float)
        self._check(self.lib.ttasr_generate_sample(self.h, A, best_of, pr.ctypes.data_as(i32p), plen, C.byref(opts),
                                                   C.c_float(temperature), C.c_uint32(seed & 0xFFFFFFFF),
                                                   toks.ctypes.data_as(i32p), lens.ctypes.data_as(i32p),
                                                   lp.ctypes.data_as(f32p), ns.ctypes.data_as(f32p)), "generate_sample")
        return GenResult([toks[a, :lens[a]].tolist() for a in range(A)], lp, ns)

    def align(self, clip: int, tokens: Sequence[int], heads: Sequence[Tuple[int, int]], want_logprob: bool = True):
        """Cross-attention rows of the (layer, head) pairs for a teacher-forced token sequence of one resident clip:
        (weights float32 [n_heads][n_tokens][audio_ctx], logprob float32 [n_tokens - 1] or None)."""
        tok = np.ascontiguousarray(tokens, dtype=np.int32)
        pr = np.ascontiguousarray(heads, dtype=np.int32).reshape(-1, 2)
        w = np.empty((len(pr), len(tok), self.audio_ctx), dtype=np.float32)
        lp = np.empty(max(len(tok) - 1, 1), dtype=np.float32) if want_logprob else None
        self._check(self.lib.ttasr_align(self.h, clip, tok.ctypes.data_as(C.POINTER(C.c_int32)), len(tok),
                                         pr.ctypes.data_as(C.POINTER(C.c_int32)), len(pr),
                                         w.ctypes.data_as(C.POINTER(C.c_float)),
                                         lp.ctypes.data_as(C.POINTER(C.c_float)) if want_logprob else None), "align")
        return w, (lp[: len(tok) - 1] if want_logprob else None)

    def decode_reset(self, B: int):
        self._check(self.lib.ttasr_decode_reset(self.h, B), "decode_reset")

    def decode_step(self, tokens: Sequence[int], want_logits: bool = True) -> Optional[np.ndarray]:
        t = np.asarray(tokens, dtype=np.int32)
        out = np.empty((len(t), self.dims.vocab), dtype=np.float32) if want_logits else None
        self._check(self.lib.ttasr_decode_step(self.h, t.ctypes.data_as(C.POINTER(C.c_int32)), len(t),
                                               _ptr(out) if want_logits else None), "decode_step")
        return out

    def apply_rules(self, rows: np.ndarray, hist: np.ndarray, opts) -> Tuple[np.ndarray, np.ndarray]:
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        hist = np.ascontiguousarray(hist, dtype=np.int32)
        n = rows.shape[0]
        out = np.empty_like(rows)
        choice = np.empty(n, dtype=np.int32)
        self._check(self.lib.ttasr_apply_rules(self.h, _ptr(rows), hist.ctypes.data_as(C.POINTER(C.c_int32)), hist.shape[1],
                                               n, C.byref(opts), _ptr(out), choice.ctypes.data_as(C.POINTER(C.c_int32))),
                    "apply_rules")
        return out, choice

    # -- measurement ---------------------------------------------------------------------------
    def phase_ms(self) -> Dict[str, float]:
        a = (C.c_float * 4)()
        self._check(self.lib.ttasr_phase_ms(self.h, a), "phase_ms")
        return dict(mel=a[0], encoder=a[1], cross_kv=a[2], decode=a[3])

    def beam_profile(self) -> Dict[str, float]:
        """Host-side split of the last beam search (ttasr_beam_profile): ms enqueueing / waiting for the GPU / selecting, positions."""
        a = (C.c_float * 4)()
        self._check(self.lib.ttasr_beam_profile(self.h, a), "beam_profile")
        return dict(enqueue_ms=a[0], gpu_wait_ms=a[1], host_select_ms=a[2], positions=int(a[3]))

    def encoder_kernel_ms(self) -> Dict[str, float]:
        """Per-class in-situ times of the last encode() run with option enc_kernel_timing = 1 (ttasr_encoder_kernel_ms)."""
        a = (C.c_float * 8)()
        self._check(self.lib.ttasr_encoder_kernel_ms(self.h, a), "encoder_kernel_ms")
        return dict(zip(("conv", "layernorm", "qkv", "attention", "out_proj", "fc1", "fc2", "cross_kv"), (float(v) for v in a)))

    def bench_kernel(self, name: str, B: int, iters: int = 20) -> Dict[str, float]:
        ms, by, fl = C.c_float(), C.c_double(), C.c_double()
        self._check(self.lib.ttasr_bench_kernel(self.h, name.encode(), B, iters, C.byref(ms), C.byref(by), C.byref(fl)),
                    f"bench_kernel({name})")
        buf = C.create_string_buffer(256)
        self._check(self.lib.ttasr_bench_kernel_signature(self.h, buf, 256), "bench_kernel_signature")
        return dict(ms=ms.value, bytes=by.value, flops=fl.value, signature=buf.value.decode())

    def sync(self):
        self._check(self.lib.ttasr_sync(self.h), "sync")
