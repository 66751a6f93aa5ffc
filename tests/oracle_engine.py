"""TEST infrastructure: an object with the Engine methods WhisperModel's host-side window loop calls, computed by the CPU
oracle (oracle/whisper_ref.py).  It lets the loop (seek / segment split / previous-text prompt / fallback ladder - SURVEY.md
row a11) be pinned against HF long-form goldens in the CPU suite; the GPU suite runs the same checks on the real HIP
engine.  Never imported by the product."""
from __future__ import annotations

import types
from typing import List, Sequence

import numpy as np
import torch

from oracle import whisper_ref as R
from taiwan_tongues_asr_ce_amd.config import SpecialTokens
from taiwan_tongues_asr_ce_amd.engine import GenResult, default_suppress

torch.set_grad_enabled(False)


class OracleEngine:
    def __init__(self, dims, compute_type=0, max_batch=1, device=0):
        self.dims = dims
        self.rd = R.Dims(**dims.as_dict())
        self.compute_type, self.max_batch = compute_type, max_batch
        self.special = SpecialTokens.for_vocab(dims.vocab)
        self.audio_ctx = dims.n_audio_ctx
        self.W = None
        self.enc = None
        self.calls: List[tuple] = []          # (method, detail) trace for tests

    def load_weights(self, tensors):
        self.W = R.to_torch({k: np.asarray(v, dtype=np.float32) for k, v in tensors})

    def close(self):
        pass

    def set_audio_ctx(self, n_ctx=0):
        assert not n_ctx or n_ctx == self.dims.n_audio_ctx, "the oracle engine only runs the full window"

    def log_mel(self, clips: Sequence[np.ndarray], want_output=True):
        n = 2 * self.audio_ctx * 160
        self.mel = np.stack([R.log_mel(np.asarray(c, dtype=np.float32), self.dims.n_mels, n) for c in clips])
        return self.mel if want_output else None

    def log_mel_windows(self, audio, seeks, floor_max=None, want_output=False, want_max=False):
        """Whole-file features (R.log_mel_file), sliced per window with feature-space zero padding; floor_max is implied
        (the oracle computes the file's features in one piece, so its floor IS the whole-file one)."""
        files = [audio] * len(seeks) if isinstance(audio, np.ndarray) else list(audio)
        if not hasattr(self, "_feat"):
            self._feat = {}
        mels, mx = [], []
        for a, k in zip(files, seeks):
            key = (id(a), len(a))
            if key not in self._feat:
                self._feat[key] = R.log_mel_file(np.asarray(a, dtype=np.float32), self.dims.n_mels)
            f = self._feat[key]
            mels.append(R.file_window(f, int(k), 2 * self.audio_ctx))
            mx.append((f[:, int(k):int(k) + 2 * self.audio_ctx] * 4.0 - 4.0).max())   # un-normalised log10 maximum
        self.mel = np.stack(mels)
        return (self.mel if want_output else None), (np.asarray(mx, dtype=np.float32) if want_max else None)

    def encode(self, B, want_output=False):
        self.enc = R.encoder_forward(torch.from_numpy(self.mel[:B]), self.W, self.rd)
        return self.enc.numpy() if want_output else None

    def gen_opts(self, max_new_tokens, timestamps, suppress=None, begin_suppress=None, suppress_eot=False, no_speech=True,
                 sot_index=0, max_initial_timestamp_index=50, check_interval=8):
        st = self.special
        return types.SimpleNamespace(
            max_new_tokens=max_new_tokens, timestamps=bool(timestamps),
            suppress=list(default_suppress(st, self.dims.vocab) if suppress is None else suppress),
            begin_suppress=list([220, st.eot] if begin_suppress is None else begin_suppress), suppress_eot=suppress_eot,
            no_speech=no_speech, sot_index=sot_index, max_initial_timestamp_index=max_initial_timestamp_index)

    def _rules(self, o):
        st = self.special
        return R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin, suppress=o.suppress,
                       begin_suppress=o.begin_suppress, timestamps=o.timestamps,
                       max_initial_timestamp_index=o.max_initial_timestamp_index, suppress_eot=o.suppress_eot)

    def generate(self, prompts, opts) -> GenResult:
        toks, lps, nss = [], [], []
        for b, p in enumerate(prompts):
            self.calls.append(("generate", list(p)))
            r = R.greedy_decode(self.enc[b:b + 1], list(p), self.W, self.rd, self._rules(opts), opts.max_new_tokens,
                                no_speech_token=self.special.no_speech if opts.no_speech else None, sot_index=opts.sot_index)
            toks.append(r.tokens[0]); lps.append(r.sum_logprob[0]); nss.append(r.no_speech_prob[0])
        return GenResult(toks, np.asarray(lps, np.float32), np.asarray(nss, np.float32))

    def generate_beam(self, prompts, beam, opts, patience=1.0, sot_index=None) -> GenResult:
        toks, lps, nss = [], [], []
        for b, p in enumerate(prompts):
            self.calls.append(("generate_beam", list(p)))
            r = R.beam_decode(self.enc[b:b + 1], list(p), self.W, self.rd, self._rules(opts), beam, opts.max_new_tokens,
                              patience=patience, no_speech_token=self.special.no_speech if opts.no_speech else None,
                              sot_index=opts.sot_index if sot_index is None else sot_index[b])
            toks.append(r.tokens[0]); lps.append(r.sum_logprob[0]); nss.append(r.no_speech_prob[0])
        return GenResult(toks, np.asarray(lps, np.float32), np.asarray(nss, np.float32))

    def generate_sample(self, prompts, best_of, opts, temperature, seed=0) -> GenResult:
        toks, lps, nss = [], [], []
        for b, p in enumerate(prompts):
            self.calls.append(("generate_sample", temperature))
            r = R.sample_decode(self.enc[b:b + 1], list(p), self.W, self.rd, self._rules(opts), best_of, temperature, seed,
                                opts.max_new_tokens)
            toks.append(r.tokens[0]); lps.append(r.sum_logprob[0]); nss.append(r.no_speech_prob[0])
        return GenResult(toks, np.asarray(lps, np.float32), np.asarray(nss, np.float32))
