"""Shared checkers for the GPU parity tests (test infrastructure: imports oracle/).

teacher_forced(): feed the oracle the ENGINE's tokens and grade every choice of the engine against the oracle's processed
logits at that step -
  * the chosen token is allowed and within `tol` of the oracle's best allowed logit;
  * wherever the oracle's own top-2 margin exceeds `margin` (= 2 x the logit tolerance of the compute mode) the token must
    BE the oracle's argmax (token equality under margin);
and count how many steps carried such a clear margin, so callers can assert the test is not vacuous."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence

import numpy as np
import torch

from oracle import whisper_ref as R


@dataclass
class Graded:
    n_steps: int = 0
    n_clear: int = 0          # steps whose oracle margin exceeded `margin` (held to token equality)
    worst: float = 0.0        # largest (best allowed logit - chosen logit)

    def add(self, other: "Graded"):
        self.n_steps += other.n_steps
        self.n_clear += other.n_clear
        self.worst = max(self.worst, other.worst)


def encode_chunked(mel: np.ndarray, W, rd, chunk: int = 4) -> torch.Tensor:
    """Oracle encoder over a batch in chunks (the [B, H, T, T] score tensor of a 30-s window is 180 MB per clip at 20 heads)."""
    outs = [R.encoder_forward(torch.from_numpy(mel[i:i + chunk]), W, rd) for i in range(0, len(mel), chunk)]
    return torch.cat(outs, dim=0)


def prompt_state(prompt: Sequence[int], enc_ref: torch.Tensor, W, rd):
    """(cross-KV, self-attention cache after the prompt, logits of its last position, logits of every prompt position): the part
    of a grading that does not depend on the graded tokens - compute it once per clip and hand it to several teacher_forced()
    calls (the full-depth oracle reads 3.6 GB of decoder weights per position)."""
    B = enc_ref.shape[0]
    xkv = R.cross_kv(enc_ref, W, rd)
    cache = R.SelfCache.empty(rd.dec_layers)
    per_pos = []
    for t in prompt:
        per_pos.append(R.decoder_forward(torch.full((B, 1), int(t), dtype=torch.long), cache, xkv, W, rd)[:, 0])
    return xkv, cache, per_pos[-1], per_pos


def teacher_forced(tokens: Sequence[Sequence[int]], prompt: Sequence[int], enc_ref: torch.Tensor, W, rd, rules, tol: float,
                   margin: float, n_check: int | None = None, start=None) -> Graded:
    """Grade rows `tokens[b]` (engine output for clip b of enc_ref) as described in the module docstring.  `start` = a
    prompt_state() result to reuse (its cache is not modified: decoder_forward replaces cache entries, never writes in place)."""
    B = len(tokens)
    assert enc_ref.shape[0] == B
    if start is None:
        start = prompt_state(prompt, enc_ref, W, rd)
    xkv, cache0, logits = start[0], start[1], start[2]
    cache = R.SelfCache(list(cache0.k), list(cache0.v))
    n = min(len(t) for t in tokens)
    if n_check is not None:
        n = min(n, n_check)
    g = Graded()
    for i in range(n):
        nxt: List[int] = []
        for b in range(B):
            c = int(tokens[b][i])
            g.add(_grade_choice(logits[b], list(tokens[b][:i]), c, rules, tol, margin, f"row {b} step {i}"))
            nxt.append(c)
        if i + 1 < n:
            logits = R.decoder_forward(torch.tensor(nxt, dtype=torch.long)[:, None], cache, xkv, W, rd)[:, 0]
    return g


def _grade_choice(row_logits, sampled, c, rules, tol, margin, where) -> Graded:
    """One choice `c` of the engine against the oracle's raw logits of that position.  The timestamp-probability rule is the one
    discontinuous processor (text is masked when the timestamp probability MASS exceeds the best text probability): when the
    oracle's own margin of that comparison is inside `tol`, a 16-bit engine may legitimately land on the other side, so the choice
    is graded under the rule outcome the ENGINE took (text token -> rule off, timestamp -> rule on) and the step is not counted
    as clear."""
    g = Graded(n_steps=1)
    mg: list = []
    s = np.asarray(R.apply_rules(row_logits, sampled, rules, ts_prob_margin=mg))
    near_rule = bool(mg) and abs(mg[0]) < tol
    if near_rule:
        s = np.asarray(R.apply_rules(row_logits, sampled, rules, ts_prob_rule=bool(c >= rules.timestamp_begin)))
    assert s[c] > -np.inf, f"{where}: the engine chose a masked token ({c}); timestamp-rule margin {mg[0] if mg else None}"
    gap = float(s.max() - s[c])
    assert gap < tol, f"{where}: chosen logit {gap:.4f} below the oracle's best (tolerance {tol})"
    g.worst = gap
    top2 = np.partition(s, -2)[-2:]
    if top2[1] - top2[0] > margin and not near_rule:
        assert int(np.argmax(s)) == c, f"{where}: token {c} != oracle argmax {int(np.argmax(s))} at a clear margin"
        g.n_clear = 1
    return g


def teacher_forced_causal(tokens: Sequence[Sequence[int]], prompt: Sequence[int], enc_ref: torch.Tensor, W, rd, rules, tol: float,
                          margin: float, rows_per_pass: int = 4) -> Graded:
    """teacher_forced() for LONG rows: instead of one oracle step per position, ONE causal oracle pass per row over
    prompt + tokens[:-1] gives the logits of every position at once (the `_teacher_forced_gap` pattern of
    test_gpu_max_context.py, batched over `rows_per_pass` rows of equal length).  Same grading: every choice allowed and
    within `tol` of the oracle's best allowed logit; equal to the oracle's argmax wherever its top-2 margin exceeds `margin`."""
    B = len(tokens)
    assert enc_ref.shape[0] == B
    n = len(tokens[0])
    assert all(len(t) == n for t in tokens), "rows of one causal pass must have one length"
    g = Graded()
    P = len(prompt)
    for lo in range(0, B, rows_per_pass):
        hi = min(B, lo + rows_per_pass)
        xkv = R.cross_kv(enc_ref[lo:hi], W, rd)
        seq = torch.tensor([list(prompt) + [int(t) for t in tokens[b][:-1]] for b in range(lo, hi)], dtype=torch.long)
        logits = R.decoder_forward(seq, R.SelfCache.empty(rd.dec_layers), xkv, W, rd)
        for b in range(lo, hi):
            toks = [int(t) for t in tokens[b]]
            for i, c in enumerate(toks):
                g.add(_grade_choice(logits[b - lo, P - 1 + i], toks[:i], c, rules, tol, margin, f"row {b} step {i}"))
        del logits
    return g
