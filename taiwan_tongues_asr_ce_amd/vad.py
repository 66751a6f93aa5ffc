"""Voice-activity pre-filter (`vad_filter=True` at every reference call site: asr_core.py:163, file_asr.py:284,461,
faster_whisper_asr.py:142) — SURVEY §8 row a4 / §8f N4.

faster-whisper runs the Silero VAD network (an ONNX file shipped inside the package) to get one speech probability
per 32-ms frame, turns the probabilities into speech chunks with Silero's published hysteresis rule, transcribes the
concatenation of the chunks, and maps every timestamp back to the original time line.  Neither the package nor the
network weights exist offline, so:

* everything AROUND the network is restated here — the chunking state machine (`get_speech_timestamps`),
  `collect_chunks`, and the time restoration (`SpeechTimestampsMap`, `restore_speech_timestamps`) — from the published
  algorithm, UNPINNED (no executable reference here; tested on hand-derived cases);
* the per-frame speech probability is pluggable: `speech_prob_fn(audio) -> float32[n_frames]` (e.g. an operator-supplied
  Silero ONNX session); the default `energy_speech_prob` is a plain short-time-energy detector and is NOT equivalent to
  Silero — `WhisperModel.transcribe` says so in a warning when it is used.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

SAMPLING_RATE = 16000
WINDOW = 512          # samples per probability frame (32 ms), Silero's frame at 16 kHz


@dataclass
class VadOptions:
    """faster-whisper's VadOptions defaults."""
    threshold: float = 0.5
    neg_threshold: Optional[float] = None
    min_speech_duration_ms: int = 0
    max_speech_duration_s: float = float("inf")
    min_silence_duration_ms: int = 2000
    speech_pad_ms: int = 400


def energy_speech_prob(audio: np.ndarray, window: int = WINDOW) -> np.ndarray:
    """Stand-in for the Silero network: logistic of the frame's RMS level over an adaptive noise floor
    (10th percentile of the frame levels, at least -60 dBFS).  6 dB above the floor -> 0.5."""
    n = int(np.ceil(len(audio) / window)) if len(audio) else 0
    if n == 0:
        return np.zeros(0, dtype=np.float32)
    padded = np.zeros(n * window, dtype=np.float32)
    padded[: len(audio)] = audio
    rms = np.sqrt(np.mean(padded.reshape(n, window) ** 2, axis=1) + 1e-12)
    db = 20.0 * np.log10(rms)
    floor = max(float(np.percentile(db, 10)), -60.0)
    return (1.0 / (1.0 + np.exp(-(db - floor - 6.0) / 1.5))).astype(np.float32)


def silero_speech_prob_fn(step: Callable[[np.ndarray, np.ndarray], "tuple[float, np.ndarray]"], context: int = 64,
                          state_shape: Sequence[int] = (2, 1, 128)) -> Callable[[np.ndarray], np.ndarray]:
    """Adapter for an operator-supplied Silero VAD network (the ONNX file faster-whisper ships; not available offline):
    `step(frame, state) -> (speech_probability, new_state)` is ONE forward of the network, e.g.
        lambda x, h: (lambda o: (float(o[0].squeeze()), o[1]))(session.run(None, {"input": x, "state": h, "sr": np.array(16000)}))
    and the returned function is what `WhisperModel.vad_speech_prob_fn` / `vad_speech_prob_fn=` expect.  It drives the
    network the way Silero's v5 graph is specified: 512-sample frames (32 ms), each prefixed with the last `context` = 64
    samples of the previous frame (zeros before the first), float32 [1, 576]; the recurrent state ([2, 1, 128], zeros at
    the start of a recording) is threaded from call to call; the last frame is zero-padded."""
    def fn(audio: np.ndarray) -> np.ndarray:
        audio = np.asarray(audio, dtype=np.float32)
        n = int(np.ceil(len(audio) / WINDOW)) if len(audio) else 0
        padded = np.zeros(n * WINDOW, dtype=np.float32)
        padded[: len(audio)] = audio
        state = np.zeros(tuple(state_shape), dtype=np.float32)
        ctx = np.zeros(context, dtype=np.float32)
        out = np.zeros(n, dtype=np.float32)
        for i in range(n):
            frame = padded[i * WINDOW:(i + 1) * WINDOW]
            p, state = step(np.concatenate([ctx, frame])[None, :], state)
            out[i] = float(p)
            ctx = frame[-context:] if context else ctx
        return out
    return fn


def get_speech_timestamps(audio: np.ndarray, options: Optional[VadOptions] = None,
                          speech_prob_fn: Optional[Callable[[np.ndarray], np.ndarray]] = None,
                          sampling_rate: int = SAMPLING_RATE) -> List[Dict[str, int]]:
    """Speech chunks [{'start': sample, 'end': sample}] by Silero's rule: a chunk opens at the first frame with
    p >= threshold, closes once p stayed below neg_threshold (threshold - 0.15) for min_silence_duration_ms, is
    dropped if shorter than min_speech_duration_ms, is cut at the last >= 98 ms pause when it exceeds
    max_speech_duration_s, and is finally padded by speech_pad_ms (gaps shorter than two pads are split in half)."""
    o = options or VadOptions()
    probs = np.asarray((speech_prob_fn or energy_speech_prob)(audio), dtype=np.float32)
    n_audio = len(audio)
    neg = o.neg_threshold if o.neg_threshold is not None else max(o.threshold - 0.15, 0.01)
    min_speech = sampling_rate * o.min_speech_duration_ms / 1000
    pad = int(sampling_rate * o.speech_pad_ms / 1000)
    max_speech = sampling_rate * o.max_speech_duration_s - WINDOW - 2 * pad
    min_silence = sampling_rate * o.min_silence_duration_ms / 1000
    min_silence_at_max = sampling_rate * 98 / 1000
    speeches: List[Dict[str, int]] = []
    cur: Dict[str, int] = {}
    triggered = False
    temp_end = prev_end = next_start = 0
    for i, p in enumerate(probs):
        at = WINDOW * i
        if p >= o.threshold and temp_end:
            temp_end = 0
            if next_start < prev_end:
                next_start = at
        if p >= o.threshold and not triggered:
            triggered = True
            cur["start"] = at
            continue
        if triggered and at - cur["start"] > max_speech:
            if prev_end:
                cur["end"] = prev_end
                speeches.append(cur)
                cur = {}
                if next_start < prev_end:
                    triggered = False
                else:
                    cur["start"] = next_start
                prev_end = next_start = temp_end = 0
            else:
                cur["end"] = at
                speeches.append(cur)
                cur = {}
                prev_end = next_start = temp_end = 0
                triggered = False
                continue
        if p < neg and triggered:
            if not temp_end:
                temp_end = at
            if at - temp_end > min_silence_at_max:
                prev_end = temp_end
            if at - temp_end < min_silence:
                continue
            cur["end"] = temp_end
            if cur["end"] - cur["start"] > min_speech:
                speeches.append(cur)
            cur = {}
            prev_end = next_start = temp_end = 0
            triggered = False
    if cur and n_audio - cur["start"] > min_speech:
        cur["end"] = n_audio
        speeches.append(cur)
    for i, sp in enumerate(speeches):
        if i == 0:
            sp["start"] = int(max(0, sp["start"] - pad))
        if i != len(speeches) - 1:
            gap = speeches[i + 1]["start"] - sp["end"]
            if gap < 2 * pad:
                sp["end"] += int(gap // 2)
                speeches[i + 1]["start"] = int(max(0, speeches[i + 1]["start"] - gap // 2))
            else:
                sp["end"] = int(min(n_audio, sp["end"] + pad))
                speeches[i + 1]["start"] = int(max(0, speeches[i + 1]["start"] - pad))
        else:
            sp["end"] = int(min(n_audio, sp["end"] + pad))
    return speeches


def collect_chunks(audio: np.ndarray, chunks: Sequence[Dict[str, int]]) -> np.ndarray:
    if not chunks:
        return np.zeros(0, dtype=np.float32)
    return np.concatenate([audio[c["start"]: c["end"]] for c in chunks]).astype(np.float32, copy=False)


class SpeechTimestampsMap:
    """Maps a time on the concatenated-speech axis back to the original recording."""

    def __init__(self, chunks: Sequence[Dict[str, int]], sampling_rate: int = SAMPLING_RATE, time_precision: int = 2):
        self.sampling_rate = sampling_rate
        self.time_precision = time_precision
        self.chunk_end_sample: List[int] = []
        self.total_silence_before: List[float] = []
        previous_end = 0
        silent = 0
        for c in chunks:
            silent += c["start"] - previous_end
            previous_end = c["end"]
            self.chunk_end_sample.append(c["end"] - silent)
            self.total_silence_before.append(silent / sampling_rate)

    def get_chunk_index(self, time: float) -> int:
        sample = int(time * self.sampling_rate)
        idx = int(np.searchsorted(self.chunk_end_sample, sample, side="right"))
        return min(idx, len(self.chunk_end_sample) - 1)

    def get_original_time(self, time: float, chunk_index: Optional[int] = None) -> float:
        if not self.chunk_end_sample:
            return round(time, self.time_precision)
        if chunk_index is None:
            chunk_index = self.get_chunk_index(time)
        return round(self.total_silence_before[chunk_index] + time, self.time_precision)


def restore_speech_timestamps(segments, chunks: Sequence[Dict[str, int]], sampling_rate: int = SAMPLING_RATE):
    """Generator over segment-like objects with `_replace` (NamedTuple): start/end (and word times) moved back to the
    original time line.  Words are mapped individually: each word goes with the chunk its MIDDLE falls in, so a word
    is never stretched across a removed pause."""
    ts_map = SpeechTimestampsMap(chunks, sampling_rate)
    for seg in segments:
        if getattr(seg, "words", None):
            words = []
            for w in seg.words:
                idx = ts_map.get_chunk_index((w.start + w.end) / 2)
                words.append(type(w)(start=ts_map.get_original_time(w.start, idx), end=ts_map.get_original_time(w.end, idx),
                                     word=w.word, probability=w.probability))
            yield seg._replace(start=words[0].start, end=words[-1].end, words=words)
        else:
            yield seg._replace(start=ts_map.get_original_time(seg.start), end=ts_map.get_original_time(seg.end))
