"""VAD plumbing around the (absent) Silero network: chunking state machine, chunk collection, time restoration.
Expected values are derived by hand from the published rule (see the comments); the module is unpinned."""
from typing import NamedTuple, Optional, List

import numpy as np
import pytest

from taiwan_tongues_asr_ce_amd import synth, vad

W = vad.WINDOW


def _probs(*runs):
    """runs of (n_frames, probability) -> (audio placeholder of the right length, prob function)"""
    p = np.concatenate([np.full(n, v, np.float32) for n, v in runs])
    return np.zeros(len(p) * W, np.float32), (lambda audio: p)


def test_two_utterances_with_a_long_pause():
    audio, fn = _probs((10, 0.0), (50, 0.9), (100, 0.0), (30, 0.9))
    got = vad.get_speech_timestamps(audio, vad.VadOptions(), fn)
    # speech 1: frames 10..59 -> [5120, 30720); the pause reaches 2 s at frame 123 -> closed at its beginning.
    # speech 2: frames 160..189, open at the end of the audio -> [81920, 97280).  Pads of 6400 samples, gap 51200 >= 12800.
    assert got == [{"start": 0, "end": 37120}, {"start": 75520, "end": 97280}]
    assert len(vad.collect_chunks(audio, got)) == 37120 + 21760


def test_short_pause_does_not_split_and_blips_are_dropped():
    audio, fn = _probs((5, 0.0), (40, 0.8), (30, 0.1), (40, 0.8), (70, 0.0))
    got = vad.get_speech_timestamps(audio, vad.VadOptions(speech_pad_ms=0), fn)
    assert got == [{"start": 5 * W, "end": 115 * W}]                      # 30 frames = 0.96 s < 2 s: one chunk
    audio, fn = _probs((5, 0.0), (5, 0.9), (80, 0.0), (40, 0.9), (70, 0.0))
    got = vad.get_speech_timestamps(audio, vad.VadOptions(min_speech_duration_ms=250, speech_pad_ms=0), fn)
    assert got == [{"start": 90 * W, "end": 130 * W}]                    # the 160-ms blip is below 250 ms
    # hysteresis: probabilities between neg_threshold (0.35) and threshold (0.5) neither open nor close a chunk
    audio, fn = _probs((5, 0.4), (20, 0.9), (80, 0.4), (5, 0.0))
    got = vad.get_speech_timestamps(audio, vad.VadOptions(speech_pad_ms=0), fn)
    assert got == [{"start": 5 * W, "end": 110 * W}]


def test_close_chunks_share_the_gap_and_long_speech_is_cut_at_a_pause():
    audio, fn = _probs((20, 0.9), (20, 0.0), (20, 0.9))
    got = vad.get_speech_timestamps(audio, vad.VadOptions(min_silence_duration_ms=500), fn)
    # closes after 500 ms (frame 36); gap 10240 samples < 2 * 6400 -> each side takes half
    assert got == [{"start": 0, "end": 20 * W + 5120}, {"start": 40 * W - 5120, "end": 60 * W}]
    audio, fn = _probs((40, 0.9), (5, 0.1), (55, 0.9))
    got = vad.get_speech_timestamps(audio, vad.VadOptions(max_speech_duration_s=2.0, speech_pad_ms=0), fn)
    # 2 s = 61.5 frames is exceeded at frame 62; the last pause >= 98 ms began at frame 40, speech resumed at 45
    assert got == [{"start": 0, "end": 40 * W}, {"start": 45 * W, "end": 100 * W}]
    assert vad.get_speech_timestamps(*_probs((50, 0.1))[:1], vad.VadOptions(), _probs((50, 0.1))[1]) == []


def test_time_restoration():
    chunks = [{"start": 0, "end": 37120}, {"start": 75520, "end": 97280}]
    m = vad.SpeechTimestampsMap(chunks)
    assert m.chunk_end_sample == [37120, 58880] and m.total_silence_before == [0.0, 2.4]
    assert m.get_original_time(1.0) == 1.0 and m.get_original_time(2.5) == 4.9 and m.get_original_time(3.68) == 6.08

    class Wd(NamedTuple):
        start: float
        end: float
        word: str
        probability: float

    class Seg(NamedTuple):
        start: float
        end: float
        text: str
        words: Optional[List[Wd]]

    segs = [Seg(0.5, 2.0, "a", None), Seg(2.4, 3.0, "b", [Wd(2.3, 2.5, "x", 0.9), Wd(2.5, 3.0, "y", 0.8)])]
    out = list(vad.restore_speech_timestamps(iter(segs), chunks))
    assert (out[0].start, out[0].end) == (0.5, 2.0)
    # word x straddles the cut (2.32 s): its middle (2.4) lies in chunk 2 -> both ends move by 2.4 s
    assert [(w.start, w.end) for w in out[1].words] == [(4.7, 4.9), (4.9, 5.4)] and (out[1].start, out[1].end) == (4.7, 5.4)


def test_energy_stand_in_finds_the_burst():
    clip = synth.burst_clip(0)                      # 3 s of noise, then zeros
    p = vad.energy_speech_prob(clip)
    n_loud = 3 * 16000 // W
    assert p[: n_loud - 1].min() > 0.9 and p[n_loud + 1:].max() < 0.1
    got = vad.get_speech_timestamps(clip, vad.VadOptions())
    assert len(got) == 1 and got[0]["start"] == 0 and abs(got[0]["end"] - (3 * 16000 + 6400)) <= W


def test_silero_shaped_network_is_driven_frame_by_frame_with_context_and_state():
    """The Silero network itself cannot be shipped (no weights offline): what is pinned is that a Silero-SHAPED stateful
    callable is driven correctly - 512-sample frames prefixed by 64 context samples, float32 [1, 576], the recurrent state
    [2, 1, 128] threaded through, the tail zero-padded - and that its probabilities reach the chunking rule."""
    from taiwan_tongues_asr_ce_amd import vad
    seen = []

    def step(x, h):
        assert x.shape == (1, 576) and x.dtype == np.float32 and h.shape == (2, 1, 128) and h.dtype == np.float32
        seen.append((x.copy(), float(h[0, 0, 0])))
        level = float(np.abs(x[0, 64:]).mean())
        return (0.9 if level > 0.05 else 0.02), h + 1.0            # the "state" counts the calls

    audio = np.zeros(16000 * 3 + 100, dtype=np.float32)
    audio[16000:32000] = 0.3 * np.sin(np.arange(16000) * 0.2)
    fn = vad.silero_speech_prob_fn(step)
    probs = fn(audio)
    n = int(np.ceil(len(audio) / 512))
    assert probs.shape == (n,) and len(seen) == n
    assert [h for _, h in seen] == [float(i) for i in range(n)]                    # state threaded call to call, zeros first
    assert not seen[0][0][0, :64].any()                                            # no context before the first frame
    for i in range(1, n):
        np.testing.assert_array_equal(seen[i][0][0, :64], seen[i - 1][0][0, -64:])  # context = tail of the previous frame
    assert not seen[-1][0][0, 64 + (len(audio) - (n - 1) * 512):].any()            # zero-padded tail
    chunks = vad.get_speech_timestamps(audio, vad.VadOptions(min_silence_duration_ms=300, speech_pad_ms=0), fn)
    assert len(chunks) == 1 and abs(chunks[0]["start"] - 16000) <= 512 and abs(chunks[0]["end"] - 32000) <= 1024


def test_silero_adapter_contract_edge_cases():
    """More of the documented contract of vad.silero_speech_prob_fn (vad.py:52-75), recorded by a fake `step` (VERDICT round 4,
    next #8): the frames tile the recording exactly (nothing dropped, nothing doubled), a recording whose length is a multiple of
    512 gets no extra frame, every recording starts from a ZERO state and ZERO context (nothing leaks from the previous call),
    the state the network returns is the very object handed to the next call, an empty recording makes no call, and the
    context length / state shape are the caller's (Silero v4 graphs: no context, [2, 1, 64] state)."""
    from taiwan_tongues_asr_ce_amd import vad
    calls = []

    def step(x, h):
        calls.append((x.copy(), h))
        return 0.25 + 0.5 * (len(calls) % 2), h + 1.0
    fn = vad.silero_speech_prob_fn(step)
    rng = np.random.default_rng(0)
    for n_samples in (512 * 7, 512 * 7 + 1, 511, 1):
        calls.clear()
        audio = rng.standard_normal(n_samples).astype(np.float32)
        probs = fn(audio)
        n = -(-n_samples // 512)
        assert probs.shape == (n,) and probs.dtype == np.float32 and len(calls) == n
        body = np.concatenate([x[0, 64:] for x, _ in calls])
        np.testing.assert_array_equal(body[:n_samples], audio)                     # the frames tile the recording
        assert not body[n_samples:].any()                                          # and only zeros pad the last one
        assert not calls[0][0][0, :64].any() and not calls[0][1].any()             # fresh context + state per recording
        for i in range(1, n):
            assert float(calls[i][1][0, 0, 0]) == float(i)                         # what step returned is what step gets
        np.testing.assert_allclose(probs, [0.25 + 0.5 * ((i + 1) % 2) for i in range(n)])
    calls.clear()
    assert fn(np.zeros(0, np.float32)).shape == (0,) and not calls
    v4 = vad.silero_speech_prob_fn(lambda x, h: (calls.append((x.shape, h.shape)) or 0.0, h), context=0, state_shape=(2, 1, 64))
    v4(np.ones(1000, np.float32))
    assert calls == [((1, 512), (2, 1, 64))] * 2
