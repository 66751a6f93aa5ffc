"""Streaming workload of SURVEY §8(d) config C5 on ONE GPU: concurrent WebSocket-like clients, each delivering a
3-s utterance (the reference transcribes when > 2.1 s have accumulated: buffering_strategies.py:118-126) every
`period` seconds; all of them go through `BatchedWhisperASR.transcribe(client)` — the ASRInterface entry point the
service calls (faster_whisper_asr.py:170-172 counterpart) — and are coalesced into batched engine passes.

Reports per-utterance latency (submit → result dict) p50/p99 and aggregate audio-seconds per wall-second.  Synthetic
weights never emit EOT, so `--new-tokens` caps the decode length (16 ≈ a 3-s Mandarin utterance).

    python tools/stream_bench.py --model large-v3 --streams 8 --rounds 6 --beam 5
    python tools/stream_bench.py --model large-v3 --streams 8 --rounds 6 --beam 5 --audio-ctx auto   # N2 short window
"""
import argparse
import asyncio
import json
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


async def run(args):
    from taiwan_tongues_asr_ce_amd import synth
    from taiwan_tongues_asr_ce_amd.streaming import BatchedWhisperASR
    beam = args.beam
    max_clips = max(1, 32 // beam) if args.max_clips == 0 else args.max_clips
    asr = BatchedWhisperASR(max_clips=max_clips, max_wait_ms=args.max_wait_ms, beam_size=beam,
                            audio_ctx=None if args.audio_ctx == "none" else (args.audio_ctx if args.audio_ctx == "auto" else int(args.audio_ctx)),
                            max_new_tokens=args.new_tokens, model_size=f"synthetic:{args.model}", compute_type="bfloat16")
    utter = [np.clip(synth.noise_clip(1000 + i, int(args.utterance_s * 16000)) * 32768.0, -32768, 32767).astype("<i2").tobytes()
             for i in range(args.streams)]
    lat = []

    async def client(i):
        c = types.SimpleNamespace(scratch_buffer=utter[i], last_start_time=0, client_id=i)
        for r in range(args.rounds + 1):
            t = time.perf_counter()
            res = await asr.transcribe(c)
            if r > 0:                                   # round 0 = warm-up (graph capture, first-touch)
                lat.append(time.perf_counter() - t)
            assert res is None or "text" in res
            if args.period > 0:
                await asyncio.sleep(max(0.0, args.period - (time.perf_counter() - t)))

    # warm-up round is inside client(); time rounds 1..n as a whole for the aggregate rate
    t0 = time.perf_counter()
    await asyncio.gather(*(client(i) for i in range(args.streams)))
    wall = time.perf_counter() - t0
    await asr.aclose()
    lat_ms = np.asarray(lat) * 1e3
    sizes = asr.batches_run
    out = {
        "workload": f"C5 streaming: {args.streams} concurrent streams x {args.rounds} utterances of {args.utterance_s} s, "
                    f"whisper-{args.model} geometry bf16, beam {beam}, <= {args.new_tokens} tokens, audio_ctx={args.audio_ctx}, "
                    f"period {args.period} s, 1 GPU",
        "latency_ms": {"p50": round(float(np.percentile(lat_ms, 50)), 1), "p99": round(float(np.percentile(lat_ms, 99)), 1),
                       "max": round(float(lat_ms.max()), 1)},
        "audio_s_per_s": round(len(lat) * args.utterance_s / (wall * args.rounds / (args.rounds + 1)), 1),
        "engine_passes": len(sizes), "mean_clips_per_pass": round(float(np.mean(sizes)), 2),
        "last_pass_phase_ms": {k: round(v, 2) for k, v in asr.asr_pipeline.engine.phase_ms().items()},
    }
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--streams", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--beam", type=int, default=5)
    ap.add_argument("--new-tokens", type=int, default=16)
    ap.add_argument("--utterance-s", type=float, default=3.0)
    ap.add_argument("--period", type=float, default=0.0, help="seconds between a stream's utterances (0 = back to back)")
    ap.add_argument("--audio-ctx", default="none", help="none | auto | <positions>")
    ap.add_argument("--max-clips", type=int, default=0)
    ap.add_argument("--max-wait-ms", type=float, default=5.0)
    asyncio.run(run(ap.parse_args()))


if __name__ == "__main__":
    main()
