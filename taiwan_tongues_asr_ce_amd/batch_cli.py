"""Folder transcription tool: the caller on the near side of the hot path (SURVEY §3 call stack 1, §8f N3).

Mirrors the observable contract of the reference's batch entry point (asr_core.py:109-369):

* every `*.wav|mp3|flac|m4a|aac` (either case) directly inside the folder is one unit of work (asr_core.py:118-131);
* each is transcribed with `language="zh", word_timestamps=False, vad_filter=True, beam_size=5,
  condition_on_previous_text=True, initial_prompt=""` (asr_core.py:159-167) and ALL lazy segments are consumed;
* the normalised text goes to `<stem>_asr.txt` beside the audio (asr_core.py:181-187); when transcription raises, that
  file holds the file name and the error instead and the run continues (asr_core.py:244-255);
* a transcript named `<stem>.txt`, `<stem>_transcript.txt`, `_original`, `_reference` or `_ground_truth`
  (asr_core.py:86-106) is scored with `scoring.compare_texts`;
* `asr_comparison_results.json` in the working directory holds `summary` + `detailed_results` with the reference's
  keys (asr_core.py:258-334).

Differences: by default several files advance in lock step through one engine pass per window round
(`WhisperModel.transcribe_many`: same per-file algorithm and outputs as one by one, batch throughput; `--group-files 1`
restores strictly sequential processing); files are processed in sorted order (the reference iterates a `set`); with several GPUs
(`torchrun --nproc-per-node N -m taiwan_tongues_asr_ce_amd.batch_cli folder`) files are sharded by rank — never by
window, windows of one file are sequentially dependent — and rank 0 merges the per-rank results; only RIFF/WAV is
decodable without librosa/PyAV, other containers are reported as per-file errors.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
from typing import Callable, Dict, List, Optional

from . import scoring

AUDIO_PATTERNS = ("*.wav", "*.mp3", "*.flac", "*.m4a", "*.aac")
TRANSCRIPT_SUFFIXES = ("", "_transcript", "_original", "_reference", "_ground_truth")
TRANSCRIBE_KWARGS = dict(language="zh", word_timestamps=False, vad_filter=True, beam_size=5,
                         condition_on_previous_text=True, initial_prompt="")


def list_audio_files(folder: str) -> List[str]:
    found = set()
    for pat in AUDIO_PATTERNS:
        found.update(glob.glob(os.path.join(folder, pat)))
        found.update(glob.glob(os.path.join(folder, pat.upper())))
    return sorted(found)


def find_original_transcript(audio_file: str) -> Optional[str]:
    stem = os.path.splitext(audio_file)[0]
    for suffix in TRANSCRIPT_SUFFIXES:
        cand = f"{stem}{suffix}.txt"
        if os.path.exists(cand):
            return cand
    return None


def _load_audio(path: str):
    from .model import decode_audio
    return decode_audio(path)


def transcribe_file(model, audio_file: str, load_audio: Callable = _load_audio, log: Callable = print, segments=None) -> Dict:
    """One unit of work → one entry of `detailed_results`.  `segments`: already transcribed (group mode), else runs it."""
    name = os.path.basename(audio_file)
    out_path = os.path.splitext(audio_file)[0] + "_asr.txt"
    try:
        if isinstance(segments, Exception):
            raise segments
        if segments is None:
            segments, _info = model.transcribe(load_audio(audio_file), **TRANSCRIBE_KWARGS)
        text = "".join(seg.text for seg in segments)            # consumes the lazy generator: this is where it runs
        processed = scoring.normalise_transcript(text)
        with open(out_path, "w", encoding="utf-8") as f:
            f.write(processed)
        log(f"{name}: {processed}")
        entry = {"audio_file": name, "asr_result": processed, "original_transcript": None, "cer_result": None,
                 "has_original_transcript": False}
        ref_path = find_original_transcript(audio_file)
        if ref_path is not None:
            try:
                with open(ref_path, "r", encoding="utf-8") as f:
                    original = f.read().strip()
                entry["original_transcript"] = original
                entry["has_original_transcript"] = True
                res = scoring.compare_texts(original, processed)
                if res is not None:
                    entry["cer_result"] = res.as_dict()
                    log(f"{name}: CER {res.cer_rate:.4f} ({res.substitutions_count} sub, {res.deletions_count} del, "
                        f"{res.insertions_count} ins)")
            except Exception as e:  # unreadable transcript: keep the ASR result, as the reference does
                log(f"{name}: transcript unreadable: {e}")
        return entry
    except Exception as e:
        with open(out_path, "w", encoding="utf-8") as f:
            f.write(f"檔案名稱: {name}\n錯誤: {e}\n")
        log(f"{name}: error: {e}")
        return {"audio_file": name, "asr_result": None, "original_transcript": None, "cer_result": None,
                "has_original_transcript": False, "error": str(e)}


def summarise(results: List[Dict]) -> Dict:
    scored = [r["cer_result"] for r in results if r.get("cer_result") is not None]
    n = len(scored)
    return {
        "summary": {
            "total_files": len(results),
            "files_with_transcript": sum(1 for r in results if r.get("has_original_transcript", False)),
            "files_with_cer": n,
            "average_cer": sum(c["cer_rate"] for c in scored) / n if n else 0,
            "average_correct_rate": sum(c["correct_rate"] for c in scored) / n if n else 0,
            "total_substitutions": sum(c["substitutions_count"] for c in scored),
            "total_deletions": sum(c["deletions_count"] for c in scored),
            "total_insertions": sum(c["insertions_count"] for c in scored),
        },
        "detailed_results": results,
    }


def process_audio_folder(folder_path: str, model=None, model_path: str = "models", device: str = "cuda",
                         device_index: int = 0, compute_type: str = "float16", max_batch: int = 120, output_json: Optional[str] = None, rank: int = 0,
                         world: int = 1, load_audio: Callable = _load_audio, log: Callable = print,
                         group_files: int = 0, pipeline_depth: int = 0) -> Optional[Dict]:
    files = list_audio_files(folder_path)
    if not files:
        log(f"no audio files in {folder_path}")
        return None
    if model is None:
        from .model import WhisperModel
        model = WhisperModel(model_path, device=device, device_index=device_index, compute_type=compute_type,
                             max_batch=max_batch, pipeline_depth=max(1, pipeline_depth))
    mine = files[rank::world]                                   # shard by file
    results = []
    many = getattr(model, "transcribe_many", None)
    group = group_files if group_files > 0 else max(1, getattr(model, "max_batch", 1) // TRANSCRIBE_KWARGS["beam_size"])
    if many is None or group < 2 or getattr(model, "vad_speech_prob_fn", None) is not None:
        results = [transcribe_file(model, f, load_audio, log) for f in mine]
    else:
        # MI355X-first: `group` files advance in lock step through one engine pass per window round; every file keeps the
        # sequential algorithm (own seek / prompt / fallback), so the outputs equal the one-by-one run
        kw = {k: v for k, v in TRANSCRIBE_KWARGS.items() if k != "vad_filter"}   # no VAD source configured: all-speech
        depth0 = max(1, int(pipeline_depth or getattr(model, "pipeline_depth", 1)))
        if group_files <= 0 and depth0 > 1 and len(mine) < group * depth0:
            group = max(1, -(-len(mine) // depth0))            # few files: one group per context rather than one big group and an idle lane
        parts = [mine[g:g + group] for g in range(0, len(mine), group)]
        # round 6: with pipeline_depth > 1 the groups are transcribed by that many engine contexts of the model at once (one shared
        # copy of the weights; one group's log-mel / encoder under another's decode) - same groups, same results, in file order.
        # Audio is loaded a few groups ahead only (a folder may hold many hours).
        depth = max(1, int(pipeline_depth or getattr(model, "pipeline_depth", 1)))
        groups_fn = getattr(model, "transcribe_groups", None) if depth > 1 else None
        span = 2 * depth if groups_fn is not None else 1
        for g0 in range(0, len(parts), span):
            chunk = parts[g0:g0 + span]
            loaded_of, audios_of = [], []
            for part in chunk:
                audios, loaded = [], []
                for f in part:
                    try:
                        audios.append(load_audio(f))
                        loaded.append(None)
                    except Exception as e:
                        loaded.append(e)
                audios_of.append(audios)
                loaded_of.append(loaded)
            done_of = None
            if groups_fn is not None and len(chunk) > 1:
                try:
                    done_of = groups_fn(audios_of, pipeline_depth=depth, **kw)
                except Exception as e:
                    log(f"pipelined transcription failed ({e}); retrying group by group")
            for gi, part in enumerate(chunk):
                try:
                    if done_of is not None:
                        done = iter(done_of[gi])
                    else:
                        done = iter(many(audios_of[gi], **kw)) if audios_of[gi] else iter(())
                    segs = [err if err is not None else next(done)[0] for err in loaded_of[gi]]
                except Exception as e:  # engine-level failure of the group: fall back to one by one
                    log(f"group transcription failed ({e}); retrying file by file")
                    segs = [None] * len(part)
                results.extend(transcribe_file(model, f, load_audio, log, segments=sg) for f, sg in zip(part, segs))
    if world > 1:
        import torch.distributed as dist
        gathered = [None] * world
        dist.all_gather_object(gathered, results)
        order = {os.path.basename(f): i for i, f in enumerate(files)}
        results = sorted((r for part in gathered for r in part), key=lambda r: order[r["audio_file"]])
    final = summarise(results)
    if rank == 0:
        path = output_json or os.path.join(os.getcwd(), "asr_comparison_results.json")
        with open(path, "w", encoding="utf-8") as f:
            json.dump(final, f, ensure_ascii=False, indent=2)
        s = final["summary"]
        log(f"{s['total_files']} files, {s['files_with_cer']} scored, average CER {s['average_cer']:.4f}; wrote {path}")
    return final


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description="Transcribe every audio file of a folder on MI355X and score against transcripts")
    ap.add_argument("folder")
    ap.add_argument("--output", default="transcription_results.txt", help="accepted for compatibility; unused")
    ap.add_argument("--model", default="models", help="HF-format Whisper directory, or synthetic:<preset>")
    ap.add_argument("--compute-type", default="float16")
    ap.add_argument("--group-files", type=int, default=0,
                    help="files transcribed in lock step per engine pass (0 = as many as fit: max_batch // beam; 1 = one by one)")
    ap.add_argument("--max-batch", type=int, default=120,
                    help="decode rows of an engine context (files in a group x beam 5; the weight-streaming decode GEMMs carry up to 128 "
                         "rows).  Measured on 48 x 60 s files with two contexts: 30 rows 893, 60 rows 1 064, 120 rows 1 212 audio-s/s; a "
                         "context holds 11 / 23 / 44 GB of KV caches and workspaces at 30 / 60 / 120 rows of large-v3")
    ap.add_argument("--pipeline-depth", type=int, default=2,
                    help="groups of files in flight per GPU (engine contexts sharing one copy of the weights; 1 = the reference's "
                         "serial loop, asr_core.py:151); results do not depend on it")
    args = ap.parse_args(argv)
    if not os.path.exists(args.folder):
        print(f"folder does not exist: {args.folder}")
        return 1
    rank, world, local = 0, 1, 0
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from .dist import init_process_group
        rank, world, local = init_process_group()
    process_audio_folder(args.folder, model_path=args.model, device="cuda", device_index=local,
                         compute_type=args.compute_type, rank=rank, world=world, group_files=args.group_files,
                         max_batch=args.max_batch, pipeline_depth=args.pipeline_depth)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
