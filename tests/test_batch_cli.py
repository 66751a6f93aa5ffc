"""Folder tool (asr_core.py counterpart) on CPU with a model double — same approach as the reference's own
test double (api/tests/test_file_asr.py:40-53): positional audio, the six kwargs, returns (segments, info)."""
import json
import os
import wave

import numpy as np
import pytest

from taiwan_tongues_asr_ce_amd import batch_cli


class _Seg:
    def __init__(self, text):
        self.text = text


class _Model:
    def __init__(self, texts):
        self.texts, self.calls = texts, []

    def transcribe(self, audio, *, language, word_timestamps, vad_filter, beam_size, condition_on_previous_text, initial_prompt):
        self.calls.append(dict(n=len(audio), language=language, word_timestamps=word_timestamps, vad_filter=vad_filter,
                               beam_size=beam_size, cond=condition_on_previous_text, prompt=initial_prompt))
        text = self.texts[len(self.calls) - 1]
        if isinstance(text, Exception):
            raise text
        return (_Seg(t) for t in text), object()


def _wav(path, n=1600):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes((np.zeros(n, dtype="<i2")).tobytes())


def test_list_and_transcript_discovery(tmp_path):
    for n in ("b.wav", "a.WAV", "c.mp3", "d.txt", "e.ogg"):
        (tmp_path / n).write_bytes(b"")
    (tmp_path / "sub").mkdir(); (tmp_path / "sub" / "x.wav").write_bytes(b"")
    assert [os.path.basename(f) for f in batch_cli.list_audio_files(str(tmp_path))] == ["a.WAV", "b.wav", "c.mp3"]
    assert batch_cli.find_original_transcript(str(tmp_path / "b.wav")) is None
    (tmp_path / "b_reference.txt").write_text("x", encoding="utf-8")
    assert batch_cli.find_original_transcript(str(tmp_path / "b.wav")).endswith("b_reference.txt")
    (tmp_path / "b.txt").write_text("x", encoding="utf-8")
    assert batch_cli.find_original_transcript(str(tmp_path / "b.wav")).endswith("b.txt")     # plain name wins


def test_process_audio_folder_contract(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    folder = tmp_path / "audio"; folder.mkdir()
    for n in ("a.wav", "b.wav", "c.wav"):
        _wav(folder / n)
    (folder / "a.txt").write_text("今天天氣很好，我們去公園散步。\n", encoding="utf-8")
    model = _Model([["今天天氣很好!，", "我去公園散步。"], RuntimeError("boom"), ["百分之十五 ＯＫ"]])
    final = batch_cli.process_audio_folder(str(folder), model=model, log=lambda *_: None)
    assert all(c == dict(n=1600, language="zh", word_timestamps=False, vad_filter=True, beam_size=5, cond=True, prompt="")
               for c in model.calls) and len(model.calls) == 3
    assert (folder / "a_asr.txt").read_text(encoding="utf-8") == "今天天氣很好我去公園散步"
    assert (folder / "b_asr.txt").read_text(encoding="utf-8") == "檔案名稱: b.wav\n錯誤: boom\n"
    assert (folder / "c_asr.txt").read_text(encoding="utf-8") == "15% ok"
    on_disk = json.load(open(tmp_path / "asr_comparison_results.json", encoding="utf-8"))
    assert on_disk == final
    assert list(final) == ["summary", "detailed_results"]
    assert list(final["summary"]) == ["total_files", "files_with_transcript", "files_with_cer", "average_cer",
                                      "average_correct_rate", "total_substitutions", "total_deletions", "total_insertions"]
    s = final["summary"]
    assert (s["total_files"], s["files_with_transcript"], s["files_with_cer"]) == (3, 1, 1)
    a, b, c = final["detailed_results"]
    assert a["has_original_transcript"] and a["cer_result"]["deletions_count"] == 1 and a["cer_result"]["total_chars"] == 13
    assert s["average_cer"] == pytest.approx(1 / 13) and s["total_deletions"] == 1
    assert b == {"audio_file": "b.wav", "asr_result": None, "original_transcript": None, "cer_result": None,
                 "has_original_transcript": False, "error": "boom"}
    assert c["asr_result"] == "15% ok" and c["cer_result"] is None and "error" not in c


def test_empty_folder_and_missing_folder(tmp_path, capsys):
    assert batch_cli.process_audio_folder(str(tmp_path), model=_Model([]), log=lambda *_: None) is None
    assert batch_cli.main([str(tmp_path / "nope")]) == 1


class _GroupModel(_Model):
    """Double with the group entry point: records how files were grouped."""
    max_batch = 10      # // beam 5 -> 2 files per engine pass

    def __init__(self, texts, fail_groups=()):
        super().__init__(texts)
        self.groups, self.fail_groups = [], set(fail_groups)

    def transcribe_many(self, audios, *, language, word_timestamps, beam_size, condition_on_previous_text, initial_prompt):
        self.groups.append([len(a) for a in audios])
        if len(self.groups) - 1 in self.fail_groups:
            raise RuntimeError("engine fault")
        out = []
        for a in audios:
            out.append(([_Seg(f"長{len(a)}")], object()))
        return out


def test_group_mode_batches_files_and_degrades_per_file(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    folder = tmp_path / "audio"; folder.mkdir()
    for i, n in enumerate((100, 200, 300, 400, 500)):
        _wav(folder / f"f{i}.wav", n)
    (folder / "f1.wav").write_bytes(b"not a wav")                       # unreadable file inside a group
    model = _GroupModel([["fallback-a"], ["fallback-b"]], fail_groups={2})
    final = batch_cli.process_audio_folder(str(folder), model=model, log=lambda *_: None)
    assert model.groups == [[100], [300, 400], [500]]                   # f0 (+ broken f1), f2+f3, f4 (group 2 fails)
    r = final["detailed_results"]
    assert [x["asr_result"] for x in r] == ["長100", None, "長300", "長400", "fallback-a"]
    assert "error" in r[1] and (folder / "f1_asr.txt").read_text(encoding="utf-8").startswith("檔案名稱: f1.wav")
    assert len(model.calls) == 1 and model.calls[0]["n"] == 500         # only the failed group was retried one by one
    # --group-files 1 restores strictly sequential processing
    model2 = _GroupModel([["x"]] * 5)
    batch_cli.process_audio_folder(str(folder), model=model2, log=lambda *_: None, group_files=1)
    assert model2.groups == [] and len(model2.calls) == 4               # the unreadable file never reaches the model


class _PipelinedModel(_GroupModel):
    """Double with the round-6 entry point: groups in flight on several engine contexts."""
    pipeline_depth = 2

    def __init__(self, texts, fail_pipeline=False):
        super().__init__(texts)
        self.pipelined, self.fail_pipeline = [], fail_pipeline

    def transcribe_groups(self, groups, pipeline_depth=None, **kw):
        self.pipelined.append((pipeline_depth, [[len(a) for a in g] for g in groups]))
        if self.fail_pipeline:
            raise RuntimeError("second context could not be created")
        return [self.transcribe_many(g, **kw) for g in groups]


def test_pipelined_folder_keeps_the_groups_and_the_results(tmp_path, monkeypatch):
    """pipeline_depth = 2: the SAME groups of files as the serial run - handed to the model a few
    groups at a time (2 x depth) so that a folder of many hours is not decoded into memory at once - the same results in file
    order; a failure of the pipelined call degrades to group by group."""
    monkeypatch.chdir(tmp_path)
    folder = tmp_path / "audio"; folder.mkdir()
    sizes = [100 * (i + 1) for i in range(11)]
    for i, n in enumerate(sizes):
        _wav(folder / f"f{i:02d}.wav", n)
    serial = _GroupModel([])
    want = batch_cli.process_audio_folder(str(folder), model=serial, log=lambda *_: None)
    assert serial.groups == [[100, 200], [300, 400], [500, 600], [700, 800], [900, 1000], [1100]]
    piped = _PipelinedModel([])
    got = batch_cli.process_audio_folder(str(folder), model=piped, log=lambda *_: None)
    assert piped.groups == serial.groups                                        # identical grouping = identical engine inputs
    assert [d for d, _ in piped.pipelined] == [2, 2]                            # two calls of 2 x depth = 4 groups, then the rest
    assert piped.pipelined[0][1] == serial.groups[:4] and piped.pipelined[1][1] == serial.groups[4:]
    assert got["detailed_results"] == want["detailed_results"]
    # explicit depth 1 never touches the pipelined entry point; a failing pipeline falls back without losing a file
    one = _PipelinedModel([])
    batch_cli.process_audio_folder(str(folder), model=one, log=lambda *_: None, pipeline_depth=1)
    assert one.pipelined == [] and one.groups == serial.groups
    broken = _PipelinedModel([], fail_pipeline=True)
    logs = []
    res = batch_cli.process_audio_folder(str(folder), model=broken, log=logs.append)
    assert res["detailed_results"] == want["detailed_results"] and any("pipelined transcription failed" in str(m) for m in logs)


def test_few_files_are_split_over_the_contexts(tmp_path, monkeypatch):
    """Fewer files than one full group per context (the default context carries 24 files): one group per context instead of one
    big group beside an idle lane; --group-files keeps whatever the caller asked for."""
    monkeypatch.chdir(tmp_path)
    folder = tmp_path / "audio"; folder.mkdir()
    for i in range(5):
        _wav(folder / f"f{i}.wav", 100 * (i + 1))

    class Wide(_PipelinedModel):
        max_batch = 120                                                          # 24 files per engine pass
    m = Wide([])
    batch_cli.process_audio_folder(str(folder), model=m, log=lambda *_: None)
    assert m.pipelined == [(2, [[100, 200, 300], [400, 500]])]
    m1 = Wide([])
    batch_cli.process_audio_folder(str(folder), model=m1, log=lambda *_: None, pipeline_depth=1)
    assert m1.groups == [[100, 200, 300, 400, 500]] and m1.pipelined == []
    m2 = Wide([])
    batch_cli.process_audio_folder(str(folder), model=m2, log=lambda *_: None, group_files=4)
    assert m2.groups == [[100, 200, 300, 400], [500]]
