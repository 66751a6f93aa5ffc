"""SURVEY §8(d) config C1 ("plumbing"): tiny geometry, B = 1, one 30-s synthetic clip and one 11-s clip (padded),
greedy, through the asr_core-equivalent folder tool — on CPU, with the ORACLE standing in for the engine behind the
`WhisperModel.transcribe` contract (tests may use the oracle; the product path never does).  It ties the host-side
pieces together without a GPU: wav decode → model contract → segment text → normalisation → `_asr.txt` → CER JSON."""
import json
import wave

import numpy as np
import torch

from oracle import whisper_ref as R
from taiwan_tongues_asr_ce_amd import batch_cli, scoring, synth
from taiwan_tongues_asr_ce_amd.config import PRESETS, SpecialTokens
from taiwan_tongues_asr_ce_amd.model import decode_audio
from taiwan_tongues_asr_ce_amd.tokenizer import ByteStubTokenizer

torch.set_grad_enabled(False)


class _Seg:
    def __init__(self, text):
        self.text = text


class OracleWhisper:
    """The reference call-site contract (asr_core.py:159-167) over the CPU oracle: one 30-s window, greedy."""

    def __init__(self, name="tiny", max_new=12):
        self.pd = PRESETS[name]
        self.dims = R.Dims(**self.pd.as_dict())
        self.W = R.to_torch(synth.state_dict(self.pd))
        self.st = SpecialTokens.for_vocab(self.pd.vocab)
        self.tok = ByteStubTokenizer(self.pd.vocab, self.st.eot)
        self.max_new = max_new
        self.seen = []

    def transcribe(self, audio, *, language, word_timestamps, vad_filter, beam_size, condition_on_previous_text,
                   initial_prompt):
        from taiwan_tongues_asr_ce_amd.engine import default_suppress
        st = self.st
        rules = R.Rules(eot=st.eot, no_timestamps=st.no_timestamps, timestamp_begin=st.timestamp_begin,
                        suppress=default_suppress(st, self.pd.vocab), begin_suppress=[220, st.eot], timestamps=True)
        res = R.transcribe_tokens([audio], self.W, self.dims, [st.sot, st.lang_zh, st.transcribe], rules, self.max_new)
        toks = [t for t in res.tokens[0] if t < st.eot]
        self.seen.append((len(audio), toks))
        return iter([_Seg(self.tok.decode(toks))]), None


def _write_wav(path, pcm):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(np.clip(pcm * 32768.0, -32768, 32767).astype("<i2").tobytes())


def test_c1_folder_tool_over_the_cpu_restatement(tmp_path):
    folder = tmp_path / "clips"
    folder.mkdir()
    _write_wav(folder / "full30s.wav", synth.noise_clip(0))                    # exactly one window
    _write_wav(folder / "warm11s.wav", synth.tonal_clip(1)[: 11 * 16000])      # shorter: zero-padded to the window
    model = OracleWhisper()
    # transcript for one of them = what the model will say, so its CER is exactly 0
    pcm = decode_audio(str(folder / "warm11s.wav"))
    say, _ = model.transcribe(pcm, language="zh", word_timestamps=False, vad_filter=True, beam_size=5,
                              condition_on_previous_text=True, initial_prompt="")
    text = next(say).text
    (folder / "warm11s_reference.txt").write_text(text, encoding="utf-8")
    model.seen.clear()
    final = batch_cli.process_audio_folder(str(folder), model=model, output_json=str(tmp_path / "out.json"), log=lambda *_: None)
    assert [n for n, _ in model.seen] == [480000, 176000]                      # both files reached the model, full length
    assert all(len(t) > 0 for _, t in model.seen)
    a, b = final["detailed_results"]
    assert a["audio_file"] == "full30s.wav" and b["audio_file"] == "warm11s.wav"
    for entry, (_, toks) in zip((a, b), model.seen):
        want = scoring.normalise_transcript(model.tok.decode(toks))
        assert entry["asr_result"] == want
        assert (folder / (entry["audio_file"][:-4] + "_asr.txt")).read_text(encoding="utf-8") == want
    assert a["has_original_transcript"] is False and b["has_original_transcript"] is True
    if scoring.clean_for_scoring(text):                                        # private-use glyphs are scored away
        assert b["cer_result"]["cer_rate"] == 0
    assert json.load(open(tmp_path / "out.json", encoding="utf-8"))["summary"]["total_files"] == 2
