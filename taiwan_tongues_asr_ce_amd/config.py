"""Whisper geometry presets and decode options for the MI355X hot path.

The reference never states the model geometry itself: it loads whatever CTranslate2 directory sits in
``models/`` (asr_core.py:141) or the size string ``large-v3-turbo`` (faster_whisper_asr.py:21).  The
numbers below are the published Whisper shapes (SURVEY.md section 8 notation) plus a ``micro`` shape that
keeps head_dim = 64 but is small enough for second-scale CPU oracle runs.
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import Dict, List, Optional, Tuple

SAMPLE_RATE = 16000
N_FFT = 400
HOP = 160
CHUNK_SECONDS = 30
N_SAMPLES = SAMPLE_RATE * CHUNK_SECONDS  # 480000
N_FRAMES = N_SAMPLES // HOP  # 3000

COMPUTE_F32 = 0
COMPUTE_BF16 = 1
COMPUTE_F16 = 2   # IEEE fp16 storage, f32 accumulate: what compute_type="float16" (asr_core.py:141, api/config.py:12) means


@dataclass(frozen=True)
class WhisperDims:
    name: str
    n_mels: int
    n_audio_ctx: int  # encoder positions (max_source_positions); mel frames = 2 * n_audio_ctx
    d_model: int
    n_heads: int
    ffn_dim: int
    enc_layers: int
    dec_layers: int
    vocab: int
    n_text_ctx: int = 448

    @property
    def head_dim(self) -> int:
        return self.d_model // self.n_heads

    @property
    def n_frames(self) -> int:
        return 2 * self.n_audio_ctx

    def as_dict(self) -> Dict[str, int]:
        d = asdict(self)
        d.pop("name")
        return d


PRESETS: Dict[str, WhisperDims] = {
    # micro: hd stays 64 (the attention kernels are specialised for it), 100 mel frames -> 50 positions.
    "micro": WhisperDims("micro", 80, 50, 128, 2, 256, 2, 2, 512, 32),
    "tiny": WhisperDims("tiny", 80, 1500, 384, 6, 1536, 4, 4, 51865),
    "base": WhisperDims("base", 80, 1500, 512, 8, 2048, 6, 6, 51865),
    "small": WhisperDims("small", 80, 1500, 768, 12, 3072, 12, 12, 51865),
    "medium": WhisperDims("medium", 80, 1500, 1024, 16, 4096, 24, 24, 51865),
    "large-v3": WhisperDims("large-v3", 128, 1500, 1280, 20, 5120, 32, 32, 51866),
    "large-v3-turbo": WhisperDims("large-v3-turbo", 128, 1500, 1280, 20, 5120, 32, 4, 51866),
    # large-v3 WIDTH at 2 + 2 layers: every kernel runs the benchmarked shapes while the CPU oracle stays affordable
    # (used by the parity tests of configs C3-C5, e.g. WhisperModel("synthetic:large-v3-w2"))
    "large-v3-w2": WhisperDims("large-v3-w2", 128, 1500, 1280, 20, 5120, 2, 2, 51866),
}


@dataclass(frozen=True)
class SpecialTokens:
    """Token ids the logits processors need.  Multilingual Whisper numbering; large-v3 shifts by one
    because it has one more language token ([HF] generation config of openai/whisper-large-v3)."""
    eot: int
    sot: int
    transcribe: int
    translate: int
    sot_prev: int
    no_speech: int
    no_timestamps: int
    timestamp_begin: int
    lang_zh: int
    sot_lm: int = -1   # <|startoflm|> (sot_prev - 1 in the released vocabularies); -1: the vocabulary has none

    @staticmethod
    def for_vocab(vocab: int) -> "SpecialTokens":
        if vocab == 51866:  # large-v3 family: 100 languages
            return SpecialTokens(eot=50257, sot=50258, transcribe=50360, translate=50359, sot_prev=50362,
                                 no_speech=50363, no_timestamps=50364, timestamp_begin=50365, lang_zh=50260, sot_lm=50361)
        if vocab == 51865:  # multilingual tiny..large-v2: 99 languages
            return SpecialTokens(eot=50257, sot=50258, transcribe=50359, translate=50358, sot_prev=50361,
                                 no_speech=50362, no_timestamps=50363, timestamp_begin=50364, lang_zh=50260, sot_lm=50360)
        # synthetic small vocabularies (micro): carve the specials from the top of the range so the
        # processors still see "text < eot < specials < timestamps".
        ts = vocab - 64  # 64 timestamp tokens
        return SpecialTokens(eot=ts - 8, sot=ts - 7, lang_zh=ts - 6, translate=ts - 5, transcribe=ts - 4,
                             sot_prev=ts - 3, no_speech=ts - 2, no_timestamps=ts - 1, timestamp_begin=ts)


# [HF] configuration_whisper.py NON_SPEECH_TOKENS_MULTI: the suppress list every multilingual checkpoint
# ships in generation_config.json (ids are vocabulary positions of punctuation/music-note tokens).
NON_SPEECH_TOKENS_MULTI: Tuple[int, ...] = (
    1, 2, 7, 8, 9, 10, 14, 25, 26, 27, 28, 29, 31, 58, 59, 60, 61, 62, 63, 90, 91, 92, 93, 359, 503, 522,
    542, 873, 893, 902, 918, 922, 931, 1350, 1853, 1982, 2460, 2627, 3246, 3253, 3268, 3536, 3846, 3961,
    4183, 4667, 6585, 6647, 7273, 9061, 9383, 10428, 10929, 11938, 12033, 12331, 12562, 13793, 14157,
    14635, 15265, 15618, 16553, 16604, 18362, 18956, 20075, 21675, 22520, 26130, 26161, 26435, 28279,
    29464, 31650, 32302, 32470, 36865, 42863, 47425, 49870, 50254,
)


@dataclass
class DecodeOptions:
    """Decoding switches.  Defaults are the literals every reference call site passes
    (asr_core.py:159-167, file_asr.py:457-465, faster_whisper_asr.py:139-149) except beam_size,
    which the metric fixes at greedy (BASELINE.json)."""
    language: str = "zh"
    task: str = "transcribe"
    beam_size: int = 1
    word_timestamps: bool = False
    vad_filter: bool = False
    condition_on_previous_text: bool = True
    initial_prompt: str = ""
    without_timestamps: bool = False
    max_new_tokens: int = 224
    suppress_blank: bool = True
    suppress_tokens: Optional[List[int]] = field(default=None)  # None -> NON_SPEECH_TOKENS_MULTI + specials
    max_initial_timestamp_index: int = 50
    suppress_eot: bool = False  # benchmark mode: fixed-length decode
