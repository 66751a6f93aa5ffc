"""Plugin-level drop-in: the reference's ASR adapter interface
(api/stt_streaming/src/asr/asr_interface.py:1-15, asr_factory.py:9-30, faster_whisper_asr.py:16-303) with
an MI355X backend registered under "mi355x_whisper" (and answering to "faster_whisper", so the streaming
server's `ASRFactory.create_asr_pipeline("faster_whisper", ...)` call needs no change)."""
from __future__ import annotations

import logging
import os
from typing import Any, Dict, Optional

import numpy as np

logger = logging.getLogger(__name__)


class ASRInterface:
    async def transcribe(self, client):
        raise NotImplementedError("This method should be implemented by subclasses.")

    def warm_up(self):
        raise NotImplementedError("This method should be implemented by subclasses.")


def pcm16_bytes_to_float(buf: bytes) -> np.ndarray:
    """The streaming client hands over 16 kHz mono s16le bytes (client.py:32-35; audio_utils.py:5-29 wraps
    them into a wav that faster-whisper decodes to float32 / 32768)."""
    return np.frombuffer(bytes(buf), dtype="<i2").astype(np.float32) / 32768.0


class MI355XWhisperASR(ASRInterface):
    def __init__(self, **kwargs):
        from .model import WhisperModel
        model_size = kwargs.get("model_size", "large-v3-turbo")  # faster_whisper_asr.py:21
        model_path = kwargs.get("model_path") or model_size
        device = kwargs.get("device", "cuda")
        compute_type = kwargs.get("compute_type", "float16")   # api/config.py:11-12
        # decode rows: the adapter's default call is beam 5 with a best_of-5 temperature fallback (faster_whisper_asr.py:139-149),
        # so the model needs at least that many rows or transcribe() refuses the beam
        self.asr_pipeline = WhisperModel(model_path, device=device, compute_type=compute_type,
                                         max_batch=max(8, int(kwargs.get("max_batch", 8))))
        # health-check attributes (faster_whisper_asr.py:111-114, streaming_asr.py:455-463)
        self.device, self.compute_type, self.model_size, self.model_path = device, compute_type, model_size, model_path
        self.default_transcribe_kwargs = {  # faster_whisper_asr.py:139-149
            "word_timestamps": False, "vad_filter": True, "beam_size": 5, "condition_on_previous_text": True,
            "initial_prompt": "繁體中文",
        }
        self.text_filter = kwargs.get("text_filter")  # utils.filter_text of the reference, injected by the caller
        # faster_whisper_asr.py:186-198 retries an empty VAD-filtered result without VAD - but it re-opens a temp file it deleted
        # at :179, always lands in its `except: pass` and returns None.  The EFFECTIVE reference behaviour (None for a chunk the
        # VAD empties: the commonest streaming case is an all-silence chunk, on which Whisper hallucinates) is the default here;
        # the retry the reference's author intended is an explicit opt-in, and its result is still dropped when the decoder
        # itself calls the chunk silence (no_speech_prob) or is unsure of it (avg_logprob).
        self.retry_without_vad = bool(kwargs.get("retry_without_vad", False))
        self.retry_no_speech_threshold = float(kwargs.get("retry_no_speech_threshold", 0.6))
        self.retry_logprob_threshold = float(kwargs.get("retry_logprob_threshold", -1.0))
        fn = kwargs.get("vad_speech_prob_fn")
        if fn is not None:   # a speech-probability source handed to the adapter lands on the model (model.vad_speech_prob_fn)
            self.asr_pipeline.vad_speech_prob_fn = fn

    async def transcribe(self, client) -> Optional[Dict[str, Any]]:
        try:
            audio = pcm16_bytes_to_float(client.scratch_buffer)
            kw = dict(self.default_transcribe_kwargs)
            kw["language"] = "zh"  # faster_whisper_asr.py:161
            import warnings
            with warnings.catch_warnings():
                # only the known, documented degradation is silenced per utterance (it is logged once at start-up by the
                # server): anything else - an option that is ignored, a beam that does not fit - stays visible
                warnings.filterwarnings("ignore", message="vad_filter=True")
                segments, info = self.asr_pipeline.transcribe(audio, **kw)
                segments = list(segments)
                if len(segments) == 0 and self.retry_without_vad and kw.get("vad_filter") and self._vad_is_active():
                    # opt-in only (see __init__): once more with VAD off, so that an over-eager filter cannot swallow a whole
                    # utterance; segments the decoder itself marks as silence or as a low-confidence guess are dropped
                    try:
                        retry = dict(kw, vad_filter=False)
                        segments, info = self.asr_pipeline.transcribe(audio, **retry)
                        segments = [s for s in segments
                                    if not (getattr(s, "no_speech_prob", 0.0) > self.retry_no_speech_threshold
                                            or getattr(s, "avg_logprob", 0.0) < self.retry_logprob_threshold)]
                    except Exception as e:      # the reference swallows a failing retry too (:197-198)
                        logger.debug("retry without VAD failed: %s", e)
                        segments = []
            if len(segments) == 0:
                return None
            text = " ".join(getattr(s, "text", "").strip() for s in segments)
            if self.text_filter is not None:
                filtered = self.text_filter(text)
                text = text if filtered is None else filtered
            words = [w for s in segments if getattr(s, "words", None) for w in s.words]
            duration = words[-1].end if words else getattr(segments[-1], "end", None)
            last = getattr(client, "last_start_time", 0) or 0
            return {  # same keys as faster_whisper_asr.py:240-255
                "language": getattr(info, "language", None),
                "language_probability": getattr(info, "language_probability", None),
                "final": True, "text": text, "duration": duration,
                "words": [{"word": w.word, "start": (w.start or 0) + last, "end": (w.end or 0) + last,
                           "probability": w.probability} for w in words],
            }
        except Exception as e:  # the reference logs and returns None (faster_whisper_asr.py:260-267)
            logger.error("transcribe failed: %s", e)
            return None

    def _vad_is_active(self) -> bool:
        """True when vad_filter=True really filters: the operator supplied a speech-probability source on the model, or the
        default kwargs opt into the energy stand-in."""
        if getattr(self.asr_pipeline, "vad_speech_prob_fn", None) is not None:
            return True
        if self.default_transcribe_kwargs.get("vad_speech_prob_fn") is not None:
            return True
        params = self.default_transcribe_kwargs.get("vad_parameters") or {}
        return params.get("backend") == "energy"

    def warm_up(self):
        wav = os.environ.get("TTASR_WARMUP_WAV")
        try:
            if wav and os.path.exists(wav):
                # faster_whisper_asr.py:289-294: the warm-up asks for word timestamps, which also warms the alignment pass
                segs, _ = self.asr_pipeline.transcribe(wav, word_timestamps=True, language="zh", initial_prompt="繁體中文")
            else:
                segs, _ = self.asr_pipeline.transcribe(np.zeros(16000, np.float32), language="zh", beam_size=1)
            list(segs)
        except Exception as e:
            logger.error("warm_up failed: %s", e)


class ASRFactory:
    _registry = {"mi355x_whisper": MI355XWhisperASR, "faster_whisper": MI355XWhisperASR}

    @staticmethod
    def create_asr_pipeline(type, **kwargs):
        if type == "mi355x_whisper_batched":  # micro-batching backend for the WebSocket server (streaming.py)
            from .streaming import BatchedWhisperASR
            return BatchedWhisperASR(**kwargs)
        cls = ASRFactory._registry.get(type)
        if cls is None:
            raise ValueError(f"不支援的 ASR 管道類型: {type}。目前只支援 {sorted(ASRFactory._registry)}")
        return cls(**kwargs)
