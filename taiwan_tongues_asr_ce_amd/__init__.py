"""MI355X-native Whisper inference hot path behind the reference's WhisperModel.transcribe surface."""
from .config import PRESETS, WhisperDims, DecodeOptions, SpecialTokens, COMPUTE_F32, COMPUTE_BF16, COMPUTE_F16  # noqa: F401
