"""CPU oracle for the Whisper hot path - test infrastructure only (see whisper_ref.py header)."""
