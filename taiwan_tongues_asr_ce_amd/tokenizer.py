"""Host-side detokenisation (a12).  The reference ships `tokenizer.json` next to the CTranslate2 model
(faster_whisper_asr.py:38) and faster-whisper reads it with the HF `tokenizers` library; this module does
the same when the file exists.  No tokenizer file exists offline, so synthetic-weight runs use a
byte-level stub that is only meant to make ids printable and round-trippable in tests."""
from __future__ import annotations

import os
from typing import List, Optional, Sequence


class ByteStubTokenizer:
    """ids 0..255 are raw bytes; every other id below `n_text` renders as a private-use code point so that
    decode() is injective on text ids; special ids render as ''."""

    def __init__(self, vocab: int, n_text: Optional[int] = None):
        self.vocab = vocab
        self.n_text = vocab if n_text is None else n_text

    def encode(self, text: str) -> List[int]:
        return list(text.encode("utf-8"))

    def decode(self, ids: Sequence[int]) -> str:
        out, buf = [], bytearray()
        for t in ids:
            if 0 <= t < 256:
                buf.append(t)
                continue
            if buf:
                out.append(buf.decode("utf-8", errors="replace"))
                buf = bytearray()
            if t < self.n_text:
                out.append(chr(0xF0000 + (t % 0xFFFD)))
        if buf:
            out.append(buf.decode("utf-8", errors="replace"))
        return "".join(out)


class HFTokenizer:
    def __init__(self, path: str):
        from tokenizers import Tokenizer
        self.tk = Tokenizer.from_file(path)

    def encode(self, text: str) -> List[int]:
        return self.tk.encode(text, add_special_tokens=False).ids

    def decode(self, ids: Sequence[int]) -> str:
        return self.tk.decode(list(ids), skip_special_tokens=True)


def load_tokenizer(model_dir: Optional[str], vocab: int):
    if model_dir:
        p = os.path.join(model_dir, "tokenizer.json")
        if os.path.exists(p):
            return HFTokenizer(p)
    from .config import SpecialTokens
    return ByteStubTokenizer(vocab, SpecialTokens.for_vocab(vocab).eot)
