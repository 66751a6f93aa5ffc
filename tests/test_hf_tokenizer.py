"""HFTokenizer (tokenizer.json via the `tokenizers` library) with a real byte-level BPE built on the spot — the same
tokenizer family Whisper ships — and the word grouping of alignment.py on top of it."""
import pytest

from taiwan_tongues_asr_ce_amd import alignment as A
from taiwan_tongues_asr_ce_amd.tokenizer import HFTokenizer, load_tokenizer


@pytest.fixture(scope="module")
def tok(tmp_path_factory):
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    tk = Tokenizer(models.BPE())
    tk.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tk.decoder = decoders.ByteLevel()
    corpus = ["今天天氣很好，我們去公園散步。", "hello world, this is a test.", "語音辨識 speech recognition", "臺灣 台灣 taiwan"] * 4
    tk.train_from_iterator(corpus, trainers.BpeTrainer(vocab_size=300, special_tokens=["<|endoftext|>"],
                                                       initial_alphabet=pre_tokenizers.ByteLevel.alphabet()))
    d = tmp_path_factory.mktemp("tok")
    tk.save(str(d / "tokenizer.json"))
    t = load_tokenizer(str(d), 300)
    assert isinstance(t, HFTokenizer)
    return t


def test_round_trip_and_no_special_tokens_added(tok):
    for text in ("今天天氣很好", " hello world", "語音 speech，測試!", "罕見字𠮷野家"):
        ids = tok.encode(text)
        assert len(ids) > 0 and tok.decode(ids) == text
    assert tok.encode("") == []


def test_unicode_word_grouping_on_byte_level_bpe(tok):
    text = "罕見字𠮷好"                       # rare characters are split into several byte tokens
    ids = tok.encode(text)
    assert len(ids) > len(text)               # at least one character spans several tokens
    words, groups = A.split_tokens_on_unicode(tok, ids)
    assert "".join(words) == text and all("�" not in w for w in words)
    assert sum(len(g) for g in groups) == len(ids) and [t for g in groups for t in g] == ids
    assert all(len(w) >= 1 for w in words)
    words, groups = A.split_tokens_on_spaces(tok, tok.encode(" hello world , ok"), eot=10 ** 9)
    assert [w.strip() for w in words if w.strip()] == ["hello", "world", ",", "ok"]
