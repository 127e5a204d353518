"""CLIP byte-level BPE tokenizer (captions -> input_ids) for the text half of the scoring path.

The reference tokenises captions through `CLIPProcessor(text=..., padding=True)` (reference
clip_scoring.py:151,177), i.e. Hugging Face's CLIP tokenizer: NFC + whitespace collapse + lower
case, a regex pre-tokeniser, byte-level symbols, BPE merges with the `</w>` end-of-word suffix,
`<|startoftext|>` ... `<|endoftext|>` framing and right padding with `<|endoftext|>`.  The
pretrained vocabulary (`vocab.json` / `merges.txt` of openai/clip-vit-*) is not available offline;
it is supplied at run time.  Pinned against `transformers.CLIPTokenizer` on a small trained
vocabulary (tests/golden/make_tokenizer_goldens.py).
"""
from __future__ import annotations

import json
import unicodedata

import numpy as np
import regex

_PRETOKEN = regex.compile(
    r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""")
_SPACE = regex.compile(r"\s+")


def bytes_to_unicode() -> dict:
    """byte value -> printable unicode character of the byte-level alphabet: printable latin-1
    bytes stand for themselves, the other 68 are moved to U+0100.. in byte order."""
    keep = list(range(0x21, 0x7F)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    table, spill = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + spill)
            spill += 1
    return table


class ClipBpeTokenizer:
    def __init__(self, vocab: dict, merges, context_length: int = 77,
                 bos: str = "<|startoftext|>", eos: str = "<|endoftext|>"):
        self.vocab = dict(vocab)
        self.rank = {tuple(m): i for i, m in enumerate(merges)}
        self.context_length = int(context_length)
        self.bos_id, self.eos_id = self.vocab[bos], self.vocab[eos]
        self.unk_id = self.eos_id
        self._specials = {bos: self.bos_id, eos: self.eos_id}
        self._byte = bytes_to_unicode()
        self._cache = {}

    @classmethod
    def from_files(cls, vocab_json: str, merges_txt: str, **kw):
        vocab = json.load(open(vocab_json, encoding="utf-8"))
        merges = []
        for line in open(merges_txt, encoding="utf-8").read().split("\n"):
            if not line or line.startswith("#version"):
                continue
            a, b = line.split(" ")
            merges.append((a, b))
        return cls(vocab, merges, **kw)

    def _bpe(self, word: str):
        """symbols of one pre-token after applying the ranked merges (lowest rank first, every
        occurrence of the chosen pair in one pass)."""
        hit = self._cache.get(word)
        if hit is not None:
            return hit
        syms = [self._byte[b] for b in word.encode("utf-8")]
        syms[-1] += "</w>"
        while len(syms) > 1:
            best, best_rank = None, None
            for pair in zip(syms[:-1], syms[1:]):
                r = self.rank.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            out, i = [], 0
            while i < len(syms):
                if i + 1 < len(syms) and (syms[i], syms[i + 1]) == best:
                    out.append(syms[i] + syms[i + 1])
                    i += 2
                else:
                    out.append(syms[i])
                    i += 1
            syms = out
        self._cache[word] = syms
        return syms

    def encode(self, text: str) -> list:
        """ids of one caption with the start/end tokens, truncated to the context length (the end
        token is kept)."""
        text = _SPACE.sub(" ", unicodedata.normalize("NFC", text)).lower()
        ids = [self.bos_id]
        for tok in _PRETOKEN.findall(text):
            if tok in self._specials:
                ids.append(self._specials[tok])
                continue
            ids.extend(self.vocab.get(s, self.unk_id) for s in self._bpe(tok))
        ids.append(self.eos_id)
        if len(ids) > self.context_length:
            ids = ids[: self.context_length - 1] + [self.eos_id]
        return ids

    def __call__(self, captions, pad_to_context: bool = False):
        """-> (input_ids int32 [C,T], attention_mask int32 [C,T]); T = longest caption (the
        reference's padding=True) or the context length."""
        rows = [self.encode(c) for c in captions]
        T = self.context_length if pad_to_context else max(len(r) for r in rows)
        ids = np.full((len(rows), T), self.eos_id, np.int32)
        mask = np.zeros((len(rows), T), np.int32)
        for i, r in enumerate(rows):
            ids[i, : len(r)] = r
            mask[i, : len(r)] = 1
        return ids, mask
