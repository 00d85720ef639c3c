"""CLIP byte-level BPE tokenizer (host side of the text path).

The reference tokenises with ``transformers.CLIPTokenizer`` in the diffusion trainer / pipeline
(diffusion/train_controlnet_genima.py:885-891 ``tokenize_captions``: ``padding="max_length", truncation=True``;
the diffusers pipeline does the same for ``prompt=``) and with openai ``clip.tokenize`` for the controller's language goal
(controller/env/rlbench_utils.py:156).  Both are the same BPE over the same ``vocab.json`` / ``merges.txt`` (49408 entries); they
differ only in the pad id (SD-2.x and SDXL's second tokenizer pad with ``"!"`` = 0, SD-1.x / SDXL's first with ``<|endoftext|>``;
``clip.tokenize`` pads with 0) and in what happens to over-long text (truncate vs raise).

This is an independent implementation (no ``transformers`` import): NFC + whitespace collapse + lower-case, the CLIP split
pattern, byte -> printable-unicode mapping, greedy lowest-rank pair merging with the ``</w>`` end-of-word marker.
``tests/test_tokenizer_cpu.py`` pins it against the installed ``transformers.CLIPTokenizer`` (Rust ``tokenizers`` backend) on a
BPE model trained in the test -- the real vocabulary is not available offline.
"""
from __future__ import annotations

import html
import json
import os
import unicodedata
from functools import lru_cache
from types import SimpleNamespace
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np
import regex
import torch

BOS, EOS = "<|startoftext|>", "<|endoftext|>"
_SPLIT = regex.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
                       regex.IGNORECASE)


@lru_cache()
def bytes_to_unicode() -> Dict[int, str]:
    """The GPT-2 / CLIP reversible byte -> printable code point table (printable bytes map to themselves, the other 68 to
    code points from 256 upwards)."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return {b: chr(c) for b, c in zip(bs, cs)}


def _clean(text: str, openai: bool) -> str:
    if openai:  # clip.simple_tokenizer: ftfy.fix_text (identity on well-formed text; ftfy is not needed for it) + double html.unescape
        text = html.unescape(html.unescape(text))
    text = unicodedata.normalize("NFC", text)
    text = regex.sub(r"\s+", " ", text)
    if openai:
        text = text.strip()
    return text.lower()


class CLIPTokenizer:
    model_max_length = 77

    def __init__(self, vocab: Union[str, Dict[str, int]], merges: Union[str, Sequence[Union[str, Tuple[str, str]]]],
                 pad_token: Optional[str] = None, model_max_length: int = 77):
        if isinstance(vocab, str):
            with open(vocab, encoding="utf-8") as f:
                vocab = json.load(f)
        if isinstance(merges, str):
            with open(merges, encoding="utf-8") as f:
                lines = f.read().split("\n")
            merges = [ln for ln in lines if ln and not ln.startswith("#version")]
        self.encoder: Dict[str, int] = dict(vocab)
        self.decoder = {v: k for k, v in self.encoder.items()}
        pairs = [tuple(m.split()) if isinstance(m, str) else tuple(m) for m in merges]
        self.bpe_ranks = {p: i for i, p in enumerate(pairs)}
        self.byte_encoder = bytes_to_unicode()
        self.bos_token_id, self.eos_token_id = self.encoder[BOS], self.encoder[EOS]
        self.pad_token = pad_token if pad_token is not None else EOS
        self.pad_token_id = self.encoder[self.pad_token]
        self.vocab_size = len(self.encoder)
        self.model_max_length = model_max_length
        self._cache: Dict[str, List[str]] = {}
        # transformers splits every special token out of the raw text first -- including the pad token, so an SD-2.x tokenizer
        # (pad_token "!") maps a literal "!" to id 0 rather than to "!</w>"
        specials = sorted({BOS, EOS, self.pad_token}, key=len, reverse=True)
        self._special_split = regex.compile("(" + "|".join(regex.escape(t) for t in specials) + ")")
        self._specials = set(specials)

    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = None):
        """Reads ``vocab.json`` + ``merges.txt`` (+ the pad token from ``special_tokens_map.json`` / ``tokenizer_config.json``)
        of a diffusers pipeline's ``tokenizer/`` directory."""
        d = os.path.join(path, subfolder) if subfolder else path
        vocab, merges = os.path.join(d, "vocab.json"), os.path.join(d, "merges.txt")
        if not (os.path.exists(vocab) and os.path.exists(merges)):
            raise FileNotFoundError(f"no CLIP BPE model under {d} (vocab.json + merges.txt)")
        pad, maxlen = None, 77
        for fn in ("special_tokens_map.json", "tokenizer_config.json"):
            p = os.path.join(d, fn)
            if os.path.exists(p):
                with open(p, encoding="utf-8") as f:
                    cfg = json.load(f)
                tok = cfg.get("pad_token")
                if isinstance(tok, dict):
                    tok = tok.get("content")
                if tok and pad is None:
                    pad = tok
                if isinstance(cfg.get("model_max_length"), int) and cfg["model_max_length"] < 100000:
                    maxlen = cfg["model_max_length"]
        return cls(vocab, merges, pad_token=pad, model_max_length=maxlen)

    # ---- BPE --------------------------------------------------------------------------------------------------------------
    def _bpe(self, token: str) -> List[str]:
        hit = self._cache.get(token)
        if hit is not None:
            return hit
        word = list(token[:-1]) + [token[-1] + "</w>"]
        while len(word) > 1:
            best, best_rank = None, None
            for pair in zip(word[:-1], word[1:]):
                r = self.bpe_ranks.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            a, b = best
            out, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == a and word[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = out
        self._cache[token] = word
        return word

    def encode(self, text: str, openai: bool = False) -> List[int]:
        """Token ids of ``text`` without BOS / EOS."""
        ids: List[int] = []
        unk = self.eos_token_id  # CLIPTokenizer's unk_token is <|endoftext|>
        segments = [text] if openai else self._special_split.split(text)  # clip.tokenize has no added-token pass
        for seg in segments:
            if not openai and seg in self._specials:
                ids.append(self.encoder[seg])
                continue
            for piece in _SPLIT.findall(_clean(seg, openai)):
                if piece in (BOS, EOS):
                    ids.append(self.encoder[piece])
                    continue
                mapped = "".join(self.byte_encoder[b] for b in piece.encode("utf-8"))
                ids.extend(self.encoder.get(t, unk) for t in self._bpe(mapped))
        return ids

    # ---- transformers call surface (tokenize_captions, the pipelines' prompt=) ----------------------------------------------------
    def __call__(self, text: Union[str, Iterable[str]], padding="max_length", max_length: Optional[int] = None, truncation=True,
                 return_tensors="pt", **kw):
        texts = [text] if isinstance(text, str) else list(text)
        L = max_length or self.model_max_length
        rows = []
        for t in texts:
            ids = [self.bos_token_id] + self.encode(t) + [self.eos_token_id]
            if len(ids) > L:
                if not truncation:
                    raise ValueError(f"prompt has {len(ids)} tokens > max_length {L}")
                ids = ids[: L - 1] + [self.eos_token_id]
            rows.append(ids)
        if padding == "max_length":
            width = L
        elif padding in (True, "longest"):
            width = max(len(r) for r in rows)
        else:
            width = None
        if width is None:
            if return_tensors is None:
                return SimpleNamespace(input_ids=rows)
            if len({len(r) for r in rows}) != 1:
                raise ValueError("rows of different length need padding= to become a tensor")
            width = len(rows[0])
        arr = np.full((len(rows), width), self.pad_token_id, dtype=np.int64)
        for i, r in enumerate(rows):
            arr[i, : len(r)] = r
        if return_tensors is None:
            return SimpleNamespace(input_ids=arr.tolist())
        return SimpleNamespace(input_ids=torch.from_numpy(arr))

    # ---- openai clip.tokenize (controller/env/rlbench_utils.py:156) -------------------------------------------------------------
    def tokenize(self, texts: Union[str, Iterable[str]], context_length: int = 77, truncate: bool = False) -> torch.Tensor:
        """``clip.tokenize``: int32 [n, context_length], zero padded, BOS ... EOS; raises on over-long text unless ``truncate``."""
        texts = [texts] if isinstance(texts, str) else list(texts)
        out = torch.zeros(len(texts), context_length, dtype=torch.int32)
        for i, t in enumerate(texts):
            ids = [self.bos_token_id] + self.encode(t, openai=True) + [self.eos_token_id]
            if len(ids) > context_length:
                if not truncate:
                    raise RuntimeError(f"Input {t} is too long for context length {context_length}")
                ids = ids[:context_length]
                ids[-1] = self.eos_token_id
            out[i, : len(ids)] = torch.tensor(ids, dtype=torch.int32)
        return out
