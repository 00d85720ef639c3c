"""genima_amd/tokenizer.py (own CLIP byte-level BPE) against the installed ``transformers.CLIPTokenizer`` (Rust ``tokenizers``
backend -- an independent implementation) on a BPE model trained here with the CLIP conventions (byte alphabet, ``</w>`` suffix).
The real 49408-entry vocabulary is not available offline; the algorithm is what is pinned.
Reference call sites: diffusion/train_controlnet_genima.py:885-891 (``tokenize_captions``), controller/env/rlbench_utils.py:156
(``clip.tokenize``)."""
import json
import os

import pytest
import torch

from genima_amd.tokenizer import BOS, EOS, CLIPTokenizer, bytes_to_unicode

CORPUS = [
    "tiled perspectives of a robot arm executing 'open the box'",
    "tiled perspectives of a robot arm executing 'close the jar'",
    "put the red block in the drawer, then slide it shut.",
    "take the USB out of the computer; it's the robot's 2nd task (of 25)!",
    "stack 4 cups -- don't drop them... we'll see",
    "turn the tap left / right and sweep dirt to the dustpan",
    "naïve café über straße façade", "reach target, push buttons, meat off grill, phone on base",
] * 4


@pytest.fixture(scope="module")
def bpe_model():
    from tokenizers import Tokenizer, models, pre_tokenizers, trainers
    from tokenizers import Regex, normalizers

    tok = Tokenizer(models.BPE(continuing_subword_prefix="", end_of_word_suffix="</w>"))
    tok.normalizer = normalizers.Sequence([normalizers.NFC(), normalizers.Replace(Regex(r"\s+"), " "), normalizers.Lowercase()])
    tok.pre_tokenizer = pre_tokenizers.Sequence([
        pre_tokenizers.Split(Regex(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"""),
                             behavior="removed", invert=True),
        pre_tokenizers.ByteLevel(add_prefix_space=False)])
    alphabet = sorted(bytes_to_unicode().values())
    trainer = trainers.BpeTrainer(vocab_size=900, initial_alphabet=alphabet, end_of_word_suffix="</w>",
                                  special_tokens=[], show_progress=False)
    tok.train_from_iterator(CORPUS, trainer)
    model = json.loads(tok.to_str())["model"]
    vocab = dict(model["vocab"])
    # CLIP's layout: every byte symbol also exists with the end-of-word marker; the two specials come last
    for ch in alphabet:
        vocab.setdefault(ch + "</w>", len(vocab))
    vocab[BOS] = len(vocab)
    vocab[EOS] = len(vocab)
    merges = [tuple(m) if not isinstance(m, str) else tuple(m.split()) for m in model["merges"]]
    return vocab, merges


TEXTS = [
    "tiled perspectives of a robot arm executing 'open the box'",
    "Tiled   perspectives\tof a ROBOT arm\nexecuting 'close jar'  ",
    "it's the robot's 2nd task, we'll see: 12345 (of 25)!!",
    "unseen wörds: zebra-crossing? ~tilde~ €uro 東京 emoji 🙂",
    "",
    "a",
    "<|startoftext|> nested specials <|endoftext|> tail",
    " ".join(["put the red block in the drawer"] * 20),  # > 77 tokens: truncation
]


def test_matches_transformers_clip_tokenizer(bpe_model):
    from transformers import CLIPTokenizer as HF

    vocab, merges = bpe_model
    for pad in (EOS, "!"):
        hf = HF(vocab=vocab, merges=list(merges), pad_token=pad)
        mine = CLIPTokenizer(vocab, merges, pad_token=pad)
        assert mine.pad_token_id == hf.pad_token_id and mine.bos_token_id == hf.bos_token_id and mine.eos_token_id == hf.eos_token_id
        for t in TEXTS:
            ref = hf(t, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
            got = mine(t, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
            assert got.dtype == torch.int64 and tuple(got.shape) == (1, 77)
            assert torch.equal(got, ref), (t, got[0, :20].tolist(), ref[0, :20].tolist())
        ref = hf(TEXTS[:4], padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
        got = mine(TEXTS[:4], padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
        assert torch.equal(got, ref)


def test_from_pretrained_layout_and_clip_tokenize(bpe_model, tmp_path):
    vocab, merges = bpe_model
    d = tmp_path / "tokenizer"
    os.makedirs(d)
    (d / "vocab.json").write_text(json.dumps(vocab), encoding="utf-8")
    (d / "merges.txt").write_text("#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n", encoding="utf-8")
    (d / "special_tokens_map.json").write_text(json.dumps({"pad_token": "!", "bos_token": {"content": BOS}}))
    (d / "tokenizer_config.json").write_text(json.dumps({"model_max_length": 77, "pad_token": "<|endoftext|>"}))
    tok = CLIPTokenizer.from_pretrained(str(tmp_path), "tokenizer")
    assert tok.pad_token == "!" and tok.model_max_length == 77  # special_tokens_map wins (SD-2.x pads with "!")
    direct = CLIPTokenizer(vocab, merges, pad_token="!")
    assert torch.equal(tok("open the box").input_ids, direct("open the box").input_ids)
    # clip.tokenize: int32, zero padded, EOT is the arg-max id (the ACT text tower pools there: controller/method/genima_act.py:338)
    ids = tok.tokenize(["open the box", "close the jar &amp; stack   cups"])
    assert ids.dtype == torch.int32 and tuple(ids.shape) == (2, 77)
    n = int((ids[0] != 0).sum())
    assert ids[0, 0] == tok.bos_token_id and ids[0, n - 1] == tok.eos_token_id and int(ids[0].argmax()) == n - 1
    assert ids[1].tolist()[: 1 + len(tok.encode("close the jar & stack cups"))][1:] == tok.encode("close the jar & stack cups")
    with pytest.raises(RuntimeError):
        tok.tokenize(" ".join(["drawer"] * 200))
    assert tok.tokenize(" ".join(["drawer"] * 200), truncate=True)[0, -1] == tok.eos_token_id
    with pytest.raises(FileNotFoundError):
        CLIPTokenizer.from_pretrained(str(tmp_path), "no_such_dir")
