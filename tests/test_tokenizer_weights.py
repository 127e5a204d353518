"""Text-side inputs of the scoring path: caption tokenisation and checkpoint loading
(SURVEY.md §8(f) rank 1).  CPU only."""
import json
import os

import numpy as np
import pytest

from dream2real_amd import clip_model
from dream2real_amd.tokenizer import ClipBpeTokenizer, bytes_to_unicode

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def tok():
    return ClipBpeTokenizer.from_files(os.path.join(G, "bpe_vocab.json"), os.path.join(G, "bpe_merges.txt"))


@pytest.fixture(scope="module")
def cases():
    return json.load(open(os.path.join(G, "bpe_cases.json"), encoding="utf-8"))


def test_byte_alphabet_is_a_bijection_onto_printable_characters():
    t = bytes_to_unicode()
    assert len(t) == 256 and len(set(t.values())) == 256
    assert t[ord("a")] == "a" and t[ord(" ")] == "Ġ" and t[0] == "Ā"
    assert all(not c.isspace() for c in t.values())


def test_input_ids_equal_hugging_face_clip_tokenizer(tok, cases):
    """every golden case was produced by transformers.CLIPTokenizer on the same vocabulary"""
    for c in cases["cases"]:
        assert tok.encode(c["text"]) == c["input_ids"], c["text"]


def test_batch_is_right_padded_with_the_end_token_like_padding_true(tok, cases):
    ids, mask = tok(cases["batch_texts"])
    assert ids.dtype == np.int32 and ids.tolist() == cases["batch_input_ids"]
    assert mask.tolist() == cases["batch_attention_mask"]
    full, _ = tok(cases["batch_texts"], pad_to_context=True)
    assert full.shape == (len(cases["batch_texts"]), cases["context_length"])
    assert (full[:, ids.shape[1]:] == tok.eos_id).all()


def test_truncation_keeps_the_end_token_and_empty_caption_is_two_tokens(tok):
    long_ids = tok.encode("a " * 200)
    assert len(long_ids) == tok.context_length and long_ids[0] == tok.bos_id and long_ids[-1] == tok.eos_id
    assert tok.encode("") == [tok.bos_id, tok.eos_id]
    assert tok.encode("   \n\t ") == [tok.bos_id, tok.eos_id]


def test_first_end_token_is_where_the_text_tower_pools(tok):
    """HF pools the text tower at the first <|endoftext|>; padding repeats that id after it"""
    ids, mask = tok(["an apple", "an apple inside a blue and white bowl"])
    first_eos = (ids == tok.eos_id).argmax(axis=1)
    assert (first_eos == mask.sum(axis=1) - 1).all()


@pytest.mark.parametrize("dtype", ["float32", "float16", "bfloat16"])
def test_safetensors_checkpoint_round_trip(tmp_path, dtype):
    import torch
    from safetensors.torch import save_file
    cfg = clip_model.CLIP_CONFIGS["vit_tiny"]
    sd = clip_model.random_clip_state_dict(cfg, seed=3)
    tdt = getattr(torch, dtype)
    tensors = {k: torch.from_numpy(np.asarray(v, np.float32).reshape(np.shape(v) or (1,)).copy()).to(tdt).contiguous()
               for k, v in sd.items()}
    tensors["vision_model.embeddings.position_ids"] = torch.arange(17).reshape(1, -1)
    path = str(tmp_path / "model.safetensors")
    save_file(tensors, path)
    cfg2, sd2 = clip_model.load_clip_safetensors(path)
    for k in ("patch_size", "hidden_size", "num_layers", "num_heads", "mlp", "image_size", "proj",
              "text_hidden", "text_layers", "text_heads", "text_mlp", "vocab", "ctx"):
        assert cfg2[k] == cfg[k], k
    assert "vision_model.embeddings.position_ids" not in sd2
    tol = {"float32": 0.0, "float16": 1e-3, "bfloat16": 8e-3}[dtype]
    for k, v in sd.items():
        a, b = np.asarray(v, np.float32), np.asarray(sd2[k], np.float32)
        assert a.reshape(-1).shape == b.reshape(-1).shape
        assert np.max(np.abs(a.reshape(-1) - b.reshape(-1))) <= tol * max(1.0, float(np.max(np.abs(a)))), k
    # the loaded dict packs into the blobs the C ABI takes
    assert clip_model.pack_vision_weights(sd2, cfg2).size == clip_model.pack_vision_weights(sd, cfg).size
    assert clip_model.pack_text_weights(sd2, cfg2).size == clip_model.pack_text_weights(sd, cfg).size
