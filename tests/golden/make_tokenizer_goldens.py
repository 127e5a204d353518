#!/usr/bin/env python3
"""Golden vectors for dream2real_amd.tokenizer.ClipBpeTokenizer.

The reference tokenises its captions with the Hugging Face CLIP processor
(`CLIPProcessor.from_pretrained("openai/clip-vit-large-patch14-336")`, reference
clip_scoring.py:151,177).  The pretrained vocabulary cannot be downloaded here, so the pin is:
a small byte-level BPE vocabulary trained on the spot with the `tokenizers` library, handed to the
REAL `transformers.CLIPTokenizer` (same normaliser, pre-tokeniser, BPE model and post-processor as
the pretrained one), and its `input_ids` for a list of awkward strings.

Run once in the build container:  python tests/golden/make_tokenizer_goldens.py
Writes bpe_vocab.json, bpe_merges.txt, bpe_cases.json (data only) next to this file.
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))

CORPUS = [
    "an apple inside a blue and white bowl", "an apple and a blue and white bowl",
    "a photo of an apple", "a bad photo of the bowl", "a low resolution photo of a shelf",
    "the black 8 ball in a triangle of pool balls", "pool balls in a triangle", "a bottle on the shelf",
    "a cropped photo of a bright painting", "a dark photo", "a good photo of a pear next to a mug",
    "it's the robot's gripper, isn't it?", "we've moved 12 objects; they'll stay", "I'm done",
    "white bowl blue bowl apple apple apple bowl bowl photo photo of of of a a a the the",
]

CASES = [
    "an apple inside a blue and white bowl",
    "an apple and a blue and white bowl",
    "A Photo of   AN Apple",
    "  leading and trailing   spaces  ",
    "it's the robot's gripper, isn't it?",
    "we've moved 12 objects; they'll stay!!!",
    "pool balls: 8-ball (black) & 15 others...",
    "café naïve über",            # accented letters (NFC)
    "café",                              # decomposed e + combining acute -> NFC
    "中文 and \U0001f34e emoji",        # CJK + emoji: byte-level fallback
    "tab\tand\nnewline",
    "",
    "x",
    "zzzqqq unknownword 9876543210",
    "<|startoftext|> literal specials <|endoftext|>",
    "a " * 100,                                 # longer than the context: truncation
]


def main():
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, trainers, Regex
    from transformers import CLIPTokenizer

    # train a small byte-level BPE with CLIP's normaliser / pre-tokeniser / "</w>" suffix
    tok = Tokenizer(models.BPE(end_of_word_suffix="</w>", continuing_subword_prefix="", unk_token="<|endoftext|>"))
    tok.normalizer = normalizers.Sequence([normalizers.NFC(), normalizers.Replace(Regex(r"\s+"), " "), normalizers.Lowercase()])
    tok.pre_tokenizer = pre_tokenizers.Sequence([
        pre_tokenizers.Split(Regex(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"""),
                             behavior="removed", invert=True),
        pre_tokenizers.ByteLevel(add_prefix_space=False)])
    trainer = trainers.BpeTrainer(vocab_size=900, min_frequency=1, end_of_word_suffix="</w>",
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet(),
                                  special_tokens=[], show_progress=False)
    tok.train_from_iterator(CORPUS, trainer)
    model = json.loads(tok.to_str())["model"]
    vocab, merges = dict(model["vocab"]), [tuple(m) if isinstance(m, list) else tuple(m.split(" ")) for m in model["merges"]]
    # CLIP's layout: every byte symbol also exists with the end-of-word suffix; specials at the end
    from dream2real_amd.tokenizer import bytes_to_unicode
    for ch in bytes_to_unicode().values():
        for sym in (ch, ch + "</w>"):
            if sym not in vocab:
                vocab[sym] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)

    hf = CLIPTokenizer(vocab=vocab, merges=[tuple(m) for m in merges])
    cases = []
    for text in CASES:
        ids = hf(text, truncation=True, max_length=77)["input_ids"]
        cases.append({"text": text, "input_ids": ids})
    batch = hf(CASES[:4], padding=True, truncation=True, max_length=77)
    json.dump(vocab, open(os.path.join(HERE, "bpe_vocab.json"), "w"), ensure_ascii=False)
    with open(os.path.join(HERE, "bpe_merges.txt"), "w", encoding="utf-8") as f:
        f.write("#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n")
    json.dump({"cases": cases, "batch_texts": CASES[:4], "batch_input_ids": batch["input_ids"],
               "batch_attention_mask": batch["attention_mask"], "context_length": 77,
               "generator": "transformers.CLIPTokenizer %s" % __import__("transformers").__version__},
              open(os.path.join(HERE, "bpe_cases.json"), "w"), ensure_ascii=False, indent=1)
    print("vocab", len(vocab), "merges", len(merges), "cases", len(cases))


if __name__ == "__main__":
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    main()
