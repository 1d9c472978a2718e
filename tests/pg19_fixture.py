"""A seeded synthetic "PG-19" corpus + a deterministic stub tokenizer (inputs of the PG-19 adapter fixture).

Shared by oracle/gen_golden.py (runs the REAL Data/data_converter.py:convert_pg19_dataset on them) and
tests/test_host_cpu.py (runs magicdec_amd.data.tokenize_pg19 on the same files)."""
import json
import os
import random
import zlib

import torch

N_BOOKS, SEQ_LEN, END = 52, 64, 3      # 52 books on disk: the reference reads the first 50


class WordTokenizer:
    """One token per whitespace-separated word (crc32 of the word mod 30000, +10), BOS prepended like HF tokenizers."""
    bos_token_id, eos_token_id, unk_token_id = 1, 2, None

    def __init__(self, bos=True):
        if not bos:
            self.bos_token_id = None

    def encode(self, text, return_tensors=None, **kw):
        ids = [1] + [10 + zlib.crc32(w.encode()) % 30000 for w in text.split()]
        return torch.tensor([ids], dtype=torch.long) if return_tensors == "pt" else ids


def write_corpus(root, seed=5):
    """<root>/Data/pg19/books.jsonl : one {"text": ...} record per book, 8000..8400 words each (so that after the
    8000-token skip a book yields 0..6 chunks of 64 tokens, the last one always dropped)."""
    rnd = random.Random(seed)
    d = os.path.join(root, "Data", "pg19")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "books.jsonl"), "w") as f:
        for b in range(N_BOOKS):
            n = 8000 + rnd.randrange(0, 400) if b != 7 else 8000 + 3 * SEQ_LEN - 1   # book 7: exactly 3 full chunks
            words = [f"w{rnd.randrange(0, 5000)}" for _ in range(n)]
            f.write(json.dumps({"text": " ".join(words), "short_book_title": f"book {b}"}) + "\n")
    return d
