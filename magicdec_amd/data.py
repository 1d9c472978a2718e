"""Benchmark inputs.  The reference tokenises PG-19 books (Data/data_converter.py:44-58: 50 books, first 8000
tokens dropped, split into seq_len chunks, token 0 = BOS, repeated 20x); that corpus is absent from the reference
checkout and unreachable from the GPU box, so the default here is a synthetic PG-19-SHAPED dataset: token ids uniform
in [0, vocab) from torch.Generator(seed), BOS in column 0.  If a pre-tokenised tensor file exists at
Data/pg19/pg19_<seq_len>.pt it is used instead."""
from __future__ import annotations

import os

import torch
from torch.utils.data import TensorDataset


class OfflineTokenizer:
    """Stand-in used when transformers.AutoTokenizer cannot load `model_name` offline: carries only the ids
    the harness needs (Llama-3 family: <|end_of_text|>=128001, <|eot_id|>=128009, BOS=128000)."""
    eos_token, pad_token = "<|end_of_text|>", "<|end_of_text|>"
    eos_token_id, unk_token_id, bos_token_id = 128001, None, 128000

    def encode(self, text, **kw):
        return [128009]

    def decode(self, ids, **kw):
        return " ".join(str(int(i)) for i in ids)


def load_tokenizer(model_name):
    try:
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained(model_name, local_files_only=True)
        tok.pad_token = tok.eos_token
        return tok
    except Exception as e:   # no network / no local files
        print(f"[magicdec_amd] tokenizer '{model_name}' unavailable offline ({type(e).__name__}); using fixed Llama-3 ids")
        return OfflineTokenizer()


def convert_pg19_dataset(tokenizer=None, seq_len=4096, vocab_size=128256, num_sequences=None, seed=123, end=20):
    path = os.path.join("Data", "pg19", f"pg19_{seq_len}.pt")
    if os.path.exists(path):
        ids = torch.load(path)
        return TensorDataset(ids.repeat(end, 1))
    n = num_sequences if num_sequences is not None else 640
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, vocab_size, (n, seq_len), generator=g)
    ids[:, 0] = getattr(tokenizer, "bos_token_id", None) or 1
    return TensorDataset(ids)
