"""Benchmark inputs: the PG-19 adapter of the reference (Data/data_converter.py:44-58: 50 books, first 8000 tokens
dropped, split into seq_len chunks, last chunk discarded, token 0 = BOS, repeated 20x) and the tokenizer-driven EOT
ids (tests/SnapKV/longspec_benchmark.py:108-115, magicdec_amd/cli.py:_eot).  The corpus is absent from the reference
checkout and unreachable from the GPU box: without it the dataset is synthetic but PG-19-SHAPED."""
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


PG19_DIR = "Data/pg19/"       # Data/data_converter.py:45 (relative to the working directory, like the reference)
PG19_BOOKS = 50               # :48
PG19_SKIP = 8000              # :50 first tokens of every book dropped (front matter)


def _read_json_records(paths):
    """The `text` field of every record of JSON / JSON-lines files, in file order (what
    datasets.load_dataset("json", data_files=paths, split="train") yields for such files)."""
    import json
    for p in paths:
        with open(p, "r", encoding="utf-8") as f:
            head = f.read(1)
            f.seek(0)
            if head == "[":
                for rec in json.load(f):
                    yield rec["text"]
            else:
                for line in f:
                    line = line.strip()
                    if line:
                        yield json.loads(line)["text"]


def pg19_available(data_dir=PG19_DIR):
    return os.path.isdir(data_dir) and any(not n.startswith(".") for n in os.listdir(data_dir))


def tokenize_pg19(tokenizer, seq_len=4096, end=20, data_dir=PG19_DIR, n_books=PG19_BOOKS, skip=PG19_SKIP):
    """Data/data_converter.py:44-58: the first 50 books of the json files under Data/pg19/, each tokenised whole, its
    first 8000 tokens dropped, the rest split into seq_len chunks of which the LAST is always discarded (`[:-1]`,
    also when it is full), column 0 of every chunk overwritten with BOS (EOS if the tokenizer has no BOS), all chunks
    concatenated and the whole set repeated `end` times.  Returns the [n_chunks * end, seq_len] int64 tensor."""
    files = [os.path.join(data_dir, name) for name in os.listdir(data_dir)]          # :46-47 (listdir order)
    bos = tokenizer.bos_token_id if tokenizer.bos_token_id is not None else tokenizer.eos_token_id
    chunks = []
    for i, text in enumerate(_read_json_records(files)):
        if i >= n_books:
            break
        ids = tokenizer.encode(text, return_tensors="pt")[:, skip:]
        for c in ids.split(seq_len, dim=-1)[:-1]:
            c = c.clone()
            c[:, 0] = bos
            chunks.append(c)
    if i + 1 < n_books:
        raise IndexError(f"{data_dir} holds {i + 1} books, the reference indexes {n_books} (data_converter.py:49)")
    return torch.cat(chunks, dim=0).repeat(end, 1)


def convert_pg19_dataset(tokenizer=None, seq_len=4096, vocab_size=128256, num_sequences=None, seed=123, end=20,
                         data_dir=PG19_DIR):
    """The reference's convert_pg19_dataset(tokenizer, seq_len, end) when the PG-19 json files are present under
    Data/pg19/ (tokenize_pg19).  The corpus is neither in the reference checkout nor reachable from the GPU box, so
    without it the dataset is synthetic but PG-19-SHAPED: token ids uniform in [0, vocab) from torch.Generator(seed),
    BOS in column 0 (a pre-tokenised tensor at Data/pg19_<seq_len>.pt is used if present)."""
    if pg19_available(data_dir) and tokenizer is not None and not isinstance(tokenizer, OfflineTokenizer):
        return TensorDataset(tokenize_pg19(tokenizer, seq_len, end, data_dir))
    path = os.path.join("Data", f"pg19_{seq_len}.pt")
    if os.path.exists(path):
        ids = torch.load(path)
        return TensorDataset(ids.repeat(end, 1))
    print(f"[magicdec_amd] no PG-19 corpus under {data_dir}: synthetic PG-19-shaped token batches (seed {seed})")
    n = num_sequences if num_sequences is not None else 640
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, vocab_size, (n, seq_len), generator=g)
    ids[:, 0] = getattr(tokenizer, "bos_token_id", None) or 1
    return TensorDataset(ids)
