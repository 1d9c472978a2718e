"""The tiny GQA configs the golden fixtures were generated with (mirrors oracle/gen_golden.py:TINY)."""
import json
import os

import numpy as np
import torch

from oracle.magicdec_ref import RefConfig, init_state_dict

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TINY = {
    "tinytgt": (RefConfig(n_layer=2, n_head=8, n_local_heads=2, dim=512, intermediate_size=1024, vocab_size=2048,
                          rope_base=500000.0, scaling_factor=8, high_freq_factor=4, low_freq_factor=1,
                          original_max_position_embeddings=8192), 11, 0.1),
    "tinydrf": (RefConfig(n_layer=1, n_head=8, n_local_heads=2, dim=512, intermediate_size=1024, vocab_size=2048,
                          rope_base=10000.0), 12, 0.1),
}
B, S, MAX_LEN, GAMMA, BUDGET, EOT_1, EOT_2 = 2, 416, 512, 3, 129, 2, 0


def tiny(name):
    cfg, seed, wo = TINY[name]
    return cfg, init_state_dict(cfg, seed, wo_scale=wo)


def synthetic_batches(n_seq=12, vocab=2048, prefix=S, seed=123):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(4, vocab, (n_seq, prefix), generator=g)
    ids[:, 0] = 1
    return [ids[i:i + B] for i in range(0, n_seq, B)]


def load_json(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def from_bits(a):
    return torch.from_numpy(a.astype(np.int16)).view(torch.bfloat16)
