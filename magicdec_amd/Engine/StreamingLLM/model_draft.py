"""Same import path and names as the reference module; the implementation is shared (Engine/model_core.py)."""
from ..model_core import (Attention, FeedForward, KVCache, ModelArgs, RMSNorm, Transformer, TransformerBlock,  # noqa: F401
                          find_multiple, transformer_configs)
