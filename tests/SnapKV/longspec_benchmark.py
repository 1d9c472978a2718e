"""Entry point kept from the reference (tests/SnapKV/longspec_benchmark.py): stand-alone-draft speculative decoding,
same command line.  e.g.
  python -m torch.distributed.run --standalone --nproc_per_node=8 tests/SnapKV/longspec_benchmark.py \
      --target checkpoints/meta-llama/Meta-Llama-3.1-8B/model.pth --model checkpoints/meta-llama/Llama-3.2-1B/model.pth \
      --rank_group 0 1 2 3 4 5 6 7 --draft_rank_group 0 1 2 3 --gamma 3 --B 64 --prefix_len 16032 --max_len 16128 \
      --draft_budget 257 --benchmark --compile
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from magicdec_amd.cli import longspec_main  # noqa: E402

if __name__ == "__main__":
    longspec_main("SnapKV")
