"""Entry point kept from the reference (tests/baseline_benchmark.py): autoregressive decoding, same command line."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magicdec_amd.cli import baseline_main  # noqa: E402

if __name__ == "__main__":
    baseline_main()
