"""Entry point kept from the reference (tests/SnapKV/selfspec_benchmark.py): self-speculation, same command line."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from magicdec_amd.cli import selfspec_main  # noqa: E402

if __name__ == "__main__":
    selfspec_main("SnapKV")
