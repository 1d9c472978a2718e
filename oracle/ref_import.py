"""Import the REAL reference (read-only checkout) on CPU, in this container only.

TEST INFRASTRUCTURE ONLY -- used by oracle/gen_golden.py to generate the
committed fixtures under tests/golden/ and by nothing else.  The reference
checkout does not exist on the GPU box and nothing in the `-m gpu` tests,
smoke() or bench.py may import this module.

The reference is pure Python but (a) every module imports itself as the package
``MagicDec`` (Engine/SnapKV/backend.py:2-3), (b) it hard-imports ``flashinfer``
(absent here, see oracle/flashinfer_ref.py) and (c) registers its custom ops for
the "cuda" dispatch key only (Engine/utils.py:36).  Three shims, none of which
touches a file of the checkout:
  1. a directory holding a symlink ``MagicDec -> <reference root>`` on sys.path;
  2. ``sys.modules['flashinfer']`` = oracle.flashinfer_ref (restated semantics);
  3. ``torch.library.impl(name, "cuda")`` re-keyed to "cpu".
"""
from __future__ import annotations

import importlib
import os
import sys
import tempfile

import torch

REFERENCE_ROOT = os.environ.get("MAGICDEC_REFERENCE", "/root/reference")
_installed = False


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "Engine"))


def install():
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError(f"reference checkout not found at {REFERENCE_ROOT}")
    from oracle import flashinfer_ref

    link_dir = tempfile.mkdtemp(prefix="magicdec_ref_")
    os.symlink(REFERENCE_ROOT, os.path.join(link_dir, "MagicDec"))
    sys.path.insert(0, link_dir)
    sys.modules["flashinfer"] = flashinfer_ref
    sys.modules["flashinfer.rope"] = flashinfer_ref.rope

    orig_impl = torch.library.impl

    def impl_cpu(qualname, types, *a, **kw):
        if types == "cuda":
            types = "cpu"
        elif isinstance(types, (list, tuple)):
            types = ["cpu" if t == "cuda" else t for t in types]
        return orig_impl(qualname, types, *a, **kw)

    torch.library.impl = impl_cpu
    _installed = True


def module(name: str):
    """e.g. module('Engine.SnapKV.model')"""
    install()
    return importlib.import_module("MagicDec." + name)
