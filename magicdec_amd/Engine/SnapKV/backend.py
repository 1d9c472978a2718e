"""Engine/SnapKV/backend.py of the reference: `LMBackend(dtype, device, dec_len[, draft_dec_len])`."""
from ..backend_core import SnapKVTargetBackend
from ..utils import load_model_snapKV


class LMBackend(SnapKVTargetBackend):
    _loader = staticmethod(load_model_snapKV)
