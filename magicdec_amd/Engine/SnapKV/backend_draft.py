"""Engine/SnapKV/backend_draft.py of the reference: `LMBackend_Draft(dtype, device, dec_len, draft_budget)`."""
from ..backend_core import SnapKVDraftBackend
from ..utils import load_model_draft_snapKV


class LMBackend_Draft(SnapKVDraftBackend):
    _loader = staticmethod(load_model_draft_snapKV)
