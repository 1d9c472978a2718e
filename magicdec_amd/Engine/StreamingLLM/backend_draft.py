"""Engine/StreamingLLM/backend_draft.py of the reference: `LMBackend_Draft(dtype, device)`."""
from ..backend_core import StreamingDraftBackend
from ..utils import load_model_draft_streamingLLM


class LMBackend_Draft(StreamingDraftBackend):
    _loader = staticmethod(load_model_draft_streamingLLM)
