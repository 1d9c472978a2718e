"""Engine/StreamingLLM/backend.py of the reference: self-speculation `LMBackend(dtype, device, dec_len)`."""
from ..backend_core import StreamingSelfSpecBackend
from ..utils import load_model_streamingLLM


class LMBackend(StreamingSelfSpecBackend):
    _loader = staticmethod(load_model_streamingLLM)
