"""Worker of tests/test_gpu_engine.py::test_a_failed_graph_capture_falls_back_to_eager_and_keeps_working (own process:
an invalidated capture poisons further captures of the process)."""
import os
import sys
import tempfile
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import golden_cfg as gc                       # noqa: E402


def main(kind):
    from pathlib import Path
    from magicdec_amd.Engine import graph as graph_mod
    from magicdec_amd.Engine import model_core
    from magicdec_amd.Engine.SnapKV.backend import LMBackend
    dev = "cuda:0"
    d = tempfile.mkdtemp(prefix="md_graphfail_")
    cfg, sd = gc.tiny("tinytgt")
    os.makedirs(os.path.join(d, "tinytgt"))
    torch.save(sd, os.path.join(d, "tinytgt", "model.pth"))
    model_core.transformer_configs["tinytgt"] = gc.config_kwargs(cfg)

    def engine():
        e = LMBackend(dtype=torch.bfloat16, device=dev, dec_len=gc.GAMMA + 1)
        e.load_model(Path(d) / "tinytgt" / "model.pth", use_tp=False)
        e.setup_caches(max_batch_size=gc.B, max_seq_length=gc.MAX_LEN)
        return e

    ids = gc.synthetic_batches()[0].to(dev)
    e_ref = engine()
    t_ref = e_ref.encode(ids)
    want = [e_ref.inference(t_ref[:, -1:].clone()).clone() for _ in range(3)]
    e = engine()
    t = e.encode(ids)
    e.compile()
    real = e.model.forward
    state = {"hits": 0}

    def sabotaged(*a, **k):
        if torch.cuda.is_current_stream_capturing():
            state["hits"] += 1
            if kind == "python_error":
                raise RuntimeError("sabotage: an op that cannot be captured")
            torch.cuda.synchronize()              # illegal during capture: invalidates it
        return real(*a, **k)
    e.model.forward = sabotaged
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = [e.inference(t[:, -1:].clone()).clone() for _ in range(3)]
    assert state["hits"] == 1 and e._use_graphs is False, (state, e._use_graphs)
    assert any("continues WITHOUT graphs" in str(x.message) for x in w)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    # another back-end of the same process: after a Python-level failure it captures and replays as usual; after an
    # INVALIDATED capture this PyTorch build refuses further captures (its generator stays "capturing"), so it runs eagerly
    # too -- right results either way, no crash (the failed graph object is never destroyed: its destructor would
    # terminate the process)
    e2 = engine()
    t2 = e2.encode(ids)
    e2.compile()
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        got2 = [e2.inference(t2[:, -1:].clone()).clone() for _ in range(3)]
    assert e2._use_graphs is (kind == "python_error"), e2._use_graphs
    assert (graph_mod._CAPTURE_POISONED[0] is not None) == (kind == "illegal_sync"), graph_mod._CAPTURE_POISONED
    for a, b in zip(got2, want):
        assert torch.equal(a, b)
    import gc as _gc
    _gc.collect()                                     # the leaked graph survives a collection
    print(f"OK eager fall-back reproduces {len(got)} + {len(got2)} steps bit for bit; second back-end graphs="
          f"{e2._use_graphs}; poisoned={graph_mod._CAPTURE_POISONED[0]}")


if __name__ == "__main__":
    main(sys.argv[1])
