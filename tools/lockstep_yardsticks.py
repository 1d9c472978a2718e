"""CPU only: the flip counts of the three yardstick oracles of tests/test_gpu_engine.py::replay (float64 linears, bf16 P,
both) on the lock-step logs of the basic layouts -- what the HIP engine's count is gated against (flip_gate).
usage: python tools/lockstep_yardsticks.py [longspec_snapkv longspec_stream selfspec_snapkv selfspec_stream cfg4]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import harness_ref as hr  # noqa: E402
from oracle import magicdec_ref as mr  # noqa: E402
from tests import golden_cfg as gc  # noqa: E402
from tests import test_gpu_engine as tg  # noqa: E402


def run(which):
    cfg, sd = gc.tiny("tinytgt")
    log = []
    if which.startswith("longspec"):
        kind = "snapkv_draft" if which.endswith("snapkv") else "stream_draft"
        tgt = tg.Recorder(mr.RefEngine("target", cfg, sd, gc.B, gc.MAX_LEN), "T", log)
        drf = tg.Recorder(mr.RefEngine(kind, cfg, sd, gc.B, gc.MAX_LEN if kind == "snapkv_draft" else 0, gc.BUDGET), "D", log)
        for ids in gc.synthetic_batches()[:tg.N_BATCH]:
            hr.longspec_batch(tgt, drf, ids, gc.GAMMA, gc.MAX_LEN, gc.EOT_1, gc.EOT_2)
        alt = {"T": tg._alt("target", cfg, sd, gc.B, gc.MAX_LEN),
               "D": tg._alt(kind, cfg, sd, gc.B, gc.MAX_LEN if kind == "snapkv_draft" else 0, gc.BUDGET)}
    elif which.startswith("selfspec"):
        kind = "snapkv_self" if which.endswith("snapkv") else "stream_self"
        eng = tg.Recorder(mr.RefEngine(kind, cfg, sd, gc.B, gc.MAX_LEN, gc.BUDGET), "T", log)
        for ids in gc.synthetic_batches()[:tg.N_BATCH]:
            hr.selfspec_batch(eng, ids, gc.GAMMA, gc.MAX_LEN, gc.EOT_1, gc.EOT_2, kind == "stream_self")
        alt = {"T": tg._alt(kind, cfg, sd, gc.B, gc.MAX_LEN, gc.BUDGET)}
    elif which == "cfg4":
        cfg_t, sd_t = tg._extra_ckpt("tiny70b")[:2]
        cfg_d, sd_d = gc.tiny("tinydrf")
        tgt = tg.Recorder(mr.RefEngine("target", cfg_t, sd_t, gc.B, gc.MAX_LEN), "T", log)
        drf = tg.Recorder(mr.RefEngine("stream_draft", cfg_d, sd_d, gc.B, 0, gc.BUDGET), "D", log)
        for ids in gc.synthetic_batches()[:tg.N_BATCH]:
            hr.longspec_batch(tgt, drf, ids, gc.GAMMA, gc.MAX_LEN, gc.EOT_1, gc.EOT_2)
        alt = {"T": tg._alt("target", cfg_t, sd_t, gc.B, gc.MAX_LEN), "D": tg._alt("stream_draft", cfg_d, sd_d, gc.B, 0, gc.BUDGET)}
    else:
        raise KeyError(which)
    st = tg.replay(log, None, alt)
    print(st.line(which + " (yardsticks only)"), flush=True)


if __name__ == "__main__":
    torch.set_num_threads(8)
    for w in (sys.argv[1:] or ["longspec_snapkv", "longspec_stream", "selfspec_snapkv", "selfspec_stream", "cfg4"]):
        run(w)
