"""bench.py's control flow (single rank and torchrun-style 2 ranks over gloo) exercised on the CPU with the device ops
replaced by oracle stand-ins: catches crashes in the multi-GPU path (draft sub-group, token broadcast, TP loaders,
max-over-ranks timing, JSON contract) that cannot be run on the 1-GPU development boxes."""
import json
import os
import subprocess
import sys
import tempfile
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]

WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["MD_ROOT"])
from tests import cpu_ops
cpu_ops.install()
import bench
sys.argv = ["bench.py", "--gpus", os.environ["WORLD_SIZE"], "--steps", "6", "--warmup", "2",
            "--workload", os.environ.get("MD_WORKLOAD", "tiny"),
            "--no-cpu-baseline", "--draft-tp", os.environ["MD_DRAFT_TP"]]
args = bench.parse()
line = bench.run(args, "cpu")
if line is not None:
    json.dump(line, open(os.environ["MD_OUT"], "w"))
'''


@pytest.mark.parametrize("world,draft_tp,workload", [(1, 4, "tiny"), (2, 4, "tiny"), (2, 1, "tiny"),
                                                    (1, 4, "tiny-selfspec-snapkv"), (2, 1, "tiny-longspec-stream"),
                                                    (4, 2, "tiny-kh8"), (8, 4, "tiny-kh8")])
def test_bench_control_flow_on_cpu(world, draft_tp, workload):
    """(2,1): the draft runs on rank 0 only -> rank 1 has no draft model and receives the tokens by broadcast, the
    8-GPU layout of the reference's README (target TP8, draft TP4) in miniature.  (4,2) and (8,4) on "tiny-kh8" (eight
    kv heads) ARE that layout at 4 and 8 ranks: one kv head per rank at TP8, the draft on the first half of the ranks,
    the other half idle during drafting and fed by the token broadcast -- the exact launch the driver's scaling run
    makes (`--gpus 4` / `--gpus 8`), over gloo."""
    out = tempfile.mkdtemp(prefix="md_bench_")
    script = os.path.join(out, "w.py")
    Path(script).write_text(WORKER)
    port = 29700 + (os.getpid() % 1500) + world * 3 + draft_tp + 11 * len(workload)
    procs = []
    for r in range(world):
        env = dict(os.environ, LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(world), RANK=str(r), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MD_ROOT=str(ROOT), MD_OUT=os.path.join(out, "line.json"), MD_DRAFT_TP=str(draft_tp),
                   MD_WORKLOAD=workload, OMP_NUM_THREADS="2" if world <= 2 else "1")
        procs.append(subprocess.Popen([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      cwd=out))
    logs = [p.communicate(timeout=1200)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    line = json.load(open(os.path.join(out, "line.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "speedup_vs_autoregressive"):
        assert key in line, key
    assert line["n_gpus"] == world and line["steps"] == 6 and line["value"] > 0
    assert line["unit"] == "tokens/s" and line["higher_is_better"] is True and line["vs_baseline"] is None
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    # the early restore that keeps a SnapKV draft inside its compressed cache's last page (bench.run: draft_step_cap) is
    # decided by rank-independent arithmetic and exists only for longspec + SnapKV draft; these short runs never need it
    early = line["config"]["draft_cache_restores_before_its_page_end"]
    assert early == (0 if workload in ("tiny", "tiny-kh8") else None), (workload, early)


SELF_LAUNCH_WRAPPER = r'''
import os, sys
sys.path.insert(0, os.environ["MD_ROOT"])
from tests import cpu_ops
cpu_ops.install()                 # every process of the job -- the launcher re-executes THIS script per rank
import bench
bench.main(device="cpu")
'''


def test_plain_bench_command_with_gpus_2_launches_itself():
    """VERDICT r5 missing #2: `python bench.py --gpus N` started PLAINLY (no torchrun, WORLD_SIZE unset) must not die on
    an assert: bench.main re-executes its own command line under `torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1` (the reference's scripts are torchrun-launched, README.md:59-69) and rank 0 prints the JSON
    line.  Driven here over gloo through a wrapper that installs the device-op stand-ins in every rank and calls
    bench.main(device="cpu") -- the launcher, the rendezvous, run() and the line contract are the production code."""
    out = tempfile.mkdtemp(prefix="md_bench_self_")
    script = os.path.join(out, "bench_cpu.py")
    Path(script).write_text(SELF_LAUNCH_WRAPPER)
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MD_ROOT=str(ROOT), OMP_NUM_THREADS="2")
    p = subprocess.run([sys.executable, script, "--gpus", "2", "--workload", "tiny", "--steps", "4", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=out, timeout=1200)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout.decode()[-2000:] + p.stderr.decode()[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["collective_backend"] == "gloo"
    assert line["steps"] == 4 and line["value"] > 0 and "TP2" in line["config"]["workload"]
    assert "launching 2 ranks" in p.stderr.decode()


def test_collective_report_runs_in_child_processes():
    """bench.collective_microbench_isolated: every rank starts tools/collective_bench.py with a rendezvous port of its
    own, waits on the host and rank 0 returns the children's report; a child that never completes (here: its peer is
    missing) is killed after the time-out and reported, the caller carries on.  (--dry: rendezvous + barrier over gloo.)"""
    import json
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import json, sys; sys.path.insert(0, %r); import bench; "
            "print('RESULT ' + json.dumps(bench.collective_microbench_isolated([('probe', 8, 16)], iters=2, "
            "timeout_s=float(sys.argv[1]), dry=True)))" % root)
    port = 23000 + os.getpid() % 4000

    def parent(rank, world, timeout):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2",
                   TORCHELASTIC_USE_AGENT_STORE="True")      # what torchrun exports; the children must not inherit it
        return subprocess.Popen([sys.executable, "-c", code, str(timeout)], env=env, stdout=subprocess.PIPE,
                                stderr=subprocess.STDOUT, cwd=tempfile.gettempdir())
    procs = [parent(r, 2, 120) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    res = [json.loads(o.split("RESULT ", 1)[1].splitlines()[0]) for o in outs]
    assert res[0] == {"dry": True, "world": 2, "shapes": ["probe"]} and res[1] is None
    # a lone rank 0 of a 2-rank job: its child blocks in the rendezvous, is killed, and the failure is the report
    port += 1
    p = parent(0, 2, 8)
    out = p.communicate(timeout=300)[0].decode()
    assert p.returncode == 0, out
    r = json.loads(out.split("RESULT ", 1)[1].splitlines()[0])
    assert "timed out" in r["error"], r


def test_allreduce_selection_from_the_collectives_report():
    """bench.xgmi_verdict: a multi-GPU run takes the xGMI fused all-reduce only when the child processes validated it on
    every message (no failure, no time-out, bit-exact stress) and measured it faster than RCCL + the add+norm launch on the
    verify message; when the WRITE-THROUGH publish fails its stress, the release-fence arm (MAGICDEC_AR_PUBLISH=fence) is
    still eligible on its own stress and timing -- the run degrades to a slower xGMI path, not to RCCL."""
    import bench
    ok = {"xgmi_fused_add_rmsnorm_auto": 11.0, "xgmi_fence_fused_add_rmsnorm_auto": 17.0,
          "rccl_allreduce_then_add_rmsnorm": 31.0, "xgmi_timeouts": 0}
    st = {"calls": 640, "mismatched_elements_all_ranks": 0, "timeouts_all_ranks": 0}
    stf = {"calls": 320, "mismatched_elements_all_ranks": 0, "timeouts_all_ranks": 0}
    full = {"verify": ok, "draft_step": dict(ok), "xgmi_stress": st, "xgmi_fence_stress": stf}
    assert bench.xgmi_verdict(full)[0] == "wt"
    assert bench.xgmi_verdict({"verify": ok})[0] is None                                                   # no stress report
    # stale bits through the write-through hand-off: the fence arm takes over ...
    arm, why = bench.xgmi_verdict(dict(full, xgmi_stress=dict(st, mismatched_elements_all_ranks=3)))
    assert arm == "fence" and "fence-publish arm" in why, why
    # ... unless it failed too, is slower than RCCL, or the first stress left the sticky status word set
    assert bench.xgmi_verdict(dict(full, xgmi_stress=dict(st, mismatched_elements_all_ranks=3),
                                   xgmi_fence_stress=dict(stf, mismatched_elements_all_ranks=1)))[0] is None
    assert bench.xgmi_verdict(dict(full, verify=dict(ok, xgmi_fence_fused_add_rmsnorm_auto=40.0),
                                   xgmi_stress=dict(st, mismatched_elements_all_ranks=3)))[0] is None
    assert bench.xgmi_verdict(dict(full, xgmi_stress=dict(st, timeouts_all_ranks=1)))[0] is None
    # write-through correct but slower than RCCL: the (slower still) fence arm cannot win either
    assert bench.xgmi_verdict(dict(full, verify=dict(ok, xgmi_fused_add_rmsnorm_auto=40.0,
                                                     xgmi_fence_fused_add_rmsnorm_auto=45.0)))[0] is None
    assert bench.xgmi_verdict(dict(full, verify=dict(ok, xgmi_timeouts=1)))[0] == "fence"      # a wt spin timed out
    assert bench.xgmi_verdict(dict(full, draft_step={"xgmi": "unavailable"}))[0] is None
    assert bench.xgmi_verdict({"error": "child timed out after 120 s"})[0] is None
    assert bench.xgmi_verdict(dict(full, rank0_child="child exited with -11"))[0] is None
    assert bench.xgmi_verdict(None)[0] is None and bench.xgmi_verdict({"autoregressive": ok, "xgmi_stress": st})[0] is None


def test_speedup_condition_states_what_the_headline_is_conditional_on():
    """bench.speedup_condition: break-even and the north-star threshold interpolated from the alpha
    sweep; says so when a threshold is never reached or already met at the lowest swept rate."""
    import bench
    s = bench.speedup_condition({0.5: 1.43, 0.6: 1.73, 0.7: 1.98, 0.8: 2.29, 0.9: 2.73}, 0.8)
    assert "alpha=0.8" in s and ">= 1.0x already at alpha = 0.5" in s and ">= 1.8x iff alpha >= 0.63" in s, s
    s = bench.speedup_condition({0.5: 0.72, 0.6: 0.82, 0.7: 0.98, 0.8: 1.12, 0.9: 1.33}, 0.8)
    assert ">= 1.0x iff alpha >= 0.71" in s and "1.8x is not reached" in s and "1.33x at 0.9" in s, s
    assert "no alpha sweep" in bench.speedup_condition({0.8: 2.2}, 0.8)


def test_peaked_synthetic_weights_predict_through_one_permutation(monkeypatch):
    """Engine/utils._peak_ (bench.py --weights peaked): the seeded layers with a dominant embedding and a head tied to it
    through a permutation that depends on (seed, vocab) only -- the next-token logit of the tied row stands far above
    the rest, and two models of one vocabulary (different widths) predict through the same map."""
    import torch
    from magicdec_amd.Engine import model_core, utils
    monkeypatch.setenv("MAGICDEC_SYNTH_WEIGHTS", "peaked")
    maps = []
    for name, dim in (("peak_a", 256), ("peak_b", 128)):
        model_core.transformer_configs[name] = dict(block_size=2048, n_layer=1, n_head=4, n_local_heads=2, dim=dim,
                                                    intermediate_size=2 * dim, vocab_size=1000)
        with torch.device("meta"):
            m = model_core.Transformer.from_name(name)
        utils._random_init_(m, 1234, "cpu", torch.bfloat16)
        e, o = m.tok_embeddings.weight.float(), m.output.weight.float()
        assert 30.0 < float(e.std()) < 50.0                                  # the embedding dominates: rms ~ 40 per element
        lg = o @ (e / e.pow(2).mean(dim=1, keepdim=True).sqrt()).t()          # [vocab out, vocab in]: logits of every token
        top2 = lg.topk(2, dim=0).values
        assert float((top2[0] - top2[1]).min()) > 4.0                        # peaked: the tied row wins by a wide margin
        maps.append(lg.argmax(dim=0))
    assert torch.equal(maps[0], maps[1])                                     # the same permutation at both widths
    assert len(set(maps[0].tolist())) == 1000 and maps[0][:4].tolist() == [0, 1, 2, 3]
    monkeypatch.setenv("MAGICDEC_SYNTH_WEIGHTS", "bogus")
    import pytest
    try:
        with pytest.raises(ValueError):
            utils._random_init_(m, 1234, "cpu", torch.bfloat16)
    finally:
        for name in ("peak_a", "peak_b"):            # the table is process-global: other tests compare it with the reference's
            model_core.transformer_configs.pop(name, None)


def test_peaked_draft_with_a_miss_fraction_and_in_place_repeak(monkeypatch):
    """bench.py --weights peaked:...:miss=f (round 6): a DRAFT model's head mispredicts a seeded fraction
    f of the vocabulary (Engine/utils._peak_perm), the target's does not; repeak_head_ rewrites the head for another
    fraction in place (same storage) so that bench.py can sweep the draft's acceptance rate without reloading; and the
    acceptance bookkeeping of the line (alpha_of, acceptance_point) inverts the truncated-geometric expectation."""
    import torch
    import bench
    from magicdec_amd.Engine import model_core, utils
    assert utils.parse_peaked("peaked") == (40.0, 12.0, 0.0)
    assert utils.parse_peaked("peaked:30:10:miss=0.25") == (30.0, 10.0, 0.25)
    assert utils.parse_peaked("peaked:miss=0.1") == (40.0, 12.0, 0.1)
    with pytest.raises(ValueError):
        utils.parse_peaked("peaked:1:2:3")
    monkeypatch.setenv("MAGICDEC_SYNTH_WEIGHTS", "peaked:40:12:miss=0.3")
    name = "peak_miss"
    model_core.transformer_configs[name] = dict(block_size=2048, n_layer=1, n_head=4, n_local_heads=2, dim=128,
                                                intermediate_size=256, vocab_size=2000)
    try:
        heads = {}
        for is_draft in (False, True):
            with torch.device("meta"):
                m = model_core.Transformer.from_name(name)
            utils._random_init_(m, 1234, "cpu", torch.bfloat16, is_draft=is_draft)
            heads[is_draft] = m.output.weight.detach().clone()
        differ = (heads[True] != heads[False]).any(dim=1)
        assert not bool(differ[:4].any())                                   # ids 0..3 stay fixed
        frac = float(differ.float().mean())
        assert 0.25 < frac < 0.35, frac                                     # ~30 % of the draft's rows answer to a wrong token
        # every mis-tied row of the draft is SOME row of the target (a confident wrong prediction, not noise)
        tgt_rows = {tuple(r.view(torch.int16).tolist()) for r in heads[False]}
        assert all(tuple(r.view(torch.int16).tolist()) in tgt_rows for r in heads[True][differ][:50])
        ptr = m.output.weight.data_ptr()
        utils.repeak_head_(m, 0.0)                                          # in place: miss = 0 gives the target's head
        assert m.output.weight.data_ptr() == ptr and torch.equal(m.output.weight, heads[False])
        utils.repeak_head_(m, 0.3)
        assert torch.equal(m.output.weight, heads[True])
    finally:
        model_core.transformer_configs.pop(name, None)
    for a in (0.3, 0.6, 0.8, 0.95):
        t = sum(a ** j for j in range(4))
        assert abs(bench.alpha_of(t, 3) - a) < 1e-6
    pt = bench.acceptance_point(0.030, 64 * 2.9, 64, 3, 2700.0, (0.0301, 64 * 2.95))
    assert pt["tokens_per_iter_per_seq"] == 2.9 and abs(pt["ms_per_step_vs_replay"] - 0.9967) < 1e-3
    assert abs(pt["alpha_equivalent"] - bench.alpha_of(2.9, 3)) < 1e-4 and pt["replay_at_matching_alpha"]["ms_per_step"] == 30.1
