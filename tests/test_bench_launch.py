"""CPU: `bench.py --gpus 2` launches itself as one process per rank when no launcher set the environment
(torch.distributed.run, 127.0.0.1 rendezvous), runs the barrier-bracketed timing skeleton over gloo and prints ONE JSON
line from rank 0 with the contract's keys.  --dry-run replaces the device work (there is no CPU path to bench)."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launch_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3",
                          "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["ms_per_step"] >= 2.0       # MAX over ranks: rank 1 sleeps 2 ms per step


SUB_KEYS = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "dtype", "config", "roofline")
ROOF_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic")


def _check_all_schema(d, world):
    """the schema of the default (--mode all) line: the query record with the fit and train records inside"""
    for k in ROOF_KEYS:
        assert k in d["roofline"], k
    assert "query_fwd_bwd_points_per_s" in d
    for name in ("fit", "train"):
        sub = d[name]
        for k in SUB_KEYS:
            assert k in sub, (name, k)
        for k in ROOF_KEYS:
            assert k in sub["roofline"], (name, k)
        if world == 1:
            for k in ("value", "unit", "cores", "kind", "sample"):
                assert k in sub["cpu_baseline"], (name, k)
    if world > 1:       # a SCALE run measures a collective: the training record carries the all-reduce share
        for k in ("ms_per_step_synced", "ms_per_step_no_sync", "share_of_step"):
            assert k in d["train"]["allreduce"], k


def test_default_mode_schema_one_and_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    for world in (1, 2):
        out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(world), "--dry-run", "--steps", "2",
                              "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env, cwd=REPO)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout
        _check_all_schema(json.loads(lines[0]), world)


def test_stdout_line_is_bounded():
    """round 5's 25 KB line left the driver's BENCH record unparsed: whatever the run measured, the stdout line stays below
    8 KB (bench.LINE_LIMIT) and keeps the contract's keys; the rest goes to bench_detail.json"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    long = "x" * 5000
    fat = {k: long for k in ("a", "b", "c")}
    roof = {"kernel": long, "bound": "mfma", "achieved": 1.0 / 3, "peak": 2500.0, "unit": "TFLOP/s", "frac": 1.0 / 7500, "traffic": 1.5e8,
            "avg_launch_ms": 0.1, "note": long}
    cpu = {"value": 1.0 / 3, "unit": "points/s", "cores": 128, "kind": "port", "sample": long, "host_cpus": long}
    sub = {"metric": long, "value": 1.0 / 3, "unit": "ms", "steps": 5, "warmup": 1, "ms_per_step": 100.0 / 3, "higher_is_better": False,
           "dtype": "fp16x3", "config": {"workload": long, "terms": long}, "roofline": dict(roof), "cpu_baseline": dict(cpu),
           "per_phase": fat, "loader_loop": fat, "other_modes": fat, "allreduce": {"ms_per_step_synced": 1.0, "ms_per_step_no_sync": 1.0,
                                                                                  "share_of_step": 0.1, "note": long}}
    out = {"metric": long, "value": 1.0 / 3, "unit": "points/s", "n_gpus": 8, "steps": 20, "warmup": 5, "ms_per_step": 14.0 / 3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp16x3", "data": "synthetic",
           "config": {"workload": long, "precision": long, "field_err": fat}, "roofline": dict(roof), "cpu_baseline": dict(cpu),
           "kernels": fat, "graph_replay": fat, "other_modes": fat, "records": fat, "query_fwd_bwd_points_per_s": 1.0 / 3,
           "train": dict(sub), "fit": dict(sub), "fit_fp16_fields": dict(sub), "records_aborted": {"stage": long, "after_s": 1.0, "why": long}}
    import io
    buf = io.StringIO()
    cwd = os.getcwd()
    bench.emit_line(out, buf)
    text = buf.getvalue()
    assert text.count("\n") == 1 and len(text) < 8192 and len(text) <= bench.LINE_LIMIT + 1, len(text)
    d = json.loads(text)
    _check_all_schema(d, 1)
    _check_all_schema(d, 2)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config"):
        assert k in d, k
    assert d["config"]["workload"] and d["cpu_baseline"]["sample"]
    full = json.load(open(os.path.join(REPO, "bench_detail.json")))
    assert full["kernels"] == fat
    os.remove(os.path.join(REPO, "bench_detail.json"))
