"""CPU-tier dry runs of ``bench.py``'s control flow (no GPU, no measurement): N = 1 directly, N = 2 under
``python -m torch.distributed.run`` with the gloo backend — the way the driver launches the multi-GPU bench, so the first real
8-GPU run cannot die in argument / rendezvous / reduction code.  The transform itself runs on the tests' numpy stand-in."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOAD = "dryrun_wavedec2_db4_L2_6x96x96_f32"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out  # rank 0 prints ONE line
    return json.loads(lines[0])


def _env():
    env = dict(os.environ)
    env.update(MIFWT_BENCH_BACKEND="gloo", MIFWT_BENCH_DEVICE="cpu", PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _check(res, n):
    assert res["n_gpus"] == n and res["steps"] == 2 and res["warmup"] == 1
    assert res["unit"] == "Msamples/s" and res["higher_is_better"] is True and res["scaling"] == "weak"
    assert res["config"]["workload"] == WORKLOAD and res["config"]["parallelism"].startswith(f"batch-sharded x{n}")
    assert res["value"] > 0 and res["ms_per_step"] > 0
    # whole-job aggregate: N shards of 6 x 96 x 96 samples per step
    assert abs(res["value"] - n * 6 * 96 * 96 / (res["ms_per_step"] * 1e-3) / 1e6) <= 0.02 * res["value"] + 0.1
    assert set(res["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}


def test_bench_single_process_dry_run():
    cmd = [sys.executable, os.path.join(ROOT, "tests", "_bench_dryrun.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--spinup-ms", "0", "--workload", WORKLOAD, "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    _check(_json_line(p.stdout), 1)


def test_bench_two_ranks_gloo_dry_run():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_bench_dryrun.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--spinup-ms", "0", "--workload", WORKLOAD]
    p = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    res = _json_line(p.stdout)
    _check(res, 2)
    assert "cpu_baseline" not in res  # N = 1 only
    # the N > 1 line also answers BASELINE's scaling question: the per-GPU shard of the multi-GPU configuration on every rank (here: the
    # dry-run stand-in of it) and the gather of its coefficient lists
    sh = res["config4_shard"]
    assert "error" not in sh, sh
    assert sh["n_gpus"] == 2 and sh["scaling"] == "weak" and sh["global_shape"][0] == 2 * sh["per_gpu_shape"][0]
    assert sh["ms_per_step"] > 0 and sh["Msamples_per_s"] > 0
    g = sh["coefficient_gather"]
    assert g["ms"] > 0 and g["bytes_per_rank"] > 0 and "all_gather" in g["collectives"]


def test_bench_rejects_mismatched_world():
    cmd = [sys.executable, os.path.join(ROOT, "tests", "_bench_dryrun.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--workload", WORKLOAD]
    p = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "torch.distributed.run" in (p.stderr + p.stdout)
