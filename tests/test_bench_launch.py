"""bench.py's N > 1 entry: `python bench.py --gpus N` must start its own ranks (the driver's N = 1 call form), and the
one JSON line must carry both the weak and the strong (51 200 / N per rank) figures."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 4])
def test_bench_self_launches_its_ranks(n):
    """No launcher, no GPU: bench.py re-executes itself under torch.distributed.run and the ranks meet (gloo)."""
    out = _run(["--gpus", str(n), "--launch-check"], timeout=300)
    assert out == {"launch_check": n, "n_gpus": n}


def test_bench_under_an_external_launcher_does_not_relaunch():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert [json.loads(l) for l in lines] == [{"launch_check": 2, "n_gpus": 2}]


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_reports_weak_and_strong():
    """The whole N = 2 control flow of bench.py on the 1-GPU box (both ranks on device 0, all-reduces through the gloo
    callback transport): self-launch, both legs, one JSON line. Not a measurement."""
    out = _run(["--gpus", "2", "--test-shared-gpu", "--steps", "3", "--warmup", "1", "--batch", "2048", "--num-words", "5000",
                "--num-entities", "4000", "--no-cpu-baseline"])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["weak"]["batch_per_rank"] == 2048 and out["weak"]["global_batch"] == 4096
    assert out["strong"]["batch_per_rank"] == 1024 and out["strong"]["global_batch"] == 2048 and out["strong"]["value"] == out["value"]
    assert out["strong"]["scaling"] == "strong" and out["weak"]["scaling"] == "weak"
    weak = _run(["--gpus", "2", "--test-shared-gpu", "--steps", "3", "--warmup", "1", "--batch", "2048", "--num-words", "5000",
                 "--num-entities", "4000", "--no-cpu-baseline", "--weak-scaling", "--repeats", "1"])
    assert weak["scaling"] == "weak" and weak["config"]["global_batch"] == 4096 and weak["strong"]["batch_per_rank"] == 1024
    assert out["roofline"]["traffic"] is None            # no PMC profile of THIS workload: no traffic claim
    assert "gloo" in out["config"]["collectives"]


@pytest.mark.gpu
def test_bench_single_gpu_line_has_the_contract_fields():
    out = _run(["--steps", "3", "--warmup", "1", "--batch", "4096", "--num-words", "5000", "--num-entities", "4000", "--cpu-steps", "1"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "roofline_gather", "cpu_baseline", "value_readback_every_step", "value_host_batches"):
        assert k in out, k
    assert out["roofline"]["bound"] == "hbm" and 0 < out["roofline"]["frac"] < 1.5
    assert out["roofline_gather"]["bytes_per_window"] == (10 * 300 + 17 * 256) * 4
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["cores"] >= 1
    assert out["value_readback_every_step"] > 0 and out["value_host_batches"] > 0
