"""bench.py's contract (one JSON line from rank 0 with the fields the driver reads), for N = 1 and — control flow only —
for N = 2: two ranks under torch.distributed.run sharing GPU 0 (`--test-shared-gpu`: gloo rendezvous, all-reduces through
the host-callback transport), i.e. the barriers, the max-over-ranks timing and the sharded batches of the multi-GPU run
without the second GPU."""
import json
import os
import socket
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def roof_ok(roof):
    """a traffic figure names where it comes from (this run's own rocprofv3 counter passes, or a committed summary of this workload
    and this kernel source), and counter bytes are never far from the algorithmic bytes of the same launch (a
    mis-mapped kernel name once reported 72 KB for a 3.8 GB launch)"""
    if roof["traffic"] is None:
        return roof["traffic_measured_in_run"] is False
    ab = roof.get("algorithmic_bytes_per_launch")
    return bool(roof["traffic_source"]) and (ab is None or 0.5 * ab <= roof["traffic"] <= 2.0 * ab)


def test_bench_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--cpu-steps", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert REQUIRED <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["scaling"] == "single" and d["dtype"] == "f32" and "bf16 planes" in d["dtype_detail"]
    assert d["value"] > 1e6 and abs(d["value"] - 51200 * 1e3 / d["ms_per_step"]) < 1e-3 * d["value"]
    # the median of `repeats` timed regions of exactly `steps` steps each, all of them in the line
    t = d["timing"]
    assert t["repeats"] == 5 and len(t["ms_per_step_all"]) == 5 and sorted(t["ms_per_step_all"])[2] == d["ms_per_step"]
    # no row-pass figure above what HBM can deliver without being marked as cache-served
    for k, e in d["kernel_breakdown"].items():
        assert e.get("algorithmic_GBps", 0) <= 8000.0 or e.get("served_from_cache"), (k, e)
    assert roof_ok(d["roofline"]) and roof_ok(d["roofline_loss"])
    # the per-rank shapes of the 8-GPU metric, the reference recipe's optimiser and the uniform worst case ride along
    assert set(d["per_rank_shapes"]) == {"6400", "12800", "25600"} and d["per_rank_shapes"]["6400"]["ms_per_step"] > 0
    assert d["strong_projection_8gpu"]["speedup_over_1gpu"] > 1.0
    assert d["secondary"]["full_adam"]["value"] > 1e6 and d["secondary"]["uniform_words"]["value"] > 1e6
    # every BASELINE config that fits one GPU is in the line: configs[4]'s tables (E out of the Infinity Cache: its own loss-kernel
    # roofline) and configs[3]
    lt, ls = d["secondary"]["large_tables"], d["secondary"]["lse_small"]
    assert lt["batch"] == 51200 and lt["ms_per_step"] > d["ms_per_step"] and 0.2 < lt["roofline_loss"]["frac"] < 1.0
    assert ls["batch"] == 4096 and ls["update_method"] == "adagrad" and 0 < ls["ms_per_step"] < 1.0
    # every leg: `roofline` = the documents update (largest kernel by GPU time), `roofline_loss` = the document gather + loss kernel,
    # `roofline_step` = Σ algorithmic (and counter) bytes of the step over the step time; reported traffic within 0.5-2x algorithmic
    for leg in (lt, ls):
        assert leg["roofline"]["kernel"] == "row_pass_entities" and leg["roofline_loss"]["kernel"] == "loss_fused"
        assert roof_ok(leg["roofline"]) and roof_ok(leg["roofline_loss"])
        rs = leg["roofline_step"]
        assert rs["algorithmic_bytes_per_step"] == sum(rs["algorithmic_bytes_by_kernel"].values()) and 0.05 < rs["frac"] < 1.0
        if "counter_bytes" in rs:
            assert 0.5 <= rs["counter_bytes"]["over_algorithmic"] <= 2.0 and rs["counter_bytes"]["frac"] < 1.0
    # the step-level figure of the headline: Σ algorithmic bytes (5.6 GB) and, measured by this run's own rocprofv3 passes, Σ counter bytes
    rs = d["roofline_step"]
    assert 5.3e9 < rs["algorithmic_bytes_per_step"] < 5.9e9 and 0.5 < rs["frac"] < 1.0
    if rs.get("counter_bytes"):
        assert 0.7 <= rs["counter_bytes"]["over_algorithmic"] <= 1.3 and 0.4 < rs["counter_bytes"]["frac"] < 1.0
    # (rocprofv3 is part of the image: the counter passes run, and the line says the traffic was measured in this run)
    import shutil
    if shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3"):
        assert d["roofline"]["traffic_measured_in_run"] is True and d["roofline_loss"]["traffic_measured_in_run"] is True, d["roofline"]
        assert rs["counter_bytes"]["traffic_measured_in_run"] is True and lt["roofline"]["traffic_measured_in_run"] is True
    # the headline step on the exact-fp32 MFMA kernels rides along (what the three-bf16-plane products buy)
    assert 0 < d["value_exact_fp32_gemm"] < d["value"] * 1.02 and d["secondary"]["exact_fp32_gemm"]["ms_per_step"] > 0
    # the dT product is timed by events riding on its own launch: a kernel time (0.10-0.19 ms depending on whether it or the
    # documents pass wins the CUs both want at the same instant — DESIGN 5.5 item 7), not the 0.3 ms a record pair around it on
    # a busy stream once reported; the figure of a pass with no other records rides along
    dt = d["kernel_breakdown"]["gemm_bwd_T"]
    assert not dt.get("overlapped") and 0.03 < dt["avg_ms"] < 0.26 and 0.03 < dt["avg_ms_no_other_records"] < 0.40      # (the racy in-step figure: 0.19-0.28 box to box)
    # ... and `avg_ms` is the kernel ALONE (what the rocprofv3 kernel trace under profiles/ shows): never longer than the racy in-step figure
    assert "alone" in dt["avg_ms_is"] and dt["avg_ms"] <= dt["in_step_event_ms"] * 1.15
    # the documents update — the largest kernel of the step by GPU time — has a roofline entry of its own: algorithmic bytes,
    # time in the step (next to the words chain) and alone, committed counter bytes (or the reason there are none)
    # ... and it IS `roofline`: the kernel with the largest share of GPU time in kernel_breakdown
    ru = d["roofline"]
    assert ru["kernel"] == "row_pass_entities" and ru["bound"] == "hbm" and ru["decay"] == "eager" and ru["walk"] == "row_walk"
    assert ru["algorithmic_bytes_per_launch"] == 51200 * 17 * 256 * 4 + 4 * 100000 * 256 * 4
    assert 0.2 < ru["in_step"]["frac"] <= ru["alone"]["frac"] < 1.0 and ru["frac"] == ru["in_step"]["frac"]
    assert "inside the timed regions" in ru["in_step_timed_by"] and 0.15 < ru["share_of_gpu_time"] < 0.5
    assert d["roofline_update"]["kernel"] == "row_pass_entities" and d["roofline_loss"]["kernel"] == "loss_fused"
    lu = lt["roofline"]
    assert lu["walk"] == "entry_walk" and lu["decay"] == "lazy" and lu["table_rows_visited"] < 2000000 and 0.2 < lu["in_step"]["frac"] < 1.0
    # what a rank of the N-GPU job adds per step: three collectives, their payloads and 1-rank RCCL latencies
    cd = d["config"]["collectives_dp"]
    assert cd["per_step"] == 3 and [c["payload_bytes"] for c in cd["calls"]] == [2 * 256 * 8, (1 + 2 * 256) * 8, 256 * 300 * 4]
    assert all(c["rccl_1rank_latency_us"] > 0 and c["model_8rank_us"] >= c["rccl_1rank_latency_us"] + 14 * cd["model"]["hop_us"] for c in cd["calls"])
    # per-shard batch-norm statistics (and no batch-norm): ONE collective per step, [dT | db | loss hi | loss lo]
    assert cd["per_step_per_shard_batch_norm"] == 1 and cd["call_per_shard_batch_norm"]["payload_bytes"] == (256 * 300 + 256 + 2) * 4
    # the 8-GPU projections carry the modelled collectives (assumptions named in the line), strong and weak, both batch-norm modes
    sp, wp = d["strong_projection_8gpu"], d["weak_projection_8gpu"]
    assert "ASSUMED" in cd["model"]["assumptions"]
    assert sp["with_collectives_model"]["speedup_over_1gpu"] < sp["per_shard_batch_norm_model"]["speedup_over_1gpu"] < sp["speedup_over_1gpu_compute_only"]
    assert sp["speedup_over_1gpu"] == sp["with_collectives_model"]["speedup_over_1gpu"]
    assert 4.0 < wp["synchronised_batch_norm"]["speedup_over_1gpu"] < wp["per_shard_batch_norm"]["speedup_over_1gpu"] < 8.0
    # BASELINE configs[0] through the trainer CLI
    cf = d["secondary"]["lse_cranfield_cli"]
    assert cf["unit"] == "batches/s" and cf["value"] > 100 and cf["cost_first_last"][1] < cf["cost_first_last"][0]
    # `value` is the last epoch alone; the reference's cumulative log figure and the one-off set-up are reported beside it
    assert cf["value"] >= 0.8 * cf["cumulative_batches_per_s"] and cf["one_off_setup_s"] >= 0 and cf["training_loop_s"] > 0
    # (the slab sum of the dT product happens inside the projection update in the fused step: no reduce launch of its own)
    assert "gemm_bwd_T_reduce" not in d["kernel_breakdown"] and d["kernel_breakdown"]["slab_sum_in_update"]["note"]
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and 0.3 < roof["frac"] < 1.0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "windows/s"
    assert "-O3" in cb["compile_flags"] and "x86-64-v3" in cb["sample"]
    assert "workload" in d["config"] and "model" not in d["config"]


def test_bench_two_ranks_control_flow():
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--test-shared-gpu", "--exact-tables-leg"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["exact_tables"]["batch_per_rank"] == 25600 and d["exact_tables"]["value"] > 0      # (opt-in leg: dp_exact_tables)
    # the metric says batch = 51 200: the headline is the strong split (25 600 windows per rank), the weak figure rides along
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 51200 and d["config"]["batch_per_rank"] == 25600
    assert d["scaling"] == "strong" and d["config"]["parallelism"] == "dp2"
    assert d["cpu_baseline"] is None
    assert abs(d["value"] - 51200 * 1e3 / d["ms_per_step"]) < 1e-3 * d["value"]
    assert d["strong"]["value"] == d["value"] and d["weak"]["global_batch"] == 2 * 51200 and d["weak"]["batch_per_rank"] == 51200
    assert d["final_cost"] == d["final_cost"] and d["final_cost"] > 0          # finite global loss
    # the N-rank line carries: the transport and the communicator's rank count, both scaling figures, and the per-shard batch-norm
    # variant (ONE collective per step) beside the synchronised one (three)
    assert "comm_ranks" in d["config"] and "gloo" in d["config"]["collectives"] and d["config"]["collectives_per_step"] == 3
    ps = d["per_shard_batch_norm"]
    assert ps["collectives_per_step"] == 1 and ps["batch_per_rank"] == 25600 and ps["value"] > 0 and ps["scaling"] == "strong"
