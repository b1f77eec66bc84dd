"""bench.py's bookkeeping that needs no GPU: the kernel-name mapping between the engine's profiler groups and rocprofv3's kernel
names, the algorithmic bytes of a step, the collective model — held against the committed rocprofv3 summaries under profiles/."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import bench  # noqa: E402
import rocprof_summary  # noqa: E402


def _kernels(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.load(f)["kernels"]


def _traffic(kernels, prefix):
    hits = [e for k, e in kernels.items() if k.startswith(prefix)]
    assert len(hits) == 1, (prefix, [k for k in kernels if k.startswith(prefix[:12])])
    return hits[0]["fetch_bytes_corrected"] + hits[0]["write_bytes"]


def test_documents_update_maps_to_the_kernel_that_does_the_work():
    """the large-tables leg reported 71 712 B of traffic for a 3.78 GB launch — `row_pass_entities` was mapped to
    table_pass_wide_kernel<4, 1, 3, ...>, which at |D| = 2 M only runs the few chunked rows; the rows are walked by entry_walk_kernel."""
    wl = dict(num_words=500000, num_entities=2000000, word_dim=300, entity_dim=256, window=10, num_random=16)
    ab = bench.algorithmic_bytes("row_pass_entities", wl, "sparse_adam", 51200, {"entities": 705551})
    assert ab == 3781226496
    large = _kernels("r05_large_hbm_pmc.json")
    t = _traffic(large, bench.pmc_prefix("row_pass_entities", "entry_walk"))
    assert 0.5 * ab <= t <= 2.0 * ab, (t, ab)
    assert _traffic(large, bench.pmc_prefix("row_pass_entities", "row_walk")) < 1e6      # (what round 5 reported)
    # the metric's shape: the dense row walk
    wl = dict(wl, num_words=50000, num_entities=100000)
    ab = bench.algorithmic_bytes("row_pass_entities", wl, "sparse_adam", 51200)
    t = _traffic(_kernels("r05_nvsm_hbm_pmc.json"), bench.pmc_prefix("row_pass_entities", "row_walk"))
    assert ab == 1300889600 and 0.5 * ab <= t <= 2.0 * ab


def test_step_bytes_algorithmic_against_counters():
    """roofline_step: Σ algorithmic bytes of the headline step (5.6 GB) against Σ counter bytes of a step in the committed PMC run."""
    wl = dict(num_words=50000, num_entities=100000, word_dim=300, entity_dim=256, window=10, num_random=16, batch_norm=1)
    rs = bench.step_roofline(wl, "sparse_adam", 51200, {}, 0.88)
    assert rs["algorithmic_bytes_per_step"] == sum(rs["algorithmic_bytes_by_kernel"].values())
    assert 5.5e9 < rs["algorithmic_bytes_per_step"] < 5.7e9 and 0.75 < rs["frac"] < 0.85
    per_step, steps = rocprof_summary.step_traffic(_kernels("r05_nvsm_hbm_pmc.json"))
    assert steps > 10 and 0.8 < per_step / rs["algorithmic_bytes_per_step"] < 1.1
    # Adagrad keeps one scalar per row, SGD nothing: their passes read and write the table only
    for method, state in (("sgd", 1), ("adagrad", 1), ("dense_adam", 2), ("full_adam", 3)):
        ab = bench.algorithmic_bytes("row_pass_entities", wl, method, 51200)
        assert ab == 51200 * 17 * 256 * 4 + 2 * 100000 * 256 * 4 * state
        assert "row_pass_words" in bench.step_kernel_groups(method)
    assert bench.step_kernel_groups("sparse_adam")[-3:] == ["row_pass_words_mv", "adam_u_words", "row_pass_words_u"]


def test_collective_model_is_the_stated_formula():
    t1 = bench.collective_model_us(1, 4096, 2.5)
    assert t1 == 2.5                                                           # one rank: the measured floor, nothing on a wire
    t8 = bench.collective_model_us(8, 307200, 2.5)
    wire = 14 / 8 * 307200 / (bench.XGMI_LINK_GBS * bench.XGMI_RING_EFFICIENCY * 1e3)
    assert abs(t8 - (2.5 + 14 * bench.XGMI_HOP_US + wire)) < 1e-9 and 25 < t8 < 40
