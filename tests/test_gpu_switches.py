"""The engine's orchestration switches change WHEN and HOW work is queued, never what is computed: events carried by
kernel launches vs plain records, intra-device events with / without the system fence, page-locked batches pulled by a
kernel vs copied by the runtime, the streaming decay of the untouched word rows on a side stream vs the main stream, the
panel vs the tiled GEMM... Each variant trains the same few steps in a process of its own (the switches are read once per
process) and must leave bit-identical parameters and costs."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import hashlib, json, sys
import numpy as np
sys.path.insert(0, %(root)r)
import cunvsm_amd as ca
from tests.helpers import PARAMS, gpu_model
shape = sys.argv[1]
if shape == "split":      # tables larger than the batch: touched-row list + streaming pass over the rest (LSE-like)
    spec = dict(num_words=20000, num_entities=30000, word_dim=64, entity_dim=96, window=5, num_random=4, nonlinearity="tanh",
                batch_norm=False, bias_negative_samples=True, update_method="adagrad", **{"lambda": 0.01})
    B = 512
elif shape == "large":    # the headline shape's kernels and stream layout: batch >= 40 960, d_w = 300, d_e = 256 — split-bf16
    # products with the batch-norm backward inside, the dT product on the main stream, both CSR builds on side stream 2
    spec = dict(num_words=3000, num_entities=5000, word_dim=300, entity_dim=256, window=4, num_random=3, nonlinearity="hard_tanh",
                batch_norm=True, bias_negative_samples=False, update_method="sparse_adam", **{"lambda": 0.01})
    B = 40960
elif shape == "lazy":     # lazily decayed sparse-Adam tables (NVSM_LAZY_MIN_MB=0 in the environment): snapshots, stamps, pending decay
    spec = dict(num_words=20000, num_entities=30000, word_dim=64, entity_dim=96, window=5, num_random=4, nonlinearity="hard_tanh",
                batch_norm=True, bias_negative_samples=False, update_method="sparse_adam", **{"lambda": 0.01})
    B = 512
else:                      # batch larger than the tables: dense passes, chunk tree, batch-norm, sparse Adam (NVSM-like)
    spec = dict(num_words=3000, num_entities=5000, word_dim=60, entity_dim=64, window=6, num_random=5, nonlinearity="hard_tanh",
                batch_norm=True, bias_negative_samples=False, update_method="sparse_adam", **{"lambda": 0.01})
    B = 8192
m = gpu_model(spec, B, sampler=ca.SAMPLER_DEVICE)
m.initialize(7)
rs = np.random.RandomState(5)
costs, keep = [], []
for step in range(6):
    words = (rs.zipf(1.3, B * spec["window"]) %% spec["num_words"]).astype(np.int64)
    labels = rs.randint(0, spec["num_entities"], B).astype(np.int64)
    ww = rs.uniform(0.5, 1.5, B * spec["window"]).astype(np.float32)
    iw = rs.uniform(0.5, 1.5, B).astype(np.float32)
    pins = [ca.model.pinned_copy(x) for x in (words, labels, ww, iw)]      # page-locked host batches: the pull / copy path
    keep.append(pins)
    costs.append(m.step(ca.Batch(pins[0].array, pins[1].array, pins[2].array, pins[3].array), 0.01, want_cost=(step %% 2 == 1)))
h = hashlib.sha256()
for p in PARAMS:
    h.update(np.ascontiguousarray(m.get_param(p)).tobytes())
print("RESULT " + json.dumps({"params": h.hexdigest(), "costs": [c for c in costs if c is not None]}))
"""

# the DOCUMENTED switches (tuning.h): read by the shipped library, once per handle
VARIANTS = [
    {},
    {"NVSM_STOP_EVENTS": "0"},
    {"NVSM_HOST_PULL": "0"},
    {"NVSM_ROCTX": "0", "NVSM_POISON": "1"},
    {"NVSM_STOP_EVENTS": "0", "NVSM_HOST_PULL": "0"},
]
# the EXPERIMENT switches: read only by the experiments build (make dbg → libcunvsm_amd_dbg.so, -DNVSM_EXPERIMENTS); the shipped
# library ignores them (test_the_shipped_library_ignores_experiment_switches)
EXP_VARIANTS = [
    {},
    {"NVSM_EVENT_FENCE": "0"},
    {"NVSM_PULL_BLOCKS": "3"},
    {"NVSM_UNTOUCHED_ASIDE": "0"},
    {"NVSM_GEMM_PANEL": "0"},
    {"NVSM_LOSS_PIPE": "1"},
    {"NVSM_LOSS_PIPE": "0"},
    {"NVSM_STOP_EVENTS": "0", "NVSM_EVENT_FENCE": "0", "NVSM_HOST_PULL": "0", "NVSM_UNTOUCHED_ASIDE": "0"},
    # round 5: where the lazy tables' bookkeeping launches sit, the decay of the words rows without entries behind the CSR build or
    # in the update's tail, who cuts T's bf16 planes and who adds up the dT product's slabs
    {"NVSM_EARLY_SNAPSHOT": "0"},
    {"NVSM_STAMP_IN_PROLOGUE": "0"},
    {"NVSM_HOIST_UNTOUCHED": "0"},
    {"NVSM_HOIST_UNTOUCHED": "0", "NVSM_UNTOUCHED_ASIDE": "0"},
    {"NVSM_HOIST_UNTOUCHED": "2"},
    {"NVSM_PLANES_IN_UPDATE": "0", "NVSM_SLAB_SUM_IN_UPDATE": "0"},
    {"NVSM_EARLY_SNAPSHOT": "0", "NVSM_STAMP_IN_PROLOGUE": "0", "NVSM_HOIST_UNTOUCHED": "0", "NVSM_PLANES_IN_UPDATE": "0", "NVSM_SLAB_SUM_IN_UPDATE": "0"},
    # round 6: the word gather-mean inside the forward product's staging (gemm_rsplit.hip GATH) instead of a launch of its own
    {"NVSM_GATHER_FUSE": "1"},
    # ... and the long rows' chunk descriptors written by the bounds kernel instead of by csr_chunk_fill_kernel
    {"NVSM_CSR_FILL_IN_BOUNDS": "1"},
]
DBG_LIB = os.path.join(ROOT, "cunvsm_amd", "libcunvsm_amd_dbg.so")


def _run(shape, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}, shape], env=env, capture_output=True, text=True, cwd=ROOT,
                       timeout=300)
    lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert r.returncode == 0 and lines, (env_extra, r.stdout[-2000:], r.stderr[-2000:])
    return json.loads(lines[-1][len("RESULT "):])


# the large shape: where the CSR builds run, host batches copied or pulled
LARGE_VARIANTS = [
    {},
    {"NVSM_SORT_LAYOUT": "4"},
    {"NVSM_SORT_LAYOUT": "2"},
    {"NVSM_SORT_LAYOUT": "1"},
    {"NVSM_STOP_EVENTS": "0", "NVSM_HOST_PULL": "0"},
    {"NVSM_POISON": "1"},      # (uncleared device buffers start as 0xFF bytes: the dT product's partials, the loss kernel's sums)
]
LARGE_EXP_VARIANTS = [
    {},
    {"NVSM_WORDS_CSR_LATE": "1"},
    {"NVSM_AUX2_PRIO": "2", "NVSM_SPLIT_NT": "1"},
    {"NVSM_SPLIT_FUSE": "0"},
    # the loss kernel with two sets of rows in flight per wave, at the single-set form's examples per wave (20 from batch 40 960 on:
    # left to its own rule the two-set form takes ~380 workgroups, and a wave that sums more examples rounds its fp32 partial sums
    # differently — the last bits of the loss, not the kernel's arithmetic)
    {"NVSM_LOSS_PIPE": "1", "NVSM_LOSS_EPW": "20"},
    {"NVSM_LOSS_PIPE": "0"},
    # (the planes the projection update writes in the large-batch layout against launch_gemm_split_planes, and the
    #  update's own slab sum against launch_splitk_reduce)
    {"NVSM_PLANES_IN_UPDATE": "0", "NVSM_SLAB_SUM_IN_UPDATE": "0"},
]      # (not NVSM_DT_ON_MAIN: on the main stream the split-bf16 product's slabs are added up by launch_splitk_reduce or by the projection
       #  update in another grouping than the tiled product's on side stream 2 — another summation order)


@pytest.mark.parametrize("shape", ["split", "dense", "large"])
def test_orchestration_switches_do_not_change_results(shape):
    variants = LARGE_VARIANTS if shape == "large" else VARIANTS
    base = _run(shape, variants[0])
    assert len(base["costs"]) == 3 and all(c == c for c in base["costs"])
    for v in variants[1:]:
        got = _run(shape, v)
        assert got == base, (shape, v, got, base)


@pytest.mark.skipif(not os.path.exists(DBG_LIB), reason="experiments build absent (make -C cunvsm_amd/csrc dbg)")
@pytest.mark.parametrize("shape", ["split", "dense", "large", "lazy"])
def test_experiment_switches_do_not_change_results(shape):
    """the experiments build reads the A/B switches; they move work between streams and kernels, never the result — and its
    result is the shipped library's"""
    variants = LARGE_EXP_VARIANTS if shape == "large" else EXP_VARIANTS
    base = {"NVSM_LAZY_MIN_MB": "0"} if shape == "lazy" else {}
    shipped = _run(shape, base)
    for v in variants:
        got = _run(shape, dict(v, CUNVSM_AMD_LIB=DBG_LIB, **base))
        assert got == shipped, (shape, v, got, shipped)


def test_the_shipped_library_ignores_experiment_switches():
    """tuning.h: a stray experiment variable in a user's environment cannot change what the shipped library does — nvsm_describe
    lists the switches a handle runs with, and an experiment switch is not among them"""
    code = ("import sys; sys.path.insert(0, %r); import cunvsm_amd as ca; from tests.helpers import gpu_model\n"
            "m = gpu_model(dict(num_words=50, num_entities=60, word_dim=8, entity_dim=8, window=4, num_random=3, update_method='sgd', **{'lambda': 0.0}), 64)\n"
            "print('DESC ' + m.describe())" % ROOT)
    def desc(env_extra):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env_extra), capture_output=True, text=True, cwd=ROOT, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return [l for l in r.stdout.splitlines() if l.startswith("DESC ")][-1]
    assert desc({}).endswith("switches: defaults")
    # behaviour, not just the switch list: NVSM_LOSS_PIPE=1 changes the loss kernel's dispatch in the experiments build (which names the
    # switch) and nothing at all in the shipped library
    exp = {"NVSM_SPLIT_FUSE": "0", "NVSM_DOCS_ON_MAIN": "1", "NVSM_GEMM_PANEL": "0", "NVSM_LOSS_PIPE": "1"}
    assert desc(exp) == desc({})
    if os.path.exists(DBG_LIB):
        dd = desc(dict(exp, CUNVSM_AMD_LIB=DBG_LIB))
        assert "two row sets per wave" in dd and "one row set per wave" in desc({}), dd
        assert "loss_pipe=1" in dd and "docs_on_main=1" in dd and "split_fuse=0" in dd and "(experiments build)" in dd, dd
    d = desc({"NVSM_STOP_EVENTS": "0", "NVSM_GEMM_SPLIT": "9"})
    assert "stop_events=0" in d and "gemm_split=9" in d


def test_describe_names_the_kernels_a_step_takes():
    """nvsm_describe against the profiler's path notes: the dT product runs the split-bf16 split-K kernel from batch 40 960 on
    (on the main stream, both CSR builds on side stream 2), the tiled fp32 kernel below"""
    import numpy as np
    import cunvsm_amd as ca
    from tests.helpers import gpu_model
    spec = dict(num_words=3000, num_entities=5000, word_dim=300, entity_dim=256, window=4, num_random=3, nonlinearity="hard_tanh",
                batch_norm=True, bias_negative_samples=False, update_method="sparse_adam", **{"lambda": 0.01})
    for B, want_dt in ((51200, True), (40960, True), (6400, False), (4096, False)):
        m = gpu_model(spec, B, sampler=ca.SAMPLER_DEVICE)
        m.initialize(3)
        d = m.describe()
        rs = np.random.RandomState(1)
        words = rs.randint(0, spec["num_words"], B * spec["window"]).astype(np.int64)
        labels = rs.randint(0, spec["num_entities"], B).astype(np.int64)
        m.profile_enable(True)
        m.step(ca.Batch(words, labels, np.ones(B * 4, np.float32), np.ones(B, np.float32)), 0.01)
        m.synchronize()
        notes = set(m.profile())
        assert ("dt_split_bf16" in notes) == want_dt, (B, notes)
        assert ("dT gemm_dt " in d) == want_dt and ("on the main stream" in d) == want_dt, d
        # (per-rank batches: the split-bf16 product in workgroups of one wave, gemm_dtw.hip)
        assert ("dt_wave_sized" in notes) == (not want_dt) and ("dT gemm_dtw" in d) == (not want_dt), (B, notes, d)
        # (per-rank batches: the split-bf16 row-panel kernel for both batch-sized products, the batch-norm backward inside the
        #  backward one; the projection update adds up the dT product's slabs in the fused step: no launch_splitk_reduce)
        assert ("forward gemm_split" in d) == (B > 8192) and ("forward gemm_rsplit" in d) == (B <= 8192), d
        assert ("backward gemm_split" in d) == (B > 8192) and ("backward gemm_rsplit" in d) == (B <= 8192), d
        assert "with the batch-norm backward / bias gradient inside" in d, d
        assert "slab_sum_in_update" in notes and "gemm_bwd_T_reduce" not in notes, (B, notes)
        assert "CSR stream layout %d" % (2 if want_dt else 4) in d, d
        assert "loss loss_rows (one row set per wave)" in d, d
    # a documents table beyond the 256 MB Infinity Cache: the loss kernel keeps two sets of rows in flight per wave
    big = gpu_model(dict(spec, num_entities=300000), 2048, sampler=ca.SAMPLER_DEVICE)
    assert "two row sets per wave" in big.describe(), big.describe()


@pytest.mark.gpu
def test_bind_host_thread_narrows_the_mask_to_the_devices_node(monkeypatch):
    """nvsm_bind_host_thread: afterwards the calling thread may run on a subset of the CPUs it had, all of them in the device's
    local_cpulist; NVSM_BIND_HOST=0 leaves the mask alone; a bad ordinal is a status code."""
    import cunvsm_amd as ca
    before = os.sched_getaffinity(0)
    try:
        monkeypatch.setenv("NVSM_BIND_HOST", "0")
        node0 = ca.bind_host_thread(0)
        assert os.sched_getaffinity(0) == before
        monkeypatch.delenv("NVSM_BIND_HOST")
        node = ca.bind_host_thread(0)
        assert node == node0
        after = os.sched_getaffinity(0)
        assert after and after <= before
        if node >= 0:
            import torch
            bus = torch.cuda.get_device_properties(0).pci_bus_id if hasattr(torch.cuda.get_device_properties(0), "pci_bus_id") else None
            lists = [p for p in os.listdir("/sys/bus/pci/devices") if os.path.exists("/sys/bus/pci/devices/%s/local_cpulist" % p)]
            local = set()
            for dev in lists:
                try:
                    if int(open("/sys/bus/pci/devices/%s/numa_node" % dev).read()) != node:
                        continue
                    for item in open("/sys/bus/pci/devices/%s/local_cpulist" % dev).read().strip().split(","):
                        lo, _, hi = item.partition("-")
                        local |= set(range(int(lo), int(hi or lo) + 1))
                    break
                except (OSError, ValueError):
                    continue
            if local:
                assert after <= local, (sorted(after)[:4], sorted(local)[:4])
        with pytest.raises(ca.NvsmError):
            ca.bind_host_thread(10 ** 6)
    finally:
        os.sched_setaffinity(0, before)
