"""Data parallelism over REAL RCCL: one rank per device, the engine's own communicator (nvsm_comm_init), no host callback.

Every test here needs at least two GPUs and is skipped — with that reason — on a 1-GPU box, where RCCL refuses a second rank on
the same device ("Duplicate GPU detected", tools/exp/rccl_same_gpu.py) and the N-rank control flow is covered through gloo instead
(tests/test_dp_gloo.py, the very same workers with transport="gloo"). On a multi-GPU box they are what proves the collectives of a
step: the two f64 all-reduces of the batch-norm statistics on the main stream, the f32 all-reduce of the projection gradient on
side stream 2 (or all three on the main stream: NVSM_DP_T_ON_MAIN=1), and the byte all-gathers of dp_exact_tables.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import ROOT
from tests import test_dp_gloo as G


def _gpus():
    try:
        import cunvsm_amd as ca
        return ca.device_count()
    except Exception:
        return 0


NEEDS_TWO = "real RCCL needs >= 2 GPUs (one rank per device); this box has fewer: the gloo-transport twins in test_dp_gloo.py cover the control flow"
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(_gpus() < 2, reason=NEEDS_TWO)]


@pytest.mark.parametrize("spec,B", [(G.SPEC, 256), (G.SPEC_NOBN, 256), (G.SPEC_WIDE, 2 * 8704)], ids=["bn", "nobn", "wide"])
def test_rccl_dp_hip_equals_single_gpu(spec, B, tmp_path):
    """loss, dense gradients and the phrase gradient of two RCCL ranks = the single-GPU values on the whole batch"""
    import torch.multiprocessing as mp
    import cunvsm_amd as ca
    from tests.helpers import gpu_model, load_params, rel_err
    port = G._free_port()
    mp.spawn(G._worker_gpu, args=(port, spec, B, str(tmp_path), "rccl"), nprocs=G.WORLD, join=True)
    params, (words, ww, labels, iw, ids) = G._global_problem(spec, B, 7)
    ref = gpu_model(spec, B)
    load_params(ref, params, True)
    ref.compute_cost(ca.Batch(words, labels, ww, iw), ids)
    ref.compute_gradients()
    r = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % k)) for k in range(G.WORLD)]
    for k in range(G.WORLD):
        assert abs(r[k]["cost"] - ref.get_cost()) <= 1e-5 * abs(ref.get_cost())
        assert rel_err(r[k]["gT"], ref.get_tensor("grad_transform")) < 1e-5
        assert rel_err(r[k]["gb"], ref.get_tensor("grad_bias")) < 1e-5
    assert rel_err(np.concatenate([r[0]["gphrase"], r[1]["gphrase"]]), ref.get_tensor("grad_phrase")) < 1e-5


@pytest.mark.parametrize("collectives", ["two_streams", "main_stream"])
@pytest.mark.parametrize("method,wide", [("sgd", False), ("sparse_adam", False), ("sparse_adam", True)], ids=["sgd", "sparse_adam", "sparse_adam_wide"])
def test_rccl_dp_fused_step(tmp_path, method, wide, collectives, monkeypatch):
    """nvsm_step over RCCL in both collective orders: replicas of the dense parameters bit-identical, the first step = the
    single-GPU step on the whole batch, the loss = the global loss"""
    import torch.multiprocessing as mp
    import cunvsm_amd as ca
    from tests.helpers import gpu_model, load_params
    if collectives == "main_stream":
        monkeypatch.setenv("NVSM_DP_T_ON_MAIN", "1")
    spec = dict(G.SPEC_WIDE if wide else G.SPEC, update_method=method)
    B = 2 * 8704 if wide else 256
    port = G._free_port()
    mp.spawn(G._worker_gpu_step, args=(port, spec, B, str(tmp_path), False, "rccl"), nprocs=G.WORLD, join=True)
    params, (words, ww, labels, iw, ids) = G._global_problem(spec, B, 7)
    ref = gpu_model(spec, B)
    load_params(ref, params, True)
    c0 = ref.step(ca.Batch(words, labels, ww, iw), 0.05, entity_ids=ids, want_cost=True)
    T1 = ref.get_param("word_entity_mapping-transform")
    r = [np.load(os.path.join(str(tmp_path), "step_rank%d.npz" % k)) for k in range(G.WORLD)]
    assert abs(r[0]["cost"][0] - c0) <= 1e-5 * abs(c0) and abs(r[1]["cost"][0] - c0) <= 1e-5 * abs(c0)
    np.testing.assert_array_equal(r[0]["T"], r[1]["T"])
    np.testing.assert_array_equal(r[0]["b"], r[1]["b"])
    np.testing.assert_array_equal(r[0]["T1"], r[1]["T1"])
    step = np.linalg.norm(T1 - params["word_entity_mapping-transform"])
    assert np.linalg.norm(r[0]["T1"] - T1) <= (2e-2 if method.endswith("adam") else 1e-4) * step
    assert not np.array_equal(r[0]["E"], r[1]["E"])          # rank-local tables


@pytest.mark.parametrize("method", ["sgd", "sparse_adam"])
def test_rccl_dp_exact_tables_hip(tmp_path, method):
    """dp_exact_tables over RCCL (ncclAllGather of the updates' inputs): 20 steps, tables bit-identical across the ranks and
    equal to one handle's on the whole batches"""
    import torch.multiprocessing as mp
    import cunvsm_amd as ca
    from tests.helpers import gpu_model, load_params
    spec = dict(G.TRAJ_SPEC, update_method=method)
    lr = G.EXACT_LR[method]
    port = G._free_port()
    mp.spawn(G._worker_traj, args=(port, spec, str(tmp_path), True, True, lr, False, "rccl"), nprocs=G.WORLD, join=True)
    params, batches = G._traj_batches(spec)
    ref = gpu_model(spec, G.TRAJ_B)
    load_params(ref, params, True)
    costs = [ref.step(ca.Batch(words, labels, ww, iw), lr, entity_ids=ids, want_cost=True) for words, ww, labels, iw, ids in batches]
    r = [np.load(os.path.join(str(tmp_path), "traj_rank%d.npz" % k)) for k in range(G.WORLD)]
    for name in ("E", "W", "T", "cost"):
        np.testing.assert_array_equal(r[0][name], r[1][name])
    np.testing.assert_allclose(r[0]["cost"], costs, rtol=2e-4)
    for name, pname in (("E", "entity_representations-representations"), ("W", "word_representations-representations"),
                        ("T", "word_entity_mapping-transform")):
        single = ref.get_param(pname).astype(np.float64).ravel()
        moved = np.linalg.norm(single - params[pname].astype(np.float64).ravel())
        assert np.linalg.norm(r[0][name].astype(np.float64).ravel() - single) <= 2e-5 * moved, name


def test_rccl_bench_two_gpus():
    """python bench.py --gpus 2: the line says the collectives ran on the engine's RCCL communicator with two ranks"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-extra-legs"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["collectives"] == "rccl" and d["config"]["comm_ranks"] == 2
    assert d["config"]["collectives_per_step"] == 3 and d["scaling"] == "strong"
    assert d["final_cost"] == d["final_cost"] and d["final_cost"] > 0
