"""world_size-2 data-parallel tests.

CPU (gloo): the sharding rule + the three reductions of cunvsm_amd.dp (sync batch-norm statistics, batch-norm
backward statistics, dense projection gradient) reproduce the single-process gradients and loss EXACTLY when
the per-rank arithmetic is the fp64 oracle — i.e. the data-parallel algorithm is the same maths as one GPU.
GPU (-m gpu): the same check with the HIP path on both ranks (two processes sharing GPU 0, gloo as the
transport through nvsm_set_allreduce_callback), against the single-process HIP path.
"""
import os
import socket
import sys

import numpy as np
import pytest

from tests.conftest import ROOT

WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _global_problem(spec, B, seed):
    from tests.helpers import random_batch, random_params
    rs = np.random.RandomState(seed)
    params = random_params(spec, rs)
    return params, random_batch(spec, rs, B, zipf=True)


SPEC = dict(num_words=60, num_entities=40, word_dim=12, entity_dim=8, window=3, num_random=4,
            nonlinearity="hard_tanh", batch_norm=True, update_method="sgd")
SPEC["lambda"] = 0.01
SPEC_NOBN = dict(SPEC, batch_norm=False, nonlinearity="tanh", bias_negative_samples=True)
# the metric's dimensions at a per-rank batch above 8 192 rows: the split-bf16 projection kernels with the (synchronised)
# batch-norm backward inside the backward product, under data parallelism
SPEC_WIDE = dict(SPEC, num_words=2000, num_entities=3000, word_dim=300, entity_dim=256, window=4, num_random=3)


def _connect(m, dist, rank, transport):
    """the handle's collectives: gloo through the host callback (ranks share GPU 0), or the engine's own RCCL communicator (one
    rank per device: tests/test_dp_rccl.py)"""
    from cunvsm_amd import dp
    if transport == "rccl":
        dp.init_rccl(m, dist, rank)
        assert m.comm_size() == WORLD
    else:
        m.set_allreduce_callback(dp.torch_allreduce(dist))


def _worker_oracle(rank, port, spec, B, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from cunvsm_amd import dp
    from tests.helpers import load_params, oracle_model
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=WORLD)
    params, (words, ww, labels, iw, ids) = _global_problem(spec, B, 7)
    w, wl, wwt, wi, wid = dp.shard_batch(words, labels, ww, iw, ids, spec["window"], spec["num_random"], rank, WORLD)
    m = oracle_model(spec)
    load_params(m, params, False)
    m.set_allreduce(dp.torch_allreduce(dist), WORLD)
    m.forward(w, wwt, wid, wi)
    m.backward()
    cost = m.get_cost()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), cost=cost, gT=m.get("grad_transform"), gb=m.get("grad_bias"),
             gphrase=m.get("grad_phrase"), sl=m.scaled_regularization_lambda())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("spec", [SPEC, SPEC_NOBN], ids=["bn", "nobn"])
def test_dp_equals_single_process_oracle(spec, tmp_path):
    import torch.multiprocessing as mp
    from tests.helpers import load_params, oracle_model
    B = 64
    port = _free_port()
    mp.spawn(_worker_oracle, args=(port, spec, B, str(tmp_path)), nprocs=WORLD, join=True)
    params, (words, ww, labels, iw, ids) = _global_problem(spec, B, 7)
    ref = oracle_model(spec)
    load_params(ref, params, False)
    ref.forward(words, ww, ids, iw)
    ref.backward()
    r = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % k)) for k in range(WORLD)]
    for k in range(WORLD):
        np.testing.assert_allclose(r[k]["cost"], ref.get_cost(), rtol=1e-13)
        np.testing.assert_allclose(r[k]["gT"], ref.get("grad_transform"), rtol=1e-11, atol=1e-16)
        np.testing.assert_allclose(r[k]["gb"], ref.get("grad_bias"), rtol=1e-11, atol=1e-16)
        np.testing.assert_allclose(r[k]["sl"], ref.scaled_regularization_lambda(), rtol=1e-15)
    # the sparse side stays shard-local: each rank holds the gradient rows of ITS windows, scaled by 1/B_global
    np.testing.assert_allclose(np.concatenate([r[0]["gphrase"], r[1]["gphrase"]]), ref.get("grad_phrase"), rtol=1e-10, atol=1e-16)


def test_shard_batch_rules():
    from cunvsm_amd import dp
    feats, labels = np.arange(24), np.arange(8)
    ids = np.arange(8 * 3)
    f, l, fw, w, i = dp.shard_batch(feats, labels, None, None, ids, 3, 2, 1, 2)
    assert list(l) == [4, 5, 6, 7] and list(f) == list(range(12, 24)) and list(i) == list(range(12, 24))
    assert fw is None and w is None
    with pytest.raises(ValueError):
        dp.shard_bounds(9, 0, 2)


def _worker_gpu(rank, port, spec, B, out_dir, transport="gloo"):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import cunvsm_amd as ca
    from cunvsm_amd import dp
    from tests.helpers import gpu_model, load_params
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=WORLD)
    params, (words, ww, labels, iw, ids) = _global_problem(spec, B, 7)
    w, wl, wwt, wi, wid = dp.shard_batch(words, labels, ww, iw, ids, spec["window"], spec["num_random"], rank, WORLD)
    m = gpu_model(spec, B // WORLD, world_size=WORLD, rank=rank, sync_batch_norm=1, device=rank if transport == "rccl" else 0)
    load_params(m, params, True)
    _connect(m, dist, rank, transport)
    m.compute_cost(ca.Batch(w, wl, wwt, wi), wid)
    m.compute_gradients()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), cost=m.get_cost(), gT=m.get_tensor("grad_transform"),
             gb=m.get_tensor("grad_bias"), gphrase=m.get_tensor("grad_phrase"), sl=m.scaled_regularization_lambda())
    m.update(0.1)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("spec,B", [(SPEC, 256), (SPEC_NOBN, 256), (SPEC_WIDE, 2 * 8704)], ids=["bn", "nobn", "wide"])
def test_dp_hip_equals_single_gpu(spec, B, tmp_path):
    import torch.multiprocessing as mp
    import cunvsm_amd as ca
    from tests.helpers import gpu_model, load_params, rel_err
    port = _free_port()
    mp.spawn(_worker_gpu, args=(port, spec, B, str(tmp_path)), nprocs=WORLD, join=True)
    params, (words, ww, labels, iw, ids) = _global_problem(spec, B, 7)
    ref = gpu_model(spec, B)
    load_params(ref, params, True)
    ref.compute_cost(ca.Batch(words, labels, ww, iw), ids)
    ref.compute_gradients()
    r = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % k)) for k in range(WORLD)]
    for k in range(WORLD):
        assert abs(r[k]["cost"] - ref.get_cost()) <= 1e-5 * abs(ref.get_cost())
        assert rel_err(r[k]["gT"], ref.get_tensor("grad_transform")) < 1e-5
        assert rel_err(r[k]["gb"], ref.get_tensor("grad_bias")) < 1e-5
        assert abs(r[k]["sl"] - ref.scaled_regularization_lambda()) < 1e-12
    assert rel_err(np.concatenate([r[0]["gphrase"], r[1]["gphrase"]]), ref.get_tensor("grad_phrase")) < 1e-5


def _worker_gpu_step(rank, port, spec, B, out_dir, exact=False, transport="gloo"):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import cunvsm_amd as ca
    from cunvsm_amd import dp
    from tests.helpers import gpu_model, load_params
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=WORLD)
    params, (words, ww, labels, iw, ids) = _global_problem(spec, B, 7)
    w, wl, wwt, wi, wid = dp.shard_batch(words, labels, ww, iw, ids, spec["window"], spec["num_random"], rank, WORLD)
    m = gpu_model(spec, B // WORLD, world_size=WORLD, rank=rank, sync_batch_norm=1, device=rank if transport == "rccl" else 0,
                  dp_exact_tables=int(exact))
    load_params(m, params, True)
    _connect(m, dist, rank, transport)
    costs = [m.step(ca.Batch(w, wl, wwt, wi), 0.05, entity_ids=wid, want_cost=True)]
    T_after_1 = m.get_param("word_entity_mapping-transform")
    for _ in range(3):
        m.step(ca.Batch(w, wl, wwt, wi), 0.05, entity_ids=wid)
    np.savez(os.path.join(out_dir, "step_rank%d.npz" % rank), cost=np.array(costs), T1=T_after_1, T=m.get_param("word_entity_mapping-transform"),
             b=m.get_param("word_entity_mapping-bias"), E=m.get_param("entity_representations-representations"),
             W=m.get_param("word_representations-representations"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("method,wide", [("sgd", False), ("sparse_adam", False), ("sparse_adam", True)], ids=["sgd", "sparse_adam", "sparse_adam_wide"])
def test_dp_fused_step(tmp_path, method, wide):
    """nvsm_step under data parallelism (documents update on side stream 1; dT GEMM, the all-reduce of the projection
    gradient and the projection update on side stream 2, joined by the next step): after the first step the replicated
    projection is identical on both ranks and equal to the single-GPU step on the whole batch; the loss of the first
    step is the global loss; further steps queued without reading the loss keep the replicas in lock-step."""
    import torch.multiprocessing as mp
    import cunvsm_amd as ca
    from tests.helpers import gpu_model, load_params, rel_err
    # (wide: the metric's dimensions at a per-rank batch above 8 192 rows — the split-bf16 kernels under the fused step)
    spec = dict(SPEC_WIDE if wide else SPEC, update_method=method)
    B = 2 * 8704 if wide else 256
    port = _free_port()
    mp.spawn(_worker_gpu_step, args=(port, spec, B, str(tmp_path)), nprocs=WORLD, join=True)
    params, (words, ww, labels, iw, ids) = _global_problem(spec, B, 7)
    ref = gpu_model(spec, B)
    load_params(ref, params, True)
    c0 = ref.step(ca.Batch(words, labels, ww, iw), 0.05, entity_ids=ids, want_cost=True)
    T1, b1 = ref.get_param("word_entity_mapping-transform"), ref.get_param("word_entity_mapping-bias")
    r = [np.load(os.path.join(str(tmp_path), "step_rank%d.npz" % k)) for k in range(WORLD)]
    assert abs(r[0]["cost"][0] - c0) <= 1e-5 * abs(c0) and abs(r[1]["cost"][0] - c0) <= 1e-5 * abs(c0)
    # replicas of the dense parameters stay in lock-step (both steps) ...
    np.testing.assert_array_equal(r[0]["T"], r[1]["T"])
    np.testing.assert_array_equal(r[0]["b"], r[1]["b"])
    # ... the first data-parallel step moves the projection as the single-GPU step on the whole batch does
    # (Adam divides by |g|: compare against the size of the step, not of T)
    np.testing.assert_array_equal(r[0]["T1"], r[1]["T1"])
    step = np.linalg.norm(T1 - params["word_entity_mapping-transform"])
    assert np.linalg.norm(r[0]["T1"] - T1) <= (2e-2 if method.endswith("adam") else 1e-4) * step
    # ... and the embedding tables are rank-local ("sparse rows stay GPU-local"): they differ between the ranks
    assert not np.array_equal(r[0]["E"], r[1]["E"])


def _worker_gpu_fold(rank, port, spec, sync_bn, fold, B, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["NVSM_DP_FOLD"] = "1" if fold else "0"      # (a documented switch: read once per handle, by nvsm_create)
    import torch.distributed as dist
    import cunvsm_amd as ca
    from cunvsm_amd import dp
    from tests.helpers import gpu_model, load_params
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=WORLD)
    params, (words, ww, labels, iw, ids) = _global_problem(spec, B, 7)
    w, wl, wwt, wi, wid = dp.shard_batch(words, labels, ww, iw, ids, spec["window"], spec["num_random"], rank, WORLD)
    m = gpu_model(spec, B // WORLD, world_size=WORLD, rank=rank, sync_batch_norm=sync_bn, device=0)
    load_params(m, params, True)
    calls = []
    inner = dp.torch_allreduce(dist)

    def counting(buf):
        calls.append((str(buf.dtype), int(buf.size)))
        inner(buf)
    m.set_allreduce_callback(counting)
    costs, per_step = [], []
    for s_ in range(3):
        n0 = len(calls)
        costs.append(m.step(ca.Batch(w, wl, wwt, wi), 0.05, entity_ids=wid, want_cost=True))
        per_step.append(calls[n0:])
    t = m.step_deferred(ca.Batch(w, wl, wwt, wi), 0.05, entity_ids=wid)      # ... and the loss read one step late
    costs.append(m.deferred_cost(t))
    np.savez(os.path.join(out_dir, "fold%d_rank%d.npz" % (int(fold), rank)), cost=np.array(costs), T=m.get_param("word_entity_mapping-transform"),
             b=m.get_param("word_entity_mapping-bias"), calls=np.array([len(c) for c in per_step]),
             sizes=np.array([sz for c in per_step[:1] for _, sz in c]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("spec,sync_bn,expect", [(SPEC, 0, (1, 2)), (SPEC_NOBN, 1, (1, 2)), (SPEC, 1, (3, 3))], ids=["per_shard_bn", "nobn", "sync_bn"])
def test_dp_one_collective_per_step_without_synchronised_statistics(tmp_path, spec, sync_bn, expect):
    """Round 6: with per-shard batch-norm statistics, and without batch-norm, nothing in front of the dT
    product needs another rank's sums — [db | loss] ride behind dT in ONE f32 all-reduce per step (model.cpp dp_fold) instead of an
    f64 all-reduce of their own (NVSM_DP_FOLD=0: the round-5 form). Same losses, same replicas; synchronised statistics keep three."""
    import torch.multiprocessing as mp
    from tests.helpers import rel_err
    B = 256
    spec = dict(spec, update_method="sparse_adam")
    res = {}
    for fold in (True, False):
        port = _free_port()
        mp.spawn(_worker_gpu_fold, args=(port, spec, sync_bn, fold, B, str(tmp_path)), nprocs=WORLD, join=True)
        res[fold] = [np.load(os.path.join(str(tmp_path), "fold%d_rank%d.npz" % (int(fold), k))) for k in range(WORLD)]
    de, dw = spec["entity_dim"], spec["word_dim"]
    for fold, want in ((True, expect[0]), (False, expect[1])):
        r = res[fold]
        assert list(r[0]["calls"]) == [want] * 3 and list(r[1]["calls"]) == [want] * 3, (fold, r[0]["calls"])
        np.testing.assert_array_equal(r[0]["T"], r[1]["T"])          # replicas in lock-step
        np.testing.assert_array_equal(r[0]["b"], r[1]["b"])
        np.testing.assert_array_equal(r[0]["cost"], r[1]["cost"])    # every rank reports the global loss
    if expect[0] == 1:
        assert list(res[True][0]["sizes"]) == [de * dw + de + 2]      # [dT | db | loss hi | loss lo]
    # folded and unfolded agree to fp32 roundoff (the bias gradient is summed over the ranks in f32 instead of f64)
    a, b = res[True][0], res[False][0]
    assert np.abs(a["cost"] - b["cost"]).max() <= 1e-6 * np.abs(b["cost"]).max()
    assert rel_err(a["T"], b["T"]) < 1e-6 and rel_err(a["b"], b["b"]) < 1e-5


# ---------------------------------------------------------------------------------------------
# loss TRAJECTORY of a data-parallel run against the single-process run: the dense parameters follow the
# all-reduced gradients, the embedding tables are rank-local and averaged at the end — not the single-GPU trajectory, but
# one that stays next to it: the loss of step 0 is the global loss exactly, the curve falls on fresh batches, and it lags the
# single-process curve by what rank-local tables cost (every replica's rows see only their own rank's share of the sparse
# gradient — with 2 ranks the tables train as if at half the learning rate): at least 60 % of the single-process
# improvement after 20 steps, never more than 15 % above its curve, never below it.
# ---------------------------------------------------------------------------------------------
TRAJ_SPEC = dict(num_words=300, num_entities=200, word_dim=16, entity_dim=12, window=4, num_random=5,
                 nonlinearity="hard_tanh", batch_norm=True, update_method="sgd")
TRAJ_SPEC["lambda"] = 0.01
TRAJ_STEPS, TRAJ_B, TRAJ_LR = 20, 512, 5.0


def _traj_batches(spec):
    """Batches with something to learn: a document's words come from its own 12-word region of the vocabulary."""
    from tests.helpers import random_params
    rs = np.random.RandomState(31)
    params = random_params(spec, rs)
    params["word_entity_mapping-transform"] = (params["word_entity_mapping-transform"] * 3).astype(np.float32)
    nV, nD, w, k, B = spec["num_words"], spec["num_entities"], spec["window"], spec["num_random"], TRAJ_B
    batches = []
    for _ in range(TRAJ_STEPS):
        labels = rs.randint(0, nD, B).astype(np.int64)
        words = ((labels[:, None] * 7 + rs.randint(0, 12, (B, w))) % nV).astype(np.int64).ravel()
        ids = rs.randint(0, nD, (B, k + 1)).astype(np.int64)
        ids[:, 0] = labels
        batches.append((words, np.ones(B * w, np.float32), labels, np.ones(B, np.float32), ids.ravel()))
    return params, batches


def _worker_traj(rank, port, spec, out_dir, use_gpu, exact=False, lr=None, separate_calls=False, transport="gloo", owner_rows=False):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from cunvsm_amd import dp
    from tests.helpers import gpu_model, load_params, oracle_model
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=WORLD)
    params, batches = _traj_batches(spec)
    if use_gpu:
        import cunvsm_amd as ca
        m = gpu_model(spec, TRAJ_B // WORLD, world_size=WORLD, rank=rank, sync_batch_norm=1, device=rank if transport == "rccl" else 0,
                      dp_exact_tables=int(exact))
        load_params(m, params, True)
        _connect(m, dist, rank, transport)
    else:
        m = oracle_model(spec)
        load_params(m, params, False)
        m.set_allreduce(dp.torch_allreduce(dist), WORLD)
        if exact:
            m.set_exact_tables(rank)
        if owner_rows:
            m.set_owner_rows(True)
    costs, tickets = [], []
    for words, ww, labels, iw, ids in batches:
        w, wl, wwt, wi, wid = dp.shard_batch(words, labels, ww, iw, ids, spec["window"], spec["num_random"], rank, WORLD)
        if use_gpu and separate_calls is True:
            m.compute_cost(ca.Batch(w, wl, wwt, wi), wid)
            m.compute_gradients()
            costs.append(m.get_cost())
            m.update(lr or TRAJ_LR)
        elif use_gpu and separate_calls == "deferred":          # the trainer's protocol: the loss is read one step late
            tickets.append(m.step_deferred(ca.Batch(w, wl, wwt, wi), lr or TRAJ_LR, entity_ids=wid))
            if len(tickets) > 1:
                costs.append(m.deferred_cost(tickets[-2]))
        elif use_gpu:
            costs.append(m.step(ca.Batch(w, wl, wwt, wi), lr or TRAJ_LR, entity_ids=wid, want_cost=True))
        else:
            m.forward(w, wwt, wid, wi)
            m.backward()
            costs.append(m.get_cost())
            m.update(lr or TRAJ_LR)
    if tickets:
        costs.append(m.deferred_cost(tickets[-1]))
    if use_gpu:
        if not exact:
            m.dp_average_tables()
        get = m.get_param
    else:
        get = m.get
    E, T, W = get("entity_representations-representations"), get("word_entity_mapping-transform"), get("word_representations-representations")
    np.savez(os.path.join(out_dir, "traj_rank%d.npz" % rank), cost=np.array(costs, np.float64), E=E, T=T, W=W)
    dist.barrier()
    dist.destroy_process_group()


def _check_trajectory(r, single_costs, params, E_single, T_single, averaged):
    single_costs = np.asarray(single_costs)
    for k in range(WORLD):
        c = r[k]["cost"]
        assert abs(c[0] - single_costs[0]) <= 1e-5 * abs(single_costs[0])          # same parameters: the global loss, exactly
        assert c[0] - c[-1] >= 0.6 * (single_costs[0] - single_costs[-1]), (c, single_costs)      # it trains
        rel = (c - single_costs) / single_costs
        assert rel.max() < 0.15 and rel.min() > -0.02, (c, single_costs)
    np.testing.assert_array_equal(r[0]["cost"], r[1]["cost"])                       # every rank reports the global loss
    np.testing.assert_array_equal(r[0]["T"], r[1]["T"])                             # dense replicas in lock-step for 20 steps
    # the projection moved the way the single process's projection moved
    T0 = params["word_entity_mapping-transform"].astype(np.float64)
    dT, dT1 = r[0]["T"] - T0, T_single - T0
    assert float(np.dot(dT, dT1) / (np.linalg.norm(dT) * np.linalg.norm(dT1))) > 0.8
    if averaged:                                                                     # one table for all ranks after nvsm_dp_average_tables
        np.testing.assert_array_equal(r[0]["E"], r[1]["E"])
        E0 = params["entity_representations-representations"].astype(np.float64)
        moved, moved_single = r[0]["E"] - E0, E_single - E0
        # parameter averaging: every rank moved its own rows, the mean moves in the single-process direction
        cos = float(np.dot(moved.ravel(), moved_single.ravel()) / (np.linalg.norm(moved) * np.linalg.norm(moved_single)))
        assert cos > 0.9, cos


def _single_oracle_trajectory(spec):
    from tests.helpers import load_params, oracle_model
    params, batches = _traj_batches(spec)
    o = oracle_model(spec)
    load_params(o, params, False)
    costs = []
    for words, ww, labels, iw, ids in batches:
        o.forward(words, ww, ids, iw)
        o.backward()
        costs.append(o.get_cost())
        o.update(TRAJ_LR)
    return params, costs, o.get("entity_representations-representations"), o.get("word_entity_mapping-transform")


# ---------------------------------------------------------------------------------------------
# exact data-parallel tables (nvsm_config.dp_exact_tables): every rank applies the sparse gradients of ALL ranks' windows —
# an all-gather of the update's inputs in front of the table passes — so N ranks follow the single-process trajectory on the
# global batch: same losses, same tables, replicas bit-identical, nothing to average.
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("method", ["sgd", "adagrad", "sparse_adam", "full_adam"])
def test_dp_exact_tables_oracle(tmp_path, method):
    import torch.multiprocessing as mp
    from tests.helpers import load_params, oracle_model
    spec = dict(TRAJ_SPEC, update_method=method)
    port = _free_port()
    mp.spawn(_worker_traj, args=(port, spec, str(tmp_path), False, True), nprocs=WORLD, join=True)
    params, batches = _traj_batches(spec)
    o = oracle_model(spec)
    load_params(o, params, False)
    costs = []
    for words, ww, labels, iw, ids in batches:
        o.forward(words, ww, ids, iw)
        o.backward()
        costs.append(o.get_cost())
        o.update(TRAJ_LR)
    r = [np.load(os.path.join(str(tmp_path), "traj_rank%d.npz" % k)) for k in range(WORLD)]
    for name in ("E", "W", "T"):
        np.testing.assert_array_equal(r[0][name], r[1][name])
    np.testing.assert_allclose(r[0]["cost"], costs, rtol=1e-9)
    np.testing.assert_allclose(r[0]["E"], o.get("entity_representations-representations"), rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(r[0]["W"], o.get("word_representations-representations"), rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(r[0]["T"], o.get("word_entity_mapping-transform"), rtol=1e-7, atol=1e-10)


# ---------------------------------------------------------------------------------------------
# owner-partitioned documents table (SURVEY.md §8e's "exact alternative"): row r of E and of its optimiser state
# belongs to rank r mod G; every rank applies, of the global batch's gathered sparse gradients, only the entries of ITS rows
# (1 / G of the update's work, optimiser state sharded), and the ranks then exchange their rows. CPU proof on the fp64 oracle
# over gloo: the 20-step trajectory is the single process's — same losses, same tables — for every optimiser, and no rank
# ever applied another rank's rows. (The HIP side is not built: there is no multi-GPU box to time the exchange on.)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("method", ["sgd", "adagrad", "sparse_adam", "dense_adam", "full_adam"])
def test_dp_owner_partitioned_documents_oracle(tmp_path, method):
    import torch.multiprocessing as mp
    from tests.helpers import load_params, oracle_model
    spec = dict(TRAJ_SPEC, update_method=method)
    port = _free_port()
    mp.spawn(_worker_traj, args=(port, spec, str(tmp_path), False, True, None, False, "gloo", True), nprocs=WORLD, join=True)
    params, batches = _traj_batches(spec)
    o = oracle_model(spec)
    load_params(o, params, False)
    costs = []
    for words, ww, labels, iw, ids in batches:
        o.forward(words, ww, ids, iw)
        o.backward()
        costs.append(o.get_cost())
        o.update(TRAJ_LR)
    r = [np.load(os.path.join(str(tmp_path), "traj_rank%d.npz" % k)) for k in range(WORLD)]
    for name in ("E", "W", "T"):
        np.testing.assert_array_equal(r[0][name], r[1][name])            # replicas identical after the row exchange
    np.testing.assert_allclose(r[0]["cost"], costs, rtol=1e-9)
    # a row is updated by ONE rank from the same entries in the same order as the single process does; what differs is the
    # summation order of the all-reduced batch statistics behind the gradients (1e-12 relative in fp64)
    np.testing.assert_allclose(r[0]["E"], o.get("entity_representations-representations"), rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(r[0]["W"], o.get("word_representations-representations"), rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(r[0]["T"], o.get("word_entity_mapping-transform"), rtol=1e-7, atol=1e-10)
    assert np.linalg.norm(r[0]["E"] - params["entity_representations-representations"].astype(np.float64).ravel()) > 0      # it trained


@pytest.mark.gpu
def test_dp_exact_tables_hip_wide(tmp_path):
    """The metric's dimensions at a per-rank batch above 8 192 rows (split-bf16 projection kernels, the large-batch CSR with
    its chunk order): four steps on one batch, tables against the single handle on the whole batch."""
    import torch.multiprocessing as mp
    import cunvsm_amd as ca
    from tests.helpers import gpu_model, load_params
    spec = dict(SPEC_WIDE, update_method="sparse_adam")
    B = 2 * 8704
    port = _free_port()
    mp.spawn(_worker_gpu_step, args=(port, spec, B, str(tmp_path), True), nprocs=WORLD, join=True)
    params, (words, ww, labels, iw, ids) = _global_problem(spec, B, 7)
    ref = gpu_model(spec, B)
    load_params(ref, params, True)
    for _ in range(4):
        ref.step(ca.Batch(words, labels, ww, iw), 0.05, entity_ids=ids)
    r = [np.load(os.path.join(str(tmp_path), "step_rank%d.npz" % k)) for k in range(WORLD)]
    for name, pname in (("E", "entity_representations-representations"), ("W", "word_representations-representations"),
                        ("T", "word_entity_mapping-transform")):
        np.testing.assert_array_equal(r[0][name], r[1][name])
        single = ref.get_param(pname).astype(np.float64).ravel()
        moved = np.linalg.norm(single - params[pname].astype(np.float64).ravel())
        diff = np.linalg.norm(r[0][name].astype(np.float64).ravel() - single)
        print(name, diff / moved)
        assert diff <= 2e-2 * moved, (name, diff, moved)      # (Adam divides by |g|: a sign flip of a tiny gradient is a whole step)


def _worker_exact_odd(rank, port, spec, B, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import cunvsm_amd as ca
    from cunvsm_amd import dp
    from tests.helpers import gpu_model, load_params
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=WORLD)
    params, (words, ww, labels, iw, ids) = _global_problem(spec, B, 11)
    w, wl, _, _, wid = dp.shard_batch(words, labels, None, None, ids, spec["window"], spec["num_random"], rank, WORLD)
    m = gpu_model(spec, B // WORLD + 5, world_size=WORLD, rank=rank, sync_batch_norm=1, device=0, dp_exact_tables=1)      # (capacity above the batch)
    load_params(m, params, True)
    m.set_allreduce_callback(dp.torch_allreduce(dist))
    for _ in range(3):
        m.step(ca.Batch(w, wl, None, None), 0.1, entity_ids=wid)
    np.savez(os.path.join(out_dir, "odd_rank%d.npz" % rank), E=m.get_param("entity_representations-representations"),
             W=m.get_param("word_representations-representations"), T=m.get_param("word_entity_mapping-transform"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["sgd", "sparse_adam"])
def test_dp_exact_tables_hip_odd_batch_no_weights(tmp_path, method):
    """An odd per-rank batch (77 windows, below the handle's capacity) without feature / instance weights (NULL pointers: nothing
    to gather for them)."""
    import torch.multiprocessing as mp
    import cunvsm_amd as ca
    from tests.helpers import gpu_model, load_params
    spec = dict(SPEC, update_method=method)
    B = 2 * 77
    port = _free_port()
    mp.spawn(_worker_exact_odd, args=(port, spec, B, str(tmp_path)), nprocs=WORLD, join=True)
    params, (words, ww, labels, iw, ids) = _global_problem(spec, B, 11)
    ref = gpu_model(spec, B)
    load_params(ref, params, True)
    for _ in range(3):
        ref.step(ca.Batch(words, labels, None, None), 0.1, entity_ids=ids)
    r = [np.load(os.path.join(str(tmp_path), "odd_rank%d.npz" % k)) for k in range(WORLD)]
    for name, pname in (("E", "entity_representations-representations"), ("W", "word_representations-representations"),
                        ("T", "word_entity_mapping-transform")):
        np.testing.assert_array_equal(r[0][name], r[1][name])
        single = ref.get_param(pname).astype(np.float64).ravel()
        moved = np.linalg.norm(single - params[pname].astype(np.float64).ravel())
        diff = np.linalg.norm(r[0][name].astype(np.float64).ravel() - single)
        assert diff <= (2e-2 if method.endswith("adam") else 2e-5) * moved, (name, diff, moved)


EXACT_LR = {"sgd": 5.0, "adagrad": 0.5, "sparse_adam": 0.02, "dense_adam": 0.02, "full_adam": 0.02}


@pytest.mark.gpu
@pytest.mark.parametrize("method,variant", [("sgd", "step"), ("sgd", "calls"), ("adagrad", "step"), ("sparse_adam", "step"),
                                            ("sparse_adam", "calls"), ("sparse_adam", "deferred"), ("dense_adam", "step"), ("full_adam", "step"),
                                            ("sparse_adam", "lazy"), ("sgd", "lazy"), ("sparse_adam", "l2_entity")])
def test_dp_exact_tables_hip(tmp_path, method, variant, monkeypatch):
    """nvsm_config.dp_exact_tables on two ranks sharing GPU 0 (gloo as the transport): after 20 steps the ranks' tables are
    bit-identical, and they are the tables of ONE handle stepping through the whole batches — up to the summation order of the
    all-reduced statistics (measured against how far the parameters moved)."""
    import torch.multiprocessing as mp
    import cunvsm_amd as ca
    from tests.helpers import gpu_model, load_params
    spec = dict(TRAJ_SPEC, update_method=method)
    if variant == "lazy":           # lazily decayed tables (rows >= entries of the GLOBAL batch; small tables need the override)
        spec.update(num_words=6000, num_entities=9000)
        monkeypatch.setenv("NVSM_LAZY_MIN_MB", "0")
    if variant == "l2_entity":
        spec["l2_entity"] = True
    lr = EXACT_LR[method]
    port = _free_port()
    mp.spawn(_worker_traj, args=(port, spec, str(tmp_path), True, True, lr, {"calls": True, "deferred": "deferred"}.get(variant, False)),
             nprocs=WORLD, join=True)
    params, batches = _traj_batches(spec)
    ref = gpu_model(spec, TRAJ_B)
    load_params(ref, params, True)
    costs = [ref.step(ca.Batch(words, labels, ww, iw), lr, entity_ids=ids, want_cost=True) for words, ww, labels, iw, ids in batches]
    r = [np.load(os.path.join(str(tmp_path), "traj_rank%d.npz" % k)) for k in range(WORLD)]
    for name in ("E", "W", "T", "cost"):
        np.testing.assert_array_equal(r[0][name], r[1][name])
    np.testing.assert_allclose(r[0]["cost"], costs, rtol=2e-4)
    for name, pname in (("E", "entity_representations-representations"), ("W", "word_representations-representations"),
                        ("T", "word_entity_mapping-transform")):
        single = ref.get_param(pname).astype(np.float64).ravel()
        moved = np.linalg.norm(single - params[pname].astype(np.float64).ravel())
        diff = np.linalg.norm(r[0][name].astype(np.float64).ravel() - single)
        print(method, variant, name, diff / moved)
        assert diff <= 2e-5 * moved, (name, diff, moved)


def test_dp_loss_trajectory_oracle(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker_traj, args=(port, TRAJ_SPEC, str(tmp_path), False), nprocs=WORLD, join=True)
    params, costs, E1, T1 = _single_oracle_trajectory(TRAJ_SPEC)
    r = [np.load(os.path.join(str(tmp_path), "traj_rank%d.npz" % k)) for k in range(WORLD)]
    _check_trajectory(r, costs, params, E1, T1, averaged=False)
    assert not np.array_equal(r[0]["E"], r[1]["E"])          # rank-local tables until somebody averages them


@pytest.mark.gpu
@pytest.mark.parametrize("collectives", ["two_streams", "main_stream"])
def test_dp_loss_trajectory_hip(tmp_path, collectives, monkeypatch):
    """The same 20 steps through nvsm_step on two ranks sharing GPU 0, with the step's collectives in either order of issue:
    side stream 2 for the projection gradient (default) or everything on the main stream (NVSM_DP_T_ON_MAIN=1, what
    nvsm_comm_init falls back to when its check of the two-stream order fails)."""
    import torch.multiprocessing as mp
    if collectives == "main_stream":
        monkeypatch.setenv("NVSM_DP_T_ON_MAIN", "1")
    else:
        monkeypatch.delenv("NVSM_DP_T_ON_MAIN", raising=False)
    port = _free_port()
    mp.spawn(_worker_traj, args=(port, TRAJ_SPEC, str(tmp_path), True), nprocs=WORLD, join=True)
    params, costs, E1, T1 = _single_oracle_trajectory(TRAJ_SPEC)
    r = [np.load(os.path.join(str(tmp_path), "traj_rank%d.npz" % k)) for k in range(WORLD)]
    _check_trajectory(r, costs, params, E1, T1, averaged=True)


@pytest.mark.gpu
@pytest.mark.parametrize("method,wide", [("sgd", False), ("sparse_adam", False), ("sparse_adam", True)], ids=["sgd", "sparse_adam", "sparse_adam_wide"])
def test_dp_fused_step_collectives_on_the_main_stream(tmp_path, method, wide, monkeypatch):
    monkeypatch.setenv("NVSM_DP_T_ON_MAIN", "1")
    test_dp_fused_step(tmp_path, method, wide)
