"""Randomised sweep of the HIP path against the fp64 oracle: 120 seeded configurations drawn over table sizes (from a
single row to tables much larger than the batch), odd and vectorisable dimensions, window 1…12, 1…20 negatives, batches
of 1…700 windows, both nonlinearities, batch-norm, biased negatives, λ ∈ {0, 0.01}, all five update methods and —
now and then — the optional L2 normalisers. Two optimiser steps each; same tolerances as tests/test_gpu_parity.py."""
import numpy as np
import pytest

import cunvsm_amd as ca
from tests.helpers import PARAMS, gpu_model, load_params, oracle_model, random_batch, random_params

pytestmark = pytest.mark.gpu

METHODS = ["sgd", "adagrad", "sparse_adam", "dense_adam", "full_adam"]


def draw_spec(rs):
    dims = [1, 2, 3, 4, 5, 7, 8, 12, 16, 20, 33, 64, 100, 128, 256, 300]
    spec = dict(
        num_words=int(rs.choice([1, 2, 7, 50, 400, 3000, 20000])),
        num_entities=int(rs.choice([1, 2, 5, 60, 500, 4000, 30000])),
        word_dim=int(rs.choice(dims)), entity_dim=int(rs.choice(dims)),
        window=int(rs.randint(1, 13)), num_random=int(rs.randint(1, 21)),
        nonlinearity=str(rs.choice(["tanh", "hard_tanh"])),
        bias_negative_samples=bool(rs.randint(2)),
        update_method=str(rs.choice(METHODS)),
        l2_phrase=bool(rs.rand() < 0.15), l2_entity=bool(rs.rand() < 0.15),
    )
    spec["lambda"] = float(rs.choice([0.0, 0.01]))
    B = int(rs.choice([1, 2, 3, 17, 64, 100, 256, 700]))
    spec["batch_norm"] = bool(rs.randint(2)) and B >= 8          # batch statistics of a handful of rows are degenerate
    # Degenerate corners in which the exact gradient is 0 and an adaptive optimiser turns fp32 rounding residue into
    # full-size steps (nothing to compare): a 1-dimensional vector through the L2 normaliser (its Jacobian vanishes), and
    # batch-norm / the phrase normaliser over a single-word vocabulary (every phrase is the same direction; the reference's
    # own constant-input tests switch batch-norm off for that reason, cpp/gradient_checking_tests.cu:68-116).
    if spec["l2_phrase"]:
        spec["word_dim"] = max(spec["word_dim"], 2)
    if spec["l2_entity"]:
        spec["entity_dim"] = max(spec["entity_dim"], 2)
    if spec["l2_phrase"] or spec["batch_norm"]:
        spec["num_words"] = max(spec["num_words"], 50)
    return spec, B


def _run_configuration(seed):
    rs = np.random.RandomState(1000 + seed)
    spec, B = draw_spec(rs)
    params = random_params(spec, rs, scale=0.3 if (spec["l2_phrase"] or spec["l2_entity"]) else None)
    o, g = oracle_model(spec), gpu_model(spec, B)
    load_params(o, params, False)
    load_params(g, params, True)
    lr = {"sgd": 0.1, "adagrad": 0.01}.get(spec["update_method"], 0.001)
    start = {p: o.get(p).copy() for p in PARAMS}
    for step in range(2):
        words, ww, labels, iw, ids = random_batch(spec, rs, B, zipf=bool(rs.randint(2)))
        o.forward(words, ww, ids, iw)
        o.backward()
        o.update(lr)
        cg = g.step(ca.Batch(words, labels, ww, iw), lr, entity_ids=ids, want_cost=True)
        co = o.get_cost()
        assert abs(cg - co) <= 5e-5 * abs(co) + 1e-7, (spec, B, step, cg, co)
    adam = spec["update_method"].endswith("adam")
    for p in PARAMS:
        delta = np.linalg.norm(o.get(p) - start[p])
        err = np.linalg.norm(g.get_param(p).astype(np.float64) - o.get(p))
        # Adam's first steps move every component by ≈ lr whatever its gradient: components whose gradient is at the
        # fp32 noise level of a small batch take a visibly different step (see tests/test_gpu_configs.py)
        tol = (2e-2 if adam else 5e-4) * max(delta, 1e-12) + 2e-7 * np.linalg.norm(o.get(p)) + 1e-9
        assert err <= tol, (spec, B, p, err, delta)


@pytest.mark.parametrize("seed", range(120))
def test_random_configuration(seed):
    _run_configuration(seed)


@pytest.mark.parametrize("seed", range(0, 120, 3))
def test_random_configuration_large_table_paths(seed, monkeypatch):
    """The same draws with the paths of tables much larger than the batch switched on whatever the sizes: lazy dense decay
    (rows brought up to date as they are read) and the walk over the sorted entries instead of the list of rows —
    update.hip entry_walk_kernel, which a batch this small would not take by itself."""
    monkeypatch.setenv("NVSM_LAZY_MIN_MB", "0")
    monkeypatch.setenv("NVSM_ENTRY_WALK_MIN", "0")
    _run_configuration(seed)
