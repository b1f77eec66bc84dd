"""Pins the CPU oracle against every golden vector / known-answer test the reference's own test-suite
holds for the hot path. Each test names the reference test it restates."""
import itertools
import math

import numpy as np
import pytest

from oracle import nvsm_oracle as orc

ULP = dict(rtol=2e-15, atol=0.0)          # googletest DoubleEq = 4 ULPs
TIGHT = dict(rtol=1e-12, atol=1e-15)
UPDATE_PARAMS = list(itertools.product([0.0, 0.1], [1.0, 0.5]))     # cpp/updates_tests.cu:28-32 (λ, lr)


# ---- cpp/model_tests.cu:52-123 -----------------------------------------------------------------
def test_get_average_representations():
    table = np.arange(12.0)                                         # 4 objects x 3 dims
    idx = [1, 3, 2, 0, 3, 1]
    avg = orc.average_repr(table, 3, idx, None, 3)
    np.testing.assert_allclose(avg, [(3 + 9 + 6) / 3., (4 + 10 + 7) / 3., (5 + 11 + 8) / 3.,
                                     (0 + 9 + 3) / 3., (1 + 10 + 4) / 3., (2 + 11 + 5) / 3.], **ULP)


def test_get_weighted_average_representations():
    table = np.arange(12.0)
    idx = [1, 3, 2, 0, 3, 1]
    w = [0.5, 0.3, 0.1, 1.0, 2.0, 0.2]
    avg = orc.average_repr(table, 3, idx, w, 3)
    expect = [(0.5 * 3 + 0.3 * 9 + 0.1 * 6) / 3., (0.5 * 4 + 0.3 * 10 + 0.1 * 7) / 3., (0.5 * 5 + 0.3 * 11 + 0.1 * 8) / 3.,
              (1.0 * 0 + 2.0 * 9 + 0.2 * 3) / 3., (1.0 * 1 + 2.0 * 10 + 0.2 * 4) / 3., (1.0 * 2 + 2.0 * 11 + 0.2 * 5) / 3.]
    np.testing.assert_allclose(avg, expect, **ULP)                  # divides by window, not by Σweights


# ---- cpp/model_tests.cu:125-151 ----------------------------------------------------------------
@pytest.mark.parametrize("seed", range(0, 11))
def test_generate_labels(seed):
    rng = orc.Rng(seed if seed else 1)
    ids = rng.generate_labels([1, 2, 3, 4, 5], 5000, 10)
    assert ids.size == 5 * 11
    assert list(ids[::11]) == [1, 2, 3, 4, 5]
    assert ids.min() >= 0 and ids.max() < 5000


# ---- cpp/model_tests.cu:153-243 ----------------------------------------------------------------
def test_representations_update_decay_only():
    r = orc.Reps(4, 3)
    r.set(np.arange(12.0))
    # regularization_lambda 0.1, batch (num instances) 2 → scaled λ = 0.05; zero gradient
    r.update([(np.zeros(6), [0, 3, 1, 0], 2, np.ones(4))], 0.1, 0.1 / 2.0)
    scale = 1.0 - (0.1 * 0.1) / 2.0
    np.testing.assert_allclose(r.get(), np.arange(12.0) * scale, **ULP)


def test_representations_update_duplicates():
    r = orc.Reps(4, 3)
    r.set(np.arange(12.0))
    r.update([([5.0, 4.0, 3.0, -3.0, -2.0, 10.0], [0, 3, 1, 0], 2, np.ones(4))], 0.1, 0.0)
    lr = 0.1
    expect = [0. + (5.0 + (-3.0)) * lr, 1. + (4.0 + (-2.0)) * lr, 2. + (3.0 + 10.0) * lr,
              3. + (-3.0) * lr, 4. + (-2.0) * lr, 5. + 10.0 * lr, 6., 7., 8.,
              9. + 5.0 * lr, 10. + 4.0 * lr, 11. + 3.0 * lr]
    np.testing.assert_allclose(r.get(), expect, rtol=1e-14)


# ---- cpp/model_tests.cu:245-275 ----------------------------------------------------------------
def test_update_dense():
    r = orc.Reps(4, 3)
    r.set(np.arange(12.0))
    r.update_dense_const(10.0, 0.1, 0.01)
    np.testing.assert_allclose(r.get(), np.arange(12.0) * (1.0 - 0.01 * 0.1) + 10.0 * 0.1, **ULP)


# ---- cpp/model_tests.cu:277-339 ----------------------------------------------------------------
def _transform_model(bn, bn_eps=1e-4):
    cfg = orc.make_config(2, 1, 3, 5, 1, 1, batch_norm=bn, nonlinearity=orc.TANH, bn_epsilon=bn_eps)
    m = orc.Model(cfg)
    m.set("word_representations-representations", [0.01, 0.02, 0.03, 0.001, 0.002, 0.003])
    m.set("entity_representations-representations", np.zeros(5))
    m.set("word_entity_mapping-transform", np.arange(15.0))
    m.set("word_entity_mapping-bias", np.arange(5.0) * 1e-3)
    m.forward([0, 1], [1.0, 1.0], [0, 0, 0, 0], [1.0, 1.0])
    return m


def test_transform_tanh():
    m = _transform_model(False)
    expect = np.tanh([0.400, 0.461, 0.522, 0.583, 0.644, 0.040, 0.047, 0.054, 0.061, 0.068])
    np.testing.assert_allclose(m.get("proj"), expect, rtol=1e-14)


# ---- cpp/model_tests.cu:468-548 ----------------------------------------------------------------
def test_transform_batchnorm(golden):
    g = golden("transform_batchnorm")
    m = _transform_model(True, g["bn_epsilon"])
    np.testing.assert_allclose(m.get("proj"), g["output"], rtol=1e-12)
    m.backward()
    assert np.isfinite(m.get("grad_transform")).all() and np.isfinite(m.get("grad_bias")).all()


# ---- cpp/model_tests.cu:341-466 — the end-to-end forward+backward KAT ---------------------------
def test_transform_backward(golden):
    g = golden("transform_backward")
    rng = orc.Rng(g["seed"])
    cfg = orc.make_config(g["num_words"], g["num_entities"], g["word_dim"], g["entity_dim"], g["window_size"],
                          g["num_random_entities"], bias_negative_samples=True, lambda_=g["regularization_lambda"],
                          clip_sigmoid=False, nonlinearity=orc.TANH)
    m = orc.Model(cfg)
    m.initialize(rng)
    B, w = g["batch_size"], g["window_size"]
    words = np.full(B * w, g["feature_value"], dtype=np.int64)
    ids = rng.generate_labels(np.full(B, g["label"], dtype=np.int64), g["num_entities"], g["num_random_entities"])
    m.forward(words, np.ones(B * w), ids, np.ones(B))
    m.get_cost()
    m.backward()
    np.testing.assert_allclose(m.get("grad_transform"), g["grad_transform"], **TIGHT)
    np.testing.assert_allclose(m.get("grad_bias"), g["grad_bias"], **TIGHT)
    np.testing.assert_allclose(m.get("grad_phrase"), g["grad_phrase"], **TIGHT)


# ---- cpp/cudnn_utils_tests.cu:18-35,114-176 -----------------------------------------------------
def test_batchnorm_constant_input():
    y, _, _ = orc.bn_forward(np.ones(1000), 100, 10, np.zeros(10), 1e-4)
    assert np.all(y == 0.0)


def test_batchnorm_forward_backward(golden):
    g = golden("batchnorm_forward_backward")
    n, d, eps = g["num_instances"], g["num_features"], g["epsilon"]
    x = np.array(g["input"])
    y, mean, inv = orc.bn_forward(x, n, d, np.zeros(d), eps)
    expect = (x.reshape(n, d) - np.array(g["mean"])) / np.sqrt(np.array(g["variance"]) + eps)
    np.testing.assert_allclose(y, expect.ravel(), rtol=1e-14)
    dx, gb = orc.bn_backward(g["grad"], x, n, d, mean, inv)
    np.testing.assert_allclose(gb, g["grad_bias"], rtol=1e-14)
    np.testing.assert_allclose(dx, g["grad_input"], rtol=1e-6, atol=1e-18)    # catastrophic cancellation in the KAT itself


# ---- cpp/cuda_utils_tests.cu:8-21 ---------------------------------------------------------------
def test_truncated_sigmoid(golden):
    g = golden("truncated_sigmoid")
    assert orc.truncated_sigmoid(0.0, 0.0) == 0.5
    np.testing.assert_allclose(orc.truncated_sigmoid(1.0, 0.0), g["sigmoid_1"], **ULP)
    np.testing.assert_allclose(orc.truncated_sigmoid(-1.0, 0.0), 1.0 - g["sigmoid_1"], **ULP)
    assert orc.truncated_sigmoid(-50.0, 0.0) > 0.0
    assert orc.truncated_sigmoid(20.0, 0.0) < 1.0
    for c in g["cases"]:
        np.testing.assert_allclose(orc.truncated_sigmoid(c["x"], c["eps"]), c["expect"], **ULP)


# ---- cpp/cuda_utils_tests.cu:51-92 --------------------------------------------------------------
def test_normalizer(golden):
    g = golden("normalizer")
    n, d = g["num_instances"], g["num_features"]
    x = np.array(g["input"], dtype=np.float64)
    y, norms = orc.normalizer_forward(x, n, d)
    l2 = np.sqrt((x.reshape(n, d) ** 2).sum(1))
    np.testing.assert_allclose(y, (x.reshape(n, d) / l2[:, None]).ravel(), **ULP)
    gin = orc.normalizer_backward(g["grad_output"], x, norms, n, d)
    np.testing.assert_allclose(gin, g["grad_input"], rtol=1e-11)


# ---- cpp/updates_tests.cu:34-172 ----------------------------------------------------------------
GRAD_MATRIX = np.arange(1.0, 25.0)
GRAD_BIAS = np.array([25.0, 26.0, 27.0])


@pytest.mark.parametrize("lam,lr", UPDATE_PARAMS)
def test_sgd_transform(lam, lr):
    t = orc.Transform(8, 3, orc.SGD)
    t.fill(5.0)
    t.update(GRAD_MATRIX, GRAD_BIAS, lr, lam)
    np.testing.assert_allclose(t.get(0), 5.0 + lr * (GRAD_MATRIX - lam * 5.0), rtol=1e-14)
    np.testing.assert_allclose(t.get(1), 5.0 + lr * GRAD_BIAS, rtol=1e-14)       # bias never regularised


@pytest.mark.parametrize("lam,lr", UPDATE_PARAMS)
def test_sgd_representations(lam, lr):
    r = orc.Reps(10, 4, orc.SGD)
    r.fill(5.0)
    g1, g2 = np.array([2.0, 2.5, 3.0, 4.0]), np.array([10.0, 11.0, 12.0, 13.0])
    r.update([(g1, [9, 0, 1], 3, None), (g2, [5, 1, 8], 3, None)], lr, lam)
    expect = np.full((10, 4), (1.0 - lr * lam) * 5.0)
    for rows, g in (([9, 0, 1], g1), ([5, 1, 8], g2)):
        for row in rows:
            expect[row] += lr * g
    np.testing.assert_allclose(r.get().reshape(10, 4), expect, rtol=1e-14)


# ---- cpp/updates_tests.cu:174-297 ---------------------------------------------------------------
@pytest.mark.parametrize("lam,lr", UPDATE_PARAMS)
def test_adagrad_transform(lam, lr):
    eps = 1e-6
    t = orc.Transform(8, 3, orc.ADAGRAD, eps=eps)
    t.fill(5.0)
    gt, gb = t.update(GRAD_MATRIX, GRAD_BIAS, lr, lam)
    np.testing.assert_allclose(t.get(2), GRAD_MATRIX ** 2, **ULP)
    np.testing.assert_allclose(t.get(3), [625.0, 676.0, 729.0], **ULP)
    np.testing.assert_allclose(gt, GRAD_MATRIX / np.sqrt(GRAD_MATRIX ** 2 + eps), **ULP)
    np.testing.assert_allclose(gb, GRAD_BIAS / np.sqrt(GRAD_BIAS ** 2 + eps), **ULP)


@pytest.mark.parametrize("lam,lr", UPDATE_PARAMS)
def test_adagrad_representations(lam, lr, golden):
    g = golden("adagrad_representations")
    eps = g["epsilon"]
    r = orc.Reps(g["num_objects"], g["repr_size"], orc.ADAGRAD, eps=eps)
    r.fill(g["initial"])
    (grad,) = r.update([(g["grad"], g["indices"], g["window_size"], None)], lr, lam)
    np.testing.assert_allclose(r.get(1), g["accumulator"], **ULP)
    d1 = math.sqrt(((8.8125 + 8.8125 + 142.3125) / 3.0) + eps)
    d2 = math.sqrt(((133.5 + 142.3125 + 133.5) / 3.0) + eps)
    np.testing.assert_allclose(grad, [2.0 / d1, 2.5 / d1, 3.0 / d1, 4.0 / d1, 10.0 / d2, 11.0 / d2, 12.0 / d2, 13.0 / d2], **ULP)


# ---- cpp/updates_tests.cu:299-425 — incl. the non-decaying bias moments quirk -------------------
@pytest.mark.parametrize("lam,lr", UPDATE_PARAMS)
def test_adam_transform(lam, lr, golden):
    g = golden("adam_transform")
    eps, b1, b2 = g["epsilon"], g["beta1"], g["beta2"]
    t = orc.Transform(8, 3, orc.ADAM, b1, b2, eps)
    t.fill(5.0)
    gt, gb = t.update(GRAD_MATRIX, GRAD_BIAS, lr, lam)
    bc1 = math.sqrt(1.0 - b2 ** 1) / (1.0 - b1 ** 1)
    gg = GRAD_MATRIX - lam * 5.0
    np.testing.assert_allclose(gt, bc1 * ((1.0 - b1) * gg) / (np.sqrt((1.0 - b2) * gg ** 2) + eps), rtol=1e-14)
    np.testing.assert_allclose(gb, g["step1"]["grad_bias_out"], **ULP)
    np.testing.assert_allclose(t.get(3), g["step1"]["m_bias"], **ULP)
    np.testing.assert_allclose(t.get(5), g["step1"]["v_bias"], **ULP)

    before = t.get(0)
    gt, gb = t.update(GRAD_MATRIX, GRAD_BIAS, lr, lam)
    bc2 = math.sqrt(1.0 - b2 ** 2) / (1.0 - b1 ** 2)
    gg2 = GRAD_MATRIX - lam * before
    m1 = (1.0 - b1) * gg
    v1 = (1.0 - b2) * gg ** 2
    np.testing.assert_allclose(gt, bc2 * (b1 * m1 + (1.0 - b1) * gg2) / (np.sqrt(b2 * v1 + (1.0 - b2) * gg2 ** 2) + eps), rtol=1e-14)
    np.testing.assert_allclose(gb, g["step2"]["grad_bias_out"], **ULP)
    np.testing.assert_allclose(t.get(3), g["step2"]["m_bias"], **ULP)           # m_b += (1−β1)·g — never decays
    np.testing.assert_allclose(t.get(5), g["step2"]["v_bias"], **ULP)


# ---- cpp/updates_tests.cu:427-775 ---------------------------------------------------------------
G1, G2 = np.array([2.0, 2.5, 3.0, 4.0]), np.array([10.0, 11.0, 12.0, 13.0])
B1, B2, EPS = 0.9, 0.999, 1e-5
BC = math.sqrt(1.0 - B2) / (1.0 - B1)


def _expected_m():
    return np.array([(1.0 - B1) * G1, (1.0 - B1) * (G1 + G2), (1.0 - B1) * G2, (1.0 - B1) * G2, (1.0 - B1) * G1])


EXPECTED_V = np.array([(1.0 - B2) * 8.8125, (1.0 - B2) * (8.8125 + 133.5), (1.0 - B2) * 133.5, (1.0 - B2) * 133.5, (1.0 - B2) * 8.8125])


@pytest.mark.parametrize("lam,lr", UPDATE_PARAMS)
def test_adam_representations_sparse(lam, lr):
    r = orc.Reps(5, 4, orc.ADAM, orc.ADAM_SPARSE, B1, B2, EPS)
    r.fill(5.0)
    (grad,) = r.update([(np.concatenate([G1, G2]), [4, 0, 1, 3, 1, 2], 3, None)], lr, lam)
    m, v = _expected_m(), EXPECTED_V
    np.testing.assert_allclose(r.get(1).reshape(5, 4), m, rtol=1e-14)
    np.testing.assert_allclose(r.get(2), v, rtol=1e-14)
    e1 = BC * ((m[4] + m[0] + m[1]) / 3) / (math.sqrt((v[4] + v[0] + v[1]) / 3) + EPS)
    e2 = BC * ((m[3] + m[1] + m[2]) / 3) / (math.sqrt((v[3] + v[1] + v[2]) / 3) + EPS)
    np.testing.assert_allclose(grad, np.concatenate([e1, e2]), rtol=1e-13)


@pytest.mark.parametrize("lam,lr", UPDATE_PARAMS)
def test_adam_representations_dense_update(lam, lr):
    r = orc.Reps(5, 4, orc.ADAM, orc.ADAM_DENSE_UPDATE, B1, B2, EPS)
    r.fill(5.0)
    r.update([(G1, [4, 0, 1], 3, None), (G2, [3, 1, 2], 3, None)], lr, lam)
    m, v = _expected_m(), EXPECTED_V
    np.testing.assert_allclose(r.get(1).reshape(5, 4), m, rtol=1e-14)
    np.testing.assert_allclose(r.get(2), v, rtol=1e-14)
    expect = 5.0 + lr * (BC * m / (np.sqrt(v)[:, None] + EPS) - lam * 5.0)
    np.testing.assert_allclose(r.get(0).reshape(5, 4), expect, rtol=1e-13)


@pytest.mark.parametrize("lam,lr", UPDATE_PARAMS)
def test_adam_representations_dense_update_dense_variance(lam, lr):
    r = orc.Reps(5, 4, orc.ADAM, orc.ADAM_DENSE_UPDATE_DENSE_VARIANCE, B1, B2, EPS)
    r.fill(5.0)
    r.update([(G1, [4, 0, 1], 3, None), (G2, [3, 1, 2], 3, None)], lr, lam)
    agg = np.array([G1, G1 + G2, G2, G2, G1]) - lam * 5.0
    np.testing.assert_allclose(r.get(1).reshape(5, 4), (1.0 - B1) * agg, rtol=1e-13)
    np.testing.assert_allclose(r.get(2).reshape(5, 4), (1.0 - B2) * agg ** 2, rtol=1e-13)
    expect = 5.0 + lr * (BC * (1.0 - B1) * agg / (np.sqrt((1.0 - B2) * agg ** 2) + EPS))
    np.testing.assert_allclose(r.get(0).reshape(5, 4), expect, rtol=1e-13)
