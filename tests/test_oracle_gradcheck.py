"""Full-parameter central-difference gradient check of the fp64 oracle, restating
cpp/gradient_checking_tests.cu:276-338 + include/cuNVSM/tests_base_cuda.h:132-232:
20 words / 15 docs / dims 3→4, B=1024, w=3, k=1, λ=.01, ε=1e-5, rel-err < 1e-4 on every parameter,
data as RandomSource (ids U{0..10}, weights U[0,2)), followed by an optimiser step between checks."""
import numpy as np
import pytest

from oracle import nvsm_oracle as orc

DESCS = {
    "tanh": dict(batch_norm=False, nonlinearity=orc.TANH),
    "bn_tanh": dict(batch_norm=True, nonlinearity=orc.TANH),
    "hard_tanh": dict(batch_norm=False, nonlinearity=orc.HARD_TANH),
    "bn_hard_tanh": dict(batch_norm=True, nonlinearity=orc.HARD_TANH),
    "bn_tanh_bias_neg": dict(batch_norm=True, nonlinearity=orc.TANH, bias_negative_samples=True),
    "bn_tanh_l2_phrase": dict(batch_norm=True, nonlinearity=orc.TANH, l2_phrase=True),
    "bn_tanh_l2_entity": dict(batch_norm=True, nonlinearity=orc.TANH, l2_entity=True),
    "bn_tanh_l2_both": dict(batch_norm=True, nonlinearity=orc.TANH, l2_phrase=True, l2_entity=True),
}
UPDATES = {
    "sgd": (orc.SGD, orc.ADAM_NONE, 0.1),
    "adagrad": (orc.ADAGRAD, orc.ADAM_NONE, 0.01),
    "adam_sparse": (orc.ADAM, orc.ADAM_SPARSE, 0.001),
    "adam_dense": (orc.ADAM, orc.ADAM_DENSE_UPDATE, 0.001),
    "adam_full": (orc.ADAM, orc.ADAM_DENSE_UPDATE_DENSE_VARIANCE, 0.001),
}


def random_batch(rs, B, w):
    words = rs.randint(0, 11, size=B * w)
    ww = rs.uniform(0.0, 2.0, size=B * w)
    labels = rs.randint(0, 11, size=B)
    iw = rs.uniform(0.0, 2.0, size=B)
    return words, ww, labels, iw


@pytest.mark.parametrize("desc", sorted(DESCS))
@pytest.mark.parametrize("upd", sorted(UPDATES))
@pytest.mark.parametrize("seed", [0, 3])
def test_random_source_gradient_check(desc, upd, seed):
    method, mode, lr = UPDATES[upd]
    cfg = orc.make_config(20, 15, 3, 4, 3, 1, lambda_=0.01, update_method=method, adam_mode=mode, **DESCS[desc])
    rng = orc.Rng(seed + 1)
    m = orc.Model(cfg)
    m.initialize(rng)
    rs = np.random.RandomState(seed)
    # hard_tanh has kinks: at B=1024 a 1e-5 step crosses some of them (the reference only logs those,
    # tests_base_cuda.h:177-185, and its own comment at :203-214 recommends fewer datapoints) — B=32 there.
    B = 1024 if DESCS[desc]["nonlinearity"] == orc.TANH else 32
    for _ in range(3):
        words, ww, labels, iw = random_batch(rs, B, 3)
        ids = rng.generate_labels(labels, 15, 1)
        failed, checked, worst = m.gradcheck(words, ww, ids, iw, eps=1e-5, thresh=1e-4)
        assert checked == 20 * 3 + 15 * 4 + 12 + 4
        assert failed == 0, (failed, worst)
        m.update(lr)


# cpp/gradient_checking_tests.cu:68-116 — constant input (no BN: all rows equal ⇒ activations collapse)
@pytest.mark.parametrize("desc", ["tanh", "hard_tanh"])
def test_constant_source_gradient_check(desc):
    cfg = orc.make_config(20, 15, 3, 4, 3, 1, lambda_=0.01, **DESCS[desc])
    rng = orc.Rng(1)
    m = orc.Model(cfg)
    m.initialize(rng)
    B = 1024 if desc == "tanh" else 32
    words, ww, labels, iw = np.full(B * 3, 10), np.ones(B * 3), np.full(B, 10), np.ones(B)
    for _ in range(2):
        ids = rng.generate_labels(labels, 15, 1)
        failed, checked, worst = m.gradcheck(words, ww, ids, iw)
        assert failed == 0, (failed, worst)
        m.update(0.1)
