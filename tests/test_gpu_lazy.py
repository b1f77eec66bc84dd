"""-m gpu: lazy dense decay (csrc/kernels.h, "lazy dense decay") against the eager path it replaces.

For tables with at least as many rows as a batch has entries (BASELINE configs[3] / configs[4]) the rows a batch does not
touch are not rewritten on every update; the per-update factors the reference applies to every row (θ·(1 − λ·lr), Adam
m·β₁, v·β₂ — cpp/storage.cu:65-67, cpp/updates_adam.cu:196-252) are applied when a row is next gathered or updated, one
factor at a time in update order, i.e. with the very roundings of the dense pass. The claim is therefore BIT equality
with the eager path (NVSM_LAZY_DECAY=0 builds the eager twin), which tests/test_gpu_parity.py / test_gpu_configs.py in turn
hold against the fp64 oracle: parameters and optimiser state after hundreds of updates with changing learning rates,
ragged batches, forward-only calls in between, parameter reads in between, fused and separate calls."""
import numpy as np
import pytest

import cunvsm_amd as ca
from tests.helpers import PARAMS, gpu_model, load_params, random_batch, random_params

pytestmark = pytest.mark.gpu

STATE = {
    "sgd": [],
    "adagrad": ["word_representations/a", "entity_representations/a"],
    "sparse_adam": ["word_representations/m", "word_representations/v", "entity_representations/m", "entity_representations/v"],
}


def twins(spec, B, monkeypatch):
    monkeypatch.setenv("NVSM_LAZY_DECAY", "0")
    eager = gpu_model(spec, B, sampler=ca.SAMPLER_DEVICE)
    monkeypatch.setenv("NVSM_LAZY_DECAY", "1")
    monkeypatch.setenv("NVSM_LAZY_MIN_MB", "0")          # by default only tables of hundreds of MB decay lazily
    lazy = gpu_model(spec, B, sampler=ca.SAMPLER_DEVICE)
    return eager, lazy


def same_everywhere(a, b, method):
    for n in list(PARAMS) + STATE[method]:
        np.testing.assert_array_equal(a.get_param(n), b.get_param(n), err_msg=n)


@pytest.mark.parametrize("method,lam,bn,dims", [("sparse_adam", 0.02, True, (12, 16)), ("sparse_adam", 0.0, False, (12, 16)),
                                                ("sparse_adam", 0.02, False, (7, 5)), ("sgd", 0.05, False, (12, 16)),
                                                ("adagrad", 0.05, False, (12, 16)), ("adagrad", 0.05, True, (300, 256))])
def test_lazy_decay_is_bit_identical_to_the_dense_passes(method, lam, bn, dims, monkeypatch):
    spec = dict(num_words=3000, num_entities=5000, word_dim=dims[0], entity_dim=dims[1], window=3, num_random=2,
                nonlinearity="hard_tanh" if bn else "tanh", batch_norm=bn, update_method=method)
    spec["lambda"] = lam
    B = 40
    rs = np.random.RandomState(hash((method, dims)) % 1000)
    params = random_params(spec, rs)
    eager, lazy = twins(spec, B, monkeypatch)
    for m in (eager, lazy):
        m.initialize(3)
        load_params(m, params, True)
    steps = 2 * 128 + 37                                    # crosses two of the periodic whole-table refreshes
    for s in range(steps):
        b = int(rs.choice([1, 7, 33, 40]))
        words, ww, labels, iw, ids = random_batch(spec, rs, b, zipf=True)
        lr = float(rs.choice([1e-3, 5e-3, 2e-2]))
        batch = ca.Batch(words, labels, ww, iw)
        for m in (eager, lazy):
            if s % 11 == 3:                                  # a forward pass that is never followed by an update
                m.compute_cost(batch, ids)
                m.get_cost()
            if s % 2:
                m.step(batch, lr, entity_ids=ids)
            else:
                m.compute_cost(batch, ids)
                m.compute_gradients()
                m.update(lr)
        if s in (5, 130, 200):
            same_everywhere(eager, lazy, method)             # reads bring every row up to date on the lazy side
    assert eager.get_cost() == lazy.get_cost()
    same_everywhere(eager, lazy, method)
    # ... and the two really took different routes
    words, ww, labels, iw, ids = random_batch(spec, rs, B, zipf=True)
    for m in (eager, lazy):
        m.profile_enable(True)
        m.step(ca.Batch(words, labels, ww, iw), 1e-3, entity_ids=ids)
    assert {"lazy_stamp_words", "lazy_stamp_entities"} <= set(lazy.profile())
    assert not any(k.startswith("lazy_") for k in eager.profile())


def test_lazy_decay_with_the_device_sampler_and_set_param(monkeypatch):
    """The sampled negatives are part of the touched rows; set_param in the middle of a run lands on up-to-date rows."""
    spec = dict(num_words=2000, num_entities=6000, word_dim=16, entity_dim=8, window=4, num_random=5, nonlinearity="hard_tanh",
                batch_norm=True, update_method="sparse_adam")
    spec["lambda"] = 0.01
    B = 64
    rs = np.random.RandomState(4)
    eager, lazy = twins(spec, B, monkeypatch)
    for m in (eager, lazy):
        m.initialize(9)
    for s in range(150):
        words, ww, labels, iw, _ = random_batch(spec, rs, B, zipf=True)
        for m in (eager, lazy):
            m.step(ca.Batch(words, labels, ww, iw), 1e-2)
        if s == 70:
            E = eager.get_param("entity_representations-representations") * 1.5
            for m in (eager, lazy):
                m.set_param("entity_representations-representations", E)
    same_everywhere(eager, lazy, "sparse_adam")


def test_tables_smaller_than_the_batch_stay_eager():
    """The bench shape (|V| = 50 k rows against 512 k entries) keeps its dense row passes: nothing to be lazy about."""
    spec = dict(num_words=50, num_entities=60, word_dim=8, entity_dim=8, window=4, num_random=3, update_method="sparse_adam")
    spec["lambda"] = 0.01
    m = gpu_model(spec, 64)
    m.initialize(1)
    rs = np.random.RandomState(0)
    words, ww, labels, iw, ids = random_batch(spec, rs, 64)
    m.step(ca.Batch(words, labels, ww, iw), 1e-3, entity_ids=ids)
    m.profile_enable(True)
    m.step(ca.Batch(words, labels, ww, iw), 1e-3, entity_ids=ids)
    assert not any(k.startswith("lazy_") for k in m.profile())


@pytest.mark.parametrize("lazy", [1, 0])
@pytest.mark.parametrize("method,dims", [("sparse_adam", (256, 256)), ("sgd", (256, 64)), ("adagrad", (64, 256)),
                                         ("sparse_adam", (12, 16)), ("sparse_adam", (300, 256)), ("adagrad", (7, 5))])
def test_entry_walk_equals_the_list_walk(method, dims, lazy, monkeypatch):
    """Tables larger than the batch: the rows with entries are done by walking the sorted entries (update.hip
    entry_walk_kernel, rows up to one wave wide) instead of a list of rows. Same sums in the same order: bit-identical to the
    three-launch form, which walks the list. The batches mix rows of one or two entries with rows of 30-64 (they straddle the
    64-position ranges of the waves) and a few of several hundred (chunk tree, finished by the chunk-only launch)."""
    spec = dict(num_words=6000, num_entities=9000, word_dim=dims[0], entity_dim=dims[1], window=10, num_random=4,
                nonlinearity="hard_tanh", batch_norm=True, update_method=method)
    spec["lambda"] = 0.01
    B = 512
    rs = np.random.RandomState(21)
    params = random_params(spec, rs)
    batches = []
    for _ in range(4):
        words, ww, labels, iw, ids = random_batch(spec, rs, B)
        u = rs.rand(words.size)
        words = np.where(u < 0.06, 7, np.where(u < 0.20, 100 + (rs.randint(0, 14, words.size)), words)).astype(words.dtype)
        u = rs.rand(ids.size)
        ids = np.where(u < 0.08, 3, np.where(u < 0.3, 500 + rs.randint(0, 12, ids.size), ids)).astype(ids.dtype)
        batches.append((words, ww, labels, iw, ids))
    cw, ce = np.bincount(batches[0][0]), np.bincount(batches[0][4])
    assert cw.max() > 256 and ce.max() > 128 and ((cw > 30) & (cw <= 64)).sum() >= 8 and ((ce > 30) & (ce <= 64)).sum() >= 8
    monkeypatch.setenv("NVSM_LAZY_DECAY", str(lazy))
    monkeypatch.setenv("NVSM_LAZY_MIN_MB", "0")
    monkeypatch.setenv("NVSM_ENTRY_WALK_MIN", "0")          # batches this small take the list walk otherwise
    results = []
    try:
        for one_launch in (1, 0):
            ca._lib.check(ca.lib().nvsm_debug_set_table_pass_form(one_launch))
            g = gpu_model(spec, B)
            load_params(g, params, True)
            for s, (words, ww, labels, iw, ids) in enumerate(batches):
                if s % 2:
                    g.step(ca.Batch(words, labels, ww, iw), 0.001, entity_ids=ids)
                else:
                    g.compute_cost(ca.Batch(words, labels, ww, iw), ids); g.compute_gradients(); g.update(0.001)
            results.append({p: g.get_param(p) for p in list(PARAMS) + STATE[method]})
    finally:
        ca._lib.check(ca.lib().nvsm_debug_set_table_pass_form(1))
    for p in results[0]:
        np.testing.assert_array_equal(results[0][p], results[1][p], err_msg=p)
