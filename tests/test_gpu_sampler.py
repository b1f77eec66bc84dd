"""-m gpu: the DEVICE negative sampler (NVSM_SAMPLER_DEVICE — the one bench.py and the trainer's --sampler device run on)
against the distribution the reference defines, and the device-side input validation.

Reference semantics (UniformLabelGenerator::generate, cpp/labels.cu:4-22 → generate_random_indexes,
include/cuNVSM/cuda_utils.h:24-33): per instance R = k + 1 document ids; slot 0 is the instance's label, slots 1..k are
independent draws, uniform over ALL documents [0, |D|) — a negative may repeat and may equal the label. The device
sampler is a counter-based hash keyed by (seed, rank, step, slot): not the reference's minstd_rand0 stream (that is
NVSM_SAMPLER_HOST_MINSTD, tested draw for draw in test_gpu_parity.py), but it must realise the same distribution.
"""
import numpy as np
import pytest

import cunvsm_amd as ca
from tests.helpers import PARAMS, gpu_model, random_batch, random_params

pytestmark = pytest.mark.gpu


def _spec(num_entities, k=16):
    return dict(num_words=500, num_entities=num_entities, word_dim=8, entity_dim=8, window=2, num_random=k)


def _draw(model, spec, B, rs, labels=None):
    words, ww, lab, iw, _ = random_batch(spec, rs, B, weighted=False)
    if labels is not None:
        lab = labels
    model.compute_cost(ca.Batch(words, lab, ww, iw))
    ids = model.get_tensor("entity_ids").astype(np.int64).reshape(B, spec["num_random"] + 1)
    return lab, ids


def test_device_sampler_slot0_range_and_uniformity_at_bench_shape():
    """B = 51 200, k = 16, |D| = 100 000 (BASELINE configs[1]): slot 0 == label, every id in range, χ² of the 819 200
    negatives against the uniform distribution over documents, and of every slot separately over 100 coarse bins."""
    nD, k, B = 100000, 16, 51200
    spec = _spec(nD, k)
    m = gpu_model(spec, B, sampler=ca.SAMPLER_DEVICE)
    m.initialize(1)
    labels, ids = _draw(m, spec, B, np.random.RandomState(5))
    np.testing.assert_array_equal(ids[:, 0], labels)
    neg = ids[:, 1:]
    assert neg.min() >= 0 and neg.max() < nD
    # χ² over documents: 819 200 draws into 100 000 cells (expected 8.192 per cell); dof = 99 999,
    # χ² ≈ N(dof, 2·dof): accept within 5 σ
    counts = np.bincount(neg.ravel(), minlength=nD).astype(np.float64)
    exp = neg.size / nD
    chi2 = ((counts - exp) ** 2 / exp).sum()
    dof = nD - 1
    assert abs(chi2 - dof) < 5.0 * np.sqrt(2.0 * dof), (chi2, dof)
    # per slot, 100 bins of 1000 documents: every one of the k streams is uniform on its own (dof 99, 5σ ≈ 70)
    for r in range(k):
        c = np.bincount(neg[:, r] // 1000, minlength=100).astype(np.float64)
        e = B / 100.0
        x2 = ((c - e) ** 2 / e).sum()
        assert x2 < 99 + 5.0 * np.sqrt(2 * 99.0), (r, x2)
    # negatives are drawn over ALL documents: some equal their instance's label at the expected rate k/|D| per instance
    hits = (neg == labels[:, None]).sum()
    assert 0 < hits < 40, hits            # expectation 8.2
    # no correlation between neighbouring slots / instances: lag-1 serial correlation of the flattened stream ≈ 0
    x = neg.ravel().astype(np.float64)
    x = (x - x.mean()) / x.std()
    assert abs((x[:-1] * x[1:]).mean()) < 5.0 / np.sqrt(x.size)
    assert abs((neg[:-1, 0].astype(np.float64) - neg[:-1, 0].mean()) @ (neg[1:, 0].astype(np.float64) - neg[1:, 0].mean())
               / (neg[:, 0].std() ** 2 * (B - 1))) < 5.0 / np.sqrt(B)


@pytest.mark.parametrize("nD", [1, 2, 7])
def test_device_sampler_tiny_document_sets(nD):
    """|D| ∈ {1, 2, 7}: ids stay in range, slot 0 is the label, every document is drawn about equally often."""
    k, B = 16, 4096
    spec = _spec(nD, k)
    m = gpu_model(spec, B, sampler=ca.SAMPLER_DEVICE)
    m.initialize(3)
    labels, ids = _draw(m, spec, B, np.random.RandomState(nD))
    np.testing.assert_array_equal(ids[:, 0], labels)
    neg = ids[:, 1:].ravel()
    assert neg.min() >= 0 and neg.max() < nD
    counts = np.bincount(neg, minlength=nD).astype(np.float64)
    exp = neg.size / nD
    if nD > 1:
        chi2 = ((counts - exp) ** 2 / exp).sum()
        assert chi2 < (nD - 1) + 6.0 * np.sqrt(2.0 * (nD - 1)) + 6.0, (counts, chi2)
    else:
        assert counts[0] == neg.size


def test_device_sampler_stream_depends_on_seed_step_and_rank_only():
    """Same (seed, rank, step) → the same ids whatever the batch content; a different step, seed or rank → different ids."""
    nD, k, B = 100000, 16, 2048
    spec = _spec(nD, k)
    rs = np.random.RandomState(0)

    def model(seed, rank=0, world=1):
        m = gpu_model(spec, B, sampler=ca.SAMPLER_DEVICE, world_size=world, rank=rank)
        m.initialize(seed)
        return m

    a, b = model(1), model(1)
    _, ia0 = _draw(a, spec, B, rs)
    _, ib0 = _draw(b, spec, B, rs)                     # other words / labels, same (seed, rank, step 0)
    np.testing.assert_array_equal(ia0[:, 1:], ib0[:, 1:])
    _, ia1 = _draw(a, spec, B, rs)                     # step 1
    _, ib1 = _draw(b, spec, B, rs)
    np.testing.assert_array_equal(ia1[:, 1:], ib1[:, 1:])
    same = (ia0[:, 1:] == ia1[:, 1:]).mean()
    assert same < 1e-3, same                           # chance level is 1 / |D| = 1e-5
    c = model(2)
    _, ic0 = _draw(c, spec, B, rs)
    assert (ia0[:, 1:] == ic0[:, 1:]).mean() < 1e-3
    # ranks of one data-parallel job draw different negatives (compute_cost alone issues no collective without batch-norm)
    r0, r1 = model(1, 0, 2), model(1, 1, 2)
    _, i0 = _draw(r0, spec, B, rs)
    _, i1 = _draw(r1, spec, B, rs)
    assert (i0[:, 1:] == i1[:, 1:]).mean() < 1e-3
    np.testing.assert_array_equal(i0[:, 1:], ia0[:, 1:])      # rank 0 of a job = the single-GPU stream


# ---------------------------------------------------------------------------------------------
# device-side validation of the ids handed over the ABI (include/cunvsm_amd.h, "Index contract")
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sampler", [ca.SAMPLER_DEVICE, ca.SAMPLER_HOST_MINSTD])
@pytest.mark.parametrize("what,bad", [("word", 500), ("word", -1), ("word", 1 << 33), ("label", 100), ("label", -5),
                                      ("label", (1 << 32) + 3)])
def test_out_of_range_ids_are_reported_not_dereferenced(sampler, what, bad):
    spec = _spec(100, 4)
    B = 64
    m = gpu_model(spec, B, sampler=sampler)
    m.initialize(1)
    rs = np.random.RandomState(1)
    words, ww, labels, iw, _ = random_batch(spec, rs, B)
    good = ca.Batch(words.copy(), labels.copy(), ww, iw)
    if what == "word":
        words[17] = bad
    else:
        labels[9] = bad
    # (the error surfaces at the next synchronisation point: get_cost — or compute_cost itself under NVSM_DEBUG=1, which
    #  synchronises after every call)
    with pytest.raises(ca.NvsmError) as e:
        m.compute_cost(ca.Batch(words, labels, ww, iw))
        m.get_cost()
    assert e.value.status == 1 and ("word id" if what == "word" else "document id") in str(e.value)
    # the handle stays usable, the error is reported once, and the tables were not corrupted
    m.compute_cost(good)
    assert np.isfinite(m.get_cost())
    m.compute_gradients()
    m.update(1e-3)
    for n in ("word_representations-representations", "entity_representations-representations"):
        assert np.isfinite(m.get_param(n)).all()


def test_out_of_range_explicit_entity_ids_and_fused_step():
    spec = _spec(100, 4)
    B = 32
    m = gpu_model(spec, B, sampler=ca.SAMPLER_DEVICE)
    m.initialize(1)
    rs = np.random.RandomState(2)
    words, ww, labels, iw, ids = random_batch(spec, rs, B)
    ids = ids.copy()
    ids[7] = 100
    with pytest.raises(ca.NvsmError) as e:            # at the cost read, or at the step itself under NVSM_DEBUG=1
        t = m.step_deferred(ca.Batch(words, labels, ww, iw), 1e-3, entity_ids=ids)
        m.deferred_cost(t)
    assert e.value.status == 1
    m.synchronize()                                   # reported once
    with pytest.raises(ValueError):
        m.compute_cost(ca.Batch(words[:-1], labels, ww[:-1], iw))          # wrong element count: caught by the binding
    with pytest.raises(ValueError):
        m.compute_cost(ca.Batch(words, labels, ww, iw), entity_ids=ids[:-1])


@pytest.mark.parametrize("method", ["sparse_adam", "adagrad"])
def test_good_steps_after_a_bad_id_in_a_fused_step_equal_a_fresh_handle(method):
    """A bad id reported by nvsm_step(cost) — whose host wait covers the loss word only, while the table passes and ordered
    sums of the same step still run on the side streams — must leave the handle as good as new: the hand-over counters of the
    ordered sums and of the one-launch table passes are cleared only once every stream is quiet (a counter zeroed under a
    running kernel loses its last arriver, and every later sum silently adds the wrong partials). Hot rows (Zipf word ids:
    both levels of the chunk tree) and a batch-norm model, so that every counter family is in use."""
    spec = dict(num_words=300, num_entities=200, word_dim=64, entity_dim=32, window=8, num_random=4, batch_norm=True,
                nonlinearity="hard_tanh", update_method=method)
    spec["lambda"] = 0.01
    B = 8192
    rs = np.random.RandomState(5)
    params = random_params(spec, rs)
    batches = [random_batch(spec, rs, B, zipf=True) for _ in range(4)]
    a, b = gpu_model(spec, B, sampler=ca.SAMPLER_DEVICE), gpu_model(spec, B, sampler=ca.SAMPLER_DEVICE)
    for m in (a, b):
        m.initialize(3)
    words, ww, labels, iw, ids = batches[0]
    for bad_what in ("word", "id"):
        w2, i2 = words.copy(), ids.copy()
        if bad_what == "word":
            w2[1234] = spec["num_words"] + 7
        else:
            i2[4321] = -3
        with pytest.raises(ca.NvsmError) as e:
            a.step(ca.Batch(w2, labels, ww, iw), 1e-3, entity_ids=i2, want_cost=True)
        assert e.value.status == 1
    # the broken steps did update `a` (bad ids are clamped to row 0): both handles start over from the same parameters and
    # optimiser state cannot be reset through the ABI, so the comparison handle replays the same (clamped) steps without error
    w2 = words.copy(); w2[1234] = 0
    i2 = ids.copy(); i2[4321] = 0
    b.step(ca.Batch(w2, labels, ww, iw), 1e-3, entity_ids=ids, want_cost=True)
    b.step(ca.Batch(words, labels, ww, iw), 1e-3, entity_ids=i2, want_cost=True)
    for (w, wwt, l, iwt, i) in batches[1:] * 3:
        ca_ = a.step(ca.Batch(w, l, wwt, iwt), 1e-3, entity_ids=i, want_cost=True)
        cb_ = b.step(ca.Batch(w, l, wwt, iwt), 1e-3, entity_ids=i, want_cost=True)
        assert ca_ == cb_
    for n in PARAMS:
        np.testing.assert_array_equal(a.get_param(n), b.get_param(n))


def test_debug_mode_reports_non_finite_values(monkeypatch):
    """NVSM_DEBUG=1 = the reference's debug build (CHECK_MATRIX, cpp/objective.cu:134,152): a NaN in a table is reported by
    the compute_cost that meets it; without the variable the same call sequence returns a NaN cost silently."""
    spec = _spec(50, 4)
    B = 16
    rs = np.random.RandomState(3)
    words, ww, labels, iw, _ = random_batch(spec, rs, B)
    monkeypatch.setenv("NVSM_DEBUG", "1")
    m = gpu_model(spec, B, sampler=ca.SAMPLER_DEVICE)
    m.initialize(1)
    m.compute_cost(ca.Batch(words, labels, ww, iw))      # clean: passes
    m.compute_gradients()
    W = m.get_param("word_representations-representations")
    W[int(words[0]) * spec["word_dim"]] = np.nan
    m.set_param("word_representations-representations", W)
    with pytest.raises(ca.NvsmError) as e:
        m.compute_cost(ca.Batch(words, labels, ww, iw))
    assert e.value.status == 3 and "non-finite" in str(e.value)
    monkeypatch.delenv("NVSM_DEBUG")
    q = gpu_model(spec, B, sampler=ca.SAMPLER_DEVICE)
    q.initialize(1)
    q.set_param("word_representations-representations", W)
    q.compute_cost(ca.Batch(words, labels, ww, iw))
    q.get_cost()            # no error without NVSM_DEBUG (the clamped sigmoid may even hide the NaN from the cost)
