"""Soak test of the last-arriver hand-overs (device_utils.h grid_sum_ordered; update.hip's one-launch table passes): their ordering
argument — s_waitcnt vmcnt(0) + relaxed agent-scope atomics — is specific to gfx90a / gfx942 / gfx950 and outside what the HIP
memory model promises, and nothing else in the suite runs them for more than twenty steps. Here: 2 000 fused steps at the per-rank
batch of the 8-GPU job (6 400 windows, the metric's dimensions, sparse Adam) and at the LSE recipe's 4 096 (|V| = 200 k, Adagrad),
(and 600 at the headline batch of 51 200 — gemm_split's mid-tile barrier, gemm_dt, the chunk order), twice from the same state —
parameters AND optimiser state bit-equal between the runs at every 500th (200th) step —, once more with the
table passes as three launches (no hand-over inside a pass): bit-equal again; and every arrival counter reads zero at the end."""
import numpy as np
import pytest

import cunvsm_amd as ca
from tests.helpers import PARAMS, gpu_model

pytestmark = pytest.mark.gpu

STEPS, EVERY = 2000, 500
STEPS_LARGE, EVERY_LARGE = 600, 200      # the headline batch: gemm_split's mid-tile barrier and wave priorities (round 6), gemm_dt, chunk order

SHAPES = {
    "b6400": (dict(num_words=50000, num_entities=100000, word_dim=300, entity_dim=256, window=10, num_random=16, nonlinearity="hard_tanh",
                   batch_norm=True, update_method="sparse_adam", **{"lambda": 0.01}), 6400, 1e-3,
              ["word_representations/m", "word_representations/v", "entity_representations/m", "entity_representations/v",
               "word_entity_mapping/s0_transform", "word_entity_mapping/s1_transform"]),
    "b51200": (dict(num_words=50000, num_entities=100000, word_dim=300, entity_dim=256, window=10, num_random=16, nonlinearity="hard_tanh",
                    batch_norm=True, update_method="sparse_adam", **{"lambda": 0.01}), 51200, 1e-3,
               ["word_representations/m", "word_representations/v", "entity_representations/m", "entity_representations/v",
                "word_entity_mapping/s0_transform", "word_entity_mapping/s1_transform"]),
    "lse4096": (dict(num_words=200000, num_entities=100000, word_dim=128, entity_dim=256, window=10, num_random=16, nonlinearity="tanh",
                     batch_norm=False, bias_negative_samples=True, update_method="adagrad", **{"lambda": 0.01}), 4096, 1e-2,
                ["word_representations/a", "entity_representations/a", "word_entity_mapping/s0_transform"]),
}


def _checksum(a):
    u = np.ascontiguousarray(a).view(np.uint32)
    return (int(np.bitwise_xor.reduce(u)), int(u.sum(dtype=np.uint64)))


def _run(spec, B, lr, state, one_launch):
    import torch
    ca._lib.check(ca.lib().nvsm_debug_set_table_pass_form(1 if one_launch else 0))
    try:
        m = gpu_model(spec, B, sampler=ca.SAMPLER_DEVICE)
        m.initialize(11)
        rs = np.random.RandomState(3)
        p = 1.0 / np.arange(1, spec["num_words"] + 1)
        p /= p.sum()
        pool = []
        for _ in range(8):
            words = rs.choice(spec["num_words"], size=B * spec["window"], p=p).astype(np.int64)
            labels = rs.randint(0, spec["num_entities"], B).astype(np.int64)
            pool.append(ca.Batch(torch.from_numpy(words).cuda(), torch.from_numpy(labels).cuda(),
                                 torch.ones(B * spec["window"], dtype=torch.float32, device="cuda"), torch.ones(B, dtype=torch.float32, device="cuda")))
        sums = []
        steps, every = (STEPS_LARGE, EVERY_LARGE) if B > 16384 else (STEPS, EVERY)
        for s in range(steps):
            m.step(pool[s % len(pool)], lr)
            if (s + 1) % every == 0:
                sums.append([_checksum(m.get_param(n)) for n in list(PARAMS) + state])
        cost = m.step(pool[0], lr, want_cost=True)
        counters = m.get_tensor("arrival_counters")
        m.close()
        return sums, cost, counters
    finally:
        ca._lib.check(ca.lib().nvsm_debug_set_table_pass_form(1))


@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_two_thousand_fused_steps_twice_and_as_three_launch_passes(shape):
    spec, B, lr, state = SHAPES[shape]
    a = _run(spec, B, lr, state, True)
    b = _run(spec, B, lr, state, True)
    c = _run(spec, B, lr, state, False)
    assert np.isfinite(a[1]) and a[1] > 0
    every = EVERY_LARGE if B > 16384 else EVERY
    assert len(a[0]) == (STEPS_LARGE // EVERY_LARGE if B > 16384 else STEPS // EVERY)
    for k in range(len(a[0])):
        assert a[0][k] == b[0][k], "step %d: two runs from the same state differ" % ((k + 1) * every)
        assert a[0][k] == c[0][k], "step %d: one-launch passes differ from the three-launch form" % ((k + 1) * every)
    assert a[1] == b[1] == c[1]
    for run in (a, b, c):
        assert not run[2].any(), "arrival counters left non-zero: %s" % run[2]
