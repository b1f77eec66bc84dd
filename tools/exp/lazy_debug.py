"""first step at which the lazy twin departs from the eager one, and where (debug aid for tests/test_gpu_lazy.py)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cunvsm_amd as ca
from tests.helpers import PARAMS, gpu_model, load_params, random_batch, random_params

method, lam, bn, dims = "sparse_adam", 0.02, True, (12, 16)
if len(sys.argv) > 1: method = sys.argv[1]
if len(sys.argv) > 2: lam = float(sys.argv[2])
STATE = {"sgd": [], "adagrad": ["word_representations/a", "entity_representations/a"],
         "sparse_adam": ["word_representations/m", "word_representations/v", "entity_representations/m", "entity_representations/v"]}
spec = dict(num_words=3000, num_entities=5000, word_dim=dims[0], entity_dim=dims[1], window=3, num_random=2,
            nonlinearity="hard_tanh" if bn else "tanh", batch_norm=bn, update_method=method)
spec["lambda"] = lam
B = 40
rs = np.random.RandomState(5)
params = random_params(spec, rs)
os.environ["NVSM_LAZY_DECAY"] = "0"
eager = gpu_model(spec, B, sampler=ca.SAMPLER_DEVICE)
os.environ["NVSM_LAZY_DECAY"] = "1"; os.environ["NVSM_LAZY_MIN_MB"] = "0"
lazy = gpu_model(spec, B, sampler=ca.SAMPLER_DEVICE)
for m in (eager, lazy):
    m.initialize(3); load_params(m, params, True)
fused = os.environ.get("FUSED", "1") == "1"
for s in range(60):
    b = int(rs.choice([1, 7, 33, 40]))
    words, ww, labels, iw, ids = random_batch(spec, rs, b, zipf=True)
    lr = float(rs.choice([1e-3, 5e-3, 2e-2]))
    batch = ca.Batch(words, labels, ww, iw)
    costs = []
    for m in (eager, lazy):
        if fused and s % 2: m.step(batch, lr, entity_ids=ids)
        else:
            m.compute_cost(batch, ids); m.compute_gradients(); m.update(lr)
        costs.append(m.get_cost())
    bad = []
    for n in list(PARAMS) + STATE[method]:
        a, l = eager.get_param(n), lazy.get_param(n)
        if not np.array_equal(a, l):
            d = np.argwhere(a != l)
            bad.append((n, len(d), d[:4].tolist(), float(np.abs(a - l).max())))
    print(s, b, costs[0] == costs[1], bad, flush=True)
    if bad:
        uw = np.unique(words); ue = np.unique(ids)
        for n, cnt, d, mx in bad:
            rows = sorted({x[0] for x in np.argwhere(eager.get_param(n) != lazy.get_param(n)).tolist()})
            tab = uw if n.startswith("word") else ue
            print("  ", n, "rows", rows[:10], "touched this step:", [int(r in tab) for r in rows[:10]])
        break
