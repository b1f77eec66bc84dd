"""Training trajectory with the split-bf16 projection kernels against the exact-fp32 kernels: the same 300 steps (device sampler,
same seed, fresh Zipf batches with a learnable document-word structure) with NVSM_GEMM_SPLIT=6, 9 and 0, and with the exact-fp32
arithmetic in another summation order (the yardstick for what roundoff alone does to a trajectory), each in a process of its own;
prints the loss every 25 steps and the largest relative differences of the curves.
   python tools/exp/split_trajectory.py"""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SCRIPT = r"""
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
import cunvsm_amd as ca
from tests.helpers import gpu_model
spec = dict(num_words=20000, num_entities=30000, word_dim=300, entity_dim=256, window=10, num_random=16, nonlinearity="hard_tanh",
            batch_norm=True, bias_negative_samples=False, update_method="sparse_adam", **{"lambda": 0.01})
B = 16384
m = gpu_model(spec, B, sampler=ca.SAMPLER_DEVICE)
m.initialize(7)
rs = np.random.RandomState(5)
costs = []
for step in range(300):
    labels = rs.randint(0, spec["num_entities"], B).astype(np.int64)
    words = ((labels[:, None] * 13 + (rs.zipf(1.3, (B, spec["window"])) %% 40)) %% spec["num_words"]).astype(np.int64).ravel()
    c = m.step(ca.Batch(words, labels, np.ones(B * spec["window"], np.float32), np.ones(B, np.float32)), 0.002, want_cost=True)
    costs.append(float(c))
print("RESULT " + json.dumps(costs))
"""

def run(split, **extra):
    env = dict(os.environ, NVSM_GEMM_SPLIT=split, **extra)
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], env=env, capture_output=True, text=True, cwd=ROOT, timeout=1200)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert r.returncode == 0 and line, r.stderr[-2000:]
    return json.loads(line[-1][7:])

a, b = run("6"), run("0")
# the yardstick: the exact-fp32 arithmetic itself with another summation order (the 128 x 128 tiled kernels instead of the
# LDS-stationary ones) — two fp32 runs that differ by roundoff only
c = run("0", NVSM_GEMM_TSTAT="0")
d = run("9")
for i in list(range(0, 300, 25)) + [299]:
    print("step %3d   six %.6f   nine %.6f   exact %.6f   exact, tiled %.6f" % (i, a[i], d[i], b[i], c[i]))
diff = lambda x, y: max(abs(p - q) / abs(q) for p, q in zip(x, y))
print("largest relative difference of the curves: six vs exact %.3e, nine vs exact %.3e, exact vs exact-tiled %.3e, six vs nine %.3e"
      % (diff(a, b), diff(d, b), diff(c, b), diff(a, d)))
