"""Bit-reproducibility of a long free-running run at the headline shape (stream layout, side-stream placement and the early loss
copy under load): N steps over rotating device batches and page-locked host batches, twice in separate processes, the loss
read back every 7th step; the parameter hashes and the losses must be identical.   python tools/exp/repro_long.py [steps]"""
import hashlib, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SCRIPT = r"""
import hashlib, json, sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
import cunvsm_amd as ca
from tests.helpers import PARAMS, gpu_model
steps = int(sys.argv[1])
spec = dict(num_words=50000, num_entities=100000, word_dim=300, entity_dim=256, window=10, num_random=16, nonlinearity="hard_tanh",
            batch_norm=True, bias_negative_samples=False, update_method="sparse_adam", **{"lambda": 0.01})
B = 51200
m = gpu_model(spec, B, sampler=ca.SAMPLER_DEVICE)
m.initialize(7)
rs = np.random.RandomState(5)
dev = torch.device("cuda", 0)
pool = []
for k in range(4):
    words = (rs.zipf(1.2, B * 10) %% spec["num_words"]).astype(np.int64)
    labels = rs.randint(0, spec["num_entities"], B).astype(np.int64)
    ww = rs.uniform(0.5, 1.5, B * 10).astype(np.float32); iw = rs.uniform(0.5, 1.5, B).astype(np.float32)
    if k %% 2 == 0:
        pool.append(ca.Batch(torch.from_numpy(words).to(dev), torch.from_numpy(labels).to(dev), torch.from_numpy(ww).to(dev), torch.from_numpy(iw).to(dev)))
    else:
        pins = [ca.model.pinned_copy(x) for x in (words, labels, ww, iw)]
        pool.append((pins, ca.Batch(pins[0].array, pins[1].array, pins[2].array, pins[3].array)))
costs = []
for s in range(steps):
    b = pool[s %% 4]
    b = b[1] if isinstance(b, tuple) else b
    c = m.step(b, 1e-3, want_cost=(s %% 7 == 3))
    if c is not None: costs.append(float(c))
h = hashlib.sha256()
for p in PARAMS:
    h.update(np.ascontiguousarray(m.get_param(p)).tobytes())
print("RESULT " + json.dumps({"params": h.hexdigest(), "costs": costs}))
"""
steps = sys.argv[1] if len(sys.argv) > 1 else "400"
out = []
for i in range(2):
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}, steps], capture_output=True, text=True, cwd=ROOT, timeout=1800)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert r.returncode == 0 and line, r.stderr[-3000:]
    out.append(json.loads(line[-1][7:]))
print("params", out[0]["params"][:16], out[1]["params"][:16], "equal" if out[0]["params"] == out[1]["params"] else "DIFFERENT")
print("losses", len(out[0]["costs"]), "equal" if out[0]["costs"] == out[1]["costs"] else "DIFFERENT", out[0]["costs"][:3], out[0]["costs"][-2:])
