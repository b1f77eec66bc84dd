# experiment: pairwise errors of the dense gradients between the split-bf16 path, the exact-fp32 twin and the fp64 oracle
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from tests.test_gpu_fullsize import *
def run(method, Bp, seed, scale):
    spec = dict(SPEC, update_method=method)
    rs = np.random.RandomState(seed)
    params = random_params(spec, rs)
    params[PARAMS[2]] = (params[PARAMS[2]] * scale).astype(np.float32)
    o, a = oracle_model(spec, orc.F64), gpu_model(spec, Bp)
    os.environ["NVSM_GEMM_SPLIT"] = "0"; e = gpu_model(spec, Bp); del os.environ["NVSM_GEMM_SPLIT"]
    load_params(o, params, False)
    for m in (a, e): load_params(m, params, True)
    words = zipf_ids(rs, spec["num_words"], Bp * spec["window"])
    labels = rs.randint(0, spec["num_entities"], Bp).astype(np.int64)
    ww = rs.uniform(0.5, 1.5, Bp * spec["window"]).astype(np.float32)
    iw = rs.uniform(0.5, 1.5, Bp).astype(np.float32)
    ids = rs.randint(0, spec["num_entities"], (Bp, spec["num_random"] + 1)).astype(np.int64)
    ids[:, 0] = labels; ids = ids.ravel()
    o.forward(words, ww, ids, iw); o.backward()
    for m in (a, e):
        m.compute_cost(ca.Batch(words, labels, ww, iw), ids); m.compute_gradients()
    for name in ("proj", "grad_transform", "grad_bias", "grad_phrase"):
        x, y, z = a.get_tensor(name), o.get(name), e.get_tensor(name)
        extra = ""
        if name == "proj":
            d = np.abs(x.astype(np.float64) - y); extra = " max|split-oracle| %.3g, elements off by > 1e-3: %d ; exact: %d" % (d.max(), (d > 1e-3).sum(), (np.abs(z.astype(np.float64) - y) > 1e-3).sum())
        print(Bp, seed, scale, name, "split-oracle %.3g exact-oracle %.3g split-exact %.3g" % (rel_err(x, y), rel_err(z, y), rel_err(x, z)), extra)
for Bp, seed, scale in ((16384, 16384, 4), (16384, 1, 4), (16384, 2, 4), (16384, 16384, 1), (51200, 2024, 4)):
    run("sparse_adam", Bp, seed, scale)
