"""What configs[3]'s full-size parity test measures (tests/test_gpu_configs.py::test_lse_small_batch_adagrad_full_size): the error of
the HIP path against the fp64 oracle after three Adagrad steps, as a fraction of the parameter change — per parameter, and the same
for an fp32 run of the ORACLE (what any fp32 implementation can promise). python tools/exp/lse_tol.py"""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import cunvsm_amd as ca
from oracle import nvsm_oracle as orc
from tests.helpers import PARAMS, gpu_model, load_params, oracle_model, random_params, zipf_ids

spec = dict(num_words=200000, num_entities=100000, word_dim=128, entity_dim=256, window=10, num_random=16,
            nonlinearity="tanh", batch_norm=False, bias_negative_samples=True, update_method="adagrad")
spec["lambda"] = 0.01
B, lr = 4096, 0.01
for seed in (404, 405, 406):
    rs = np.random.RandomState(seed)
    params = random_params(spec, rs)
    params[PARAMS[2]] = (params[PARAMS[2]] * 4).astype(np.float32)
    o, o32, g = oracle_model(spec, orc.F64), oracle_model(spec, orc.F32), gpu_model(spec, B)
    load_params(o, params, False); load_params(o32, params, False); load_params(g, params, True)
    w, k = spec["window"], spec["num_random"]
    for step in range(3):
        words = zipf_ids(rs, spec["num_words"], B * w)
        labels = rs.randint(0, spec["num_entities"], B).astype(np.int64)
        ww = rs.uniform(0.2, 2.0, B * w).astype(np.float32)
        iw = np.ones(B, np.float32)
        ids = rs.randint(0, spec["num_entities"], (B, k + 1)).astype(np.int64)
        ids[:, 0] = labels
        ids = ids.ravel()
        for m in (o, o32):
            m.forward(words, ww, ids, iw); m.backward(); m.update(lr)
        g.step(ca.Batch(words, labels, ww, iw), lr, entity_ids=ids)
    out = []
    for name in PARAMS:
        new_o, old = o.get(name), params[name].astype(np.float64)
        change = np.linalg.norm(new_o - old)
        out.append("%s hip %.2e oracle-fp32 %.2e" % (name.split("-")[0][:12], np.linalg.norm(g.get_param(name).astype(np.float64) - new_o) / change,
                                                    np.linalg.norm(np.asarray(o32.get(name), np.float64) - new_o) / change))
    print("seed", seed, " | ".join(out), flush=True)
    g.close()
