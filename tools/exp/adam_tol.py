import sys, numpy as np
sys.path.insert(0, '/root/repo')
import cunvsm_amd as ca
from oracle import nvsm_oracle as orc
from tests.helpers import PARAMS, gpu_model, load_params, oracle_model, random_params
import tests.test_gpu_fullsize as T
rs = np.random.RandomState(2024)
params = random_params(T.SPEC, rs)
params[PARAMS[2]] = (params[PARAMS[2]] * 4).astype(np.float32)
words, ww, labels, iw, ids = T.full_batch(rs, weighted=True)
for method in ("sparse_adam", "dense_adam", "full_adam", "sgd", "adagrad"):
    spec = dict(T.SPEC, update_method=method)
    o, g = oracle_model(spec, orc.F64), gpu_model(spec, T.B)
    load_params(o, params, False); load_params(g, params, True)
    o.forward(words, ww, ids, iw); o.backward()
    g.compute_cost(ca.Batch(words, labels, ww, iw), ids); g.compute_gradients()
    o.update(1e-3); g.update(1e-3)
    out = []
    for name in PARAMS:
        new_o, new_g, old = o.get(name), g.get_param(name).astype(np.float64), params[name].astype(np.float64)
        out.append("%.2e" % (np.linalg.norm(new_g - new_o) / np.linalg.norm(new_o - old)))
    print(method, out, flush=True)
    g.close()
