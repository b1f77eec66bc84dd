"""The loss kernel's time with the documents table warm / cold in the Infinity Cache, through the library itself:
compute_cost only (word gather, projection product, loss kernel: ~0.3 GB between two loss kernels, E = 102 MB stays cached)
against compute_cost with a 1 GB fill in between (E evicted, as the table updates of a training step evict it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import cunvsm_amd as ca
import bench

class A: pass
args = A(); args.config = "nvsm"; args.num_words = args.num_entities = args.batch = args.word_dim = None; args.update_method = None
wl = bench.workload(args)
B, w = wl["batch"], wl["window"]
nD = int(sys.argv[1]) if len(sys.argv) > 1 else wl["num_entities"]
cfg = ca.default_config(num_words=wl["num_words"], num_entities=nD, word_repr_size=wl["word_dim"], entity_repr_size=wl["entity_dim"],
                        window_size=w, num_random_entities=wl["num_random"], batch_normalization=1, nonlinearity="hard_tanh", clip_sigmoid=1,
                        bias_negative_samples=0, regularization_lambda=1e-2, update_method="sparse_adam", max_batch_size=B, device=0,
                        sampler=ca.SAMPLER_DEVICE)
m = ca.Model(cfg); m.initialize(1)
rs = np.random.RandomState(3); dev = torch.device("cuda", 0)
batch = ca.Batch(torch.from_numpy(bench.zipf_ids(rs, wl["num_words"], B * w)).to(dev), torch.from_numpy(rs.randint(0, nD, B).astype(np.int64)).to(dev),
                 torch.ones(B * w, dtype=torch.float32, device=dev), torch.ones(B, dtype=torch.float32, device=dev))
junk = torch.empty(256 << 20, dtype=torch.float32, device=dev)      # 1 GB
def run(cold, steps=20):
    for _ in range(3): m.compute_cost(batch)
    m.synchronize(); m.profile_enable(True); m.profile_select("loss_fused"); m.profile_reset()
    for i in range(steps):
        if cold: junk.fill_(float(i)); torch.cuda.synchronize()
        m.compute_cost(batch)
    m.synchronize(); ms, n = m.profile()["loss_fused"]; m.profile_enable(False)
    return ms / n * 1e3
print("|D| = %d: loss kernel, documents table warm in the Infinity Cache %.1f us; evicted by a 1 GB fill %.1f us" % (nD, run(False), run(True)))
