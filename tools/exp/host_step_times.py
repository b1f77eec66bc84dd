"""experiment: host wall-clock per model.step() call in a free-running loop (is the host ahead of the GPU?)"""
import os, sys, time, json
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np, torch
import cunvsm_amd as ca
import bench
host = "--host-batches" in sys.argv
class A: pass
args = A(); args.config = "nvsm"; args.num_words = args.num_entities = args.batch = args.update_method = args.word_dim = None
wl = bench.workload(args); B, w = wl["batch"], wl["window"]
cfg = ca.default_config(num_words=wl["num_words"], num_entities=wl["num_entities"], word_repr_size=wl["word_dim"],
                        entity_repr_size=wl["entity_dim"], window_size=w, num_random_entities=wl["num_random"],
                        batch_normalization=wl["batch_norm"], nonlinearity=wl["nonlinearity"], clip_sigmoid=1,
                        bias_negative_samples=wl["bias_negative_samples"], regularization_lambda=1e-2,
                        update_method=args.update_method, max_batch_size=B, device=0, sampler=ca.SAMPLER_DEVICE)
m = ca.Model(cfg); m.initialize(1)
dev = torch.device("cuda", 0); rs = np.random.RandomState(1); pool = []; keep = []
for i in range(4):
    words = bench.zipf_ids(rs, wl["num_words"], B * w); labels = rs.randint(0, wl["num_entities"], size=B).astype(np.int64)
    if host:
        pins = [ca.model.pinned_copy(x) for x in (words, labels, np.ones(B * w, np.float32), np.ones(B, np.float32))]
        keep.append(pins); pool.append(ca.Batch(pins[0].array, pins[1].array, pins[2].array, pins[3].array))
    else:
        pool.append(ca.Batch(torch.from_numpy(words).to(dev), torch.from_numpy(labels).to(dev),
                             torch.ones(B * w, dtype=torch.float32, device=dev), torch.ones(B, dtype=torch.float32, device=dev)))
for i in range(10): m.step(pool[i % 4], wl["lr"])
m.synchronize()
ts = [time.perf_counter()]
for i in range(40):
    m.step(pool[i % 4], wl["lr"]); ts.append(time.perf_counter())
m.synchronize(); tend = time.perf_counter()
d = [round((b - a) * 1e6) for a, b in zip(ts, ts[1:])]
print(json.dumps({"host_batches": host, "us_per_call": d, "total_ms": round((tend - ts[0]) * 1e3, 2)}))
