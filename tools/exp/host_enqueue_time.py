"""experiment: host time to queue one fused step (the GPU is parked behind a spin kernel meanwhile) against the GPU time of
a step: a step whose GPU time is below the host's enqueue time is host-bound.  usage: host_enqueue_time.py [bench flags]"""
import os, sys, time, json, subprocess
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import argparse
import numpy as np
import torch
import cunvsm_amd as ca
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="nvsm")
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--host-batches", action="store_true")
a = ap.parse_args()
class A: pass
args = A(); args.config = a.config; args.num_words = args.num_entities = args.batch = args.update_method = args.word_dim = None
wl = bench.workload(args)
B, w = wl["batch"], wl["window"]
cfg = ca.default_config(num_words=wl["num_words"], num_entities=wl["num_entities"], word_repr_size=wl["word_dim"],
                        entity_repr_size=wl["entity_dim"], window_size=w, num_random_entities=wl["num_random"],
                        batch_normalization=wl["batch_norm"], nonlinearity=wl["nonlinearity"], clip_sigmoid=1,
                        bias_negative_samples=wl["bias_negative_samples"], regularization_lambda=1e-2,
                        update_method=args.update_method, max_batch_size=B, device=0, sampler=ca.SAMPLER_DEVICE)
m = ca.Model(cfg); m.initialize(1)
dev = torch.device("cuda", 0)
rs = np.random.RandomState(1)
pool = []
keep = []
for i in range(4):
    words = bench.zipf_ids(rs, wl["num_words"], B * w)
    labels = rs.randint(0, wl["num_entities"], size=B).astype(np.int64)
    if a.host_batches:
        pins = [ca.model.pinned_copy(x) for x in (words, labels, np.ones(B * w, np.float32), np.ones(B, np.float32))]
        keep.append(pins)
        pool.append(ca.Batch(pins[0].array, pins[1].array, pins[2].array, pins[3].array))
    else:
        pool.append(ca.Batch(torch.from_numpy(words).to(dev), torch.from_numpy(labels).to(dev),
                             torch.ones(B * w, dtype=torch.float32, device=dev), torch.ones(B, dtype=torch.float32, device=dev)))
for i in range(10):
    m.step_deferred(pool[i % 4], wl["lr"])
m.synchronize()
t0 = time.perf_counter()
for i in range(a.steps):
    m.step_deferred(pool[i % 4], wl["lr"])
m.synchronize()
free = (time.perf_counter() - t0) / a.steps
m.debug_delay(int(a.steps * 1500 + 20000))       # park the GPU
t0 = time.perf_counter()
for i in range(a.steps):
    m.step_deferred(pool[i % 4], wl["lr"])
host = (time.perf_counter() - t0) / a.steps
m.synchronize()
print(json.dumps({"config": a.config, "host_batches": a.host_batches, "host_enqueue_us_per_step": round(host * 1e6, 1), "free_running_us_per_step": round(free * 1e6, 1)}))
