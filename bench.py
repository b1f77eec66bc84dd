#!/usr/bin/env python3
"""Benchmark of the NVSM training hot path on MI355X (BASELINE.json metric: n-gram windows/sec).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run (one rank per GPU,
rendezvous on 127.0.0.1, a free port), so both call forms work.

One "step" = one pass of the hot path (compute_cost → compute_gradients → update, negatives sampled on
device) over one synthetic batch of 51 200 windows per GPU that is already resident in HBM. Workload =
BASELINE.json configs[1]: |V| = 50k, |D| = 100k, d_word = 300, d_doc = 256, window 10, 16 negatives,
batch 51 200, hard_tanh + batch-norm, Adam (sparse_adam; --update-method selects the others), λ = 1e-2,
lr = 1e-3, Zipf(1) word ids, uniform document ids, all weights 1. `value` is the weak-scaling figure (every rank
gets its own 51 200-window batch; the dense projection gradient and the batch-norm statistics are all-reduced over
RCCL each step); with N > 1 the line also carries `strong` — SURVEY.md §8d row 3, the 51 200-window batch split
51 200 / N per rank — measured in the same run. With N = 1 it also carries `value_readback_every_step` (the loss
read back after every step, as the reference's loop does, cpp/main.cu:427-444) and `value_host_batches` (page-locked
host batches handed over each step, PCIe inclusive) — never used for `value`.

Rank 0 prints ONE JSON line. `roofline` is for the document-embedding gather + loss kernel (the largest HBM
gather of the step), timed with HIP events on the engine's stream inside the timed region; `kernel_breakdown`
comes from a second, untimed pass with events around every kernel group (they cost ≈5 % of a step, so they stay
out of the timed region); `cpu_baseline` is the CPU oracle (fp32, OpenMP) timed on this box's host cores on a
bounded sample of the same workload (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
F32_MFMA_PEAK_TFLOPS = 157.3


def zipf_ids(rs, n, size):
    p = 1.0 / np.arange(1, n + 1)
    p /= p.sum()
    return rs.choice(n, size=size, p=p).astype(np.int64)


# --config presets: BASELINE.json configs[1] is the bench line; configs[3] / configs[4] are parity cases
# (tests/test_gpu_configs.py) that can also be timed here as secondary figures.
PRESETS = {
    "nvsm": dict(),
    "lse_small": dict(num_words=200000, word_dim=128, batch=4096, nonlinearity="tanh", batch_norm=0,
                      bias_negative_samples=1, update_method="adagrad", lr=1e-2),
    "large_tables": dict(num_words=500000, num_entities=2000000),
}


def workload(args):
    wl = dict(num_words=50000, num_entities=100000, word_dim=300, entity_dim=256, window=10, num_random=16, batch=51200,
              nonlinearity="hard_tanh", batch_norm=1, bias_negative_samples=0, lr=1e-3, update_method="sparse_adam")
    wl.update(PRESETS[args.config])
    for k, v in (("num_words", args.num_words), ("num_entities", args.num_entities), ("batch", args.batch),
                 ("update_method", args.update_method), ("word_dim", getattr(args, "word_dim", None))):
        if v is not None:                      # explicit flags win over the preset
            wl[k] = v
    args.update_method = wl.pop("update_method")
    return wl


def algorithmic_bytes(kernel, wl, method):
    """Algorithmic HBM bytes per launch of each kernel group (SURVEY.md §8d per-window figures
    x the windows one launch processes). Index/weight traffic (<1 %) is excluded as in the survey."""
    B, w, dw, de, R = wl["batch"], wl["window"], wl["word_dim"], wl["entity_dim"], wl["num_random"] + 1
    nV, nD = wl["num_words"], wl["num_entities"]
    F = 4
    word_gather = B * w * dw * F              # 12 000 B / window
    ent_gather = B * R * de * F               # 17 408 B / window
    table = {
        "gather_mean_words": word_gather + B * dw * F,
        "loss_fused": ent_gather + 3 * B * de * F,
        "gemm_fwd": None, "gemm_bwd_x": None, "gemm_bwd_T": None,
        # row passes: gather of the gradient source rows + read/write of the table (+ state) rows
        "row_pass_entities": ent_gather + 2 * nD * de * F * (2 if method != "sgd" else 1),
        "row_pass_words": word_gather + 2 * nV * dw * F * (2 if method != "sgd" else 1),
        "row_pass_words_mv": word_gather + 2 * nV * dw * F,
        "row_pass_words_u": word_gather + 2 * nV * dw * F,
        "adam_u_words": word_gather + B * dw * F,
        "bn_backward": 3 * B * de * F,
    }
    return table.get(kernel)


# kernel group (engine profiler name) -> rocprofv3 kernel name prefix, for the PMC traffic figures kept under profiles/
PMC_KERNEL = {"loss_fused": "loss_rows_kernel", "row_pass_entities": "table_pass_kernel<4, 1, 3",
              "row_pass_words_mv": "table_pass_kernel<4, 0, 2", "row_pass_words_u": "table_pass_kernel<4, 0, 0",
              "gather_mean_words": "gather_mean_kernel", "adam_u_words": "adam_u_kernel"}


def workload_signature(wl, method, uniform_words):
    """What a PMC summary must have been taken on to say anything about this run's kernels."""
    return "V%d_D%d_dw%d_de%d_w%d_k%d_B%d_%s_%s" % (wl["num_words"], wl["num_entities"], wl["word_dim"], wl["entity_dim"],
                                                     wl["window"], wl["num_random"], wl["batch"], method,
                                                     "uniform" if uniform_words else "zipf")


def pmc_traffic(kernel, signature):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary (profiles/*_hbm_pmc.json,
    written by tools/profile_round.sh from separate --pmc FETCH_SIZE / WRITE_SIZE passes of this same command).
    Counters cannot be read from inside the process. None unless that summary was taken on exactly this workload
    (its "workload" key equals `signature`): a figure measured on another configuration says nothing about this one."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_pmc.json")))
    pref = PMC_KERNEL.get(kernel)
    if not files or not pref:
        return None, None
    for path in reversed(files):                 # the newest summary taken on this workload
        with open(path) as f:
            js = json.load(f)
        if js.get("workload") != signature:
            continue
        for name, e in js["kernels"].items():
            if name.startswith(pref):
                return int(e["fetch_bytes_corrected"] + e["write_bytes"]), os.path.basename(path)
        return None, None
    return None, None


def gemm_flops(wl):
    return 2.0 * wl["batch"] * wl["word_dim"] * wl["entity_dim"]


def cpu_baseline(args, wl, method):
    """Times the CPU oracle (test infrastructure, used here ONLY as the reported host baseline)."""
    from oracle import nvsm_oracle as orc
    from tests.helpers import METHODS
    m, mode = METHODS[method]
    cfg = orc.make_config(wl["num_words"], wl["num_entities"], wl["word_dim"], wl["entity_dim"], wl["window"],
                          wl["num_random"], batch_norm=bool(wl["batch_norm"]),
                          nonlinearity=orc.HARD_TANH if wl["nonlinearity"] == "hard_tanh" else orc.TANH, clip_sigmoid=True,
                          bias_negative_samples=bool(wl["bias_negative_samples"]), lambda_=1e-2, update_method=m, adam_mode=mode)
    model = orc.Model(cfg, orc.F32)
    rng = orc.Rng(1)
    model.initialize(rng)
    rs = np.random.RandomState(1)
    B = wl["batch"]
    steps, warm = args.cpu_steps, 1
    t_total = 0.0
    for s in range(warm + steps):
        words = zipf_ids(rs, wl["num_words"], B * wl["window"])
        labels = rs.randint(0, wl["num_entities"], B).astype(np.int64)
        ww = np.ones(B * wl["window"], np.float32)
        iw = np.ones(B, np.float32)
        t0 = time.perf_counter()
        ids = rng.generate_labels(labels, wl["num_entities"], wl["num_random"])      # host sampling, as the reference
        model.forward_native(words, ww, ids, iw)
        model.backward()
        model.update(wl["lr"])
        model.get_cost()
        dt = time.perf_counter() - t0
        if s >= warm:
            t_total += dt
    return {"value": B * steps / t_total, "unit": "windows/s", "cores": orc.lib().orc_num_threads(), "kind": "port",
            "sample": "%d full steps (batch %d) of the fp32 OpenMP oracle after %d warm-up, incl. host negative sampling; OpenMP team = "
                      "the CPUs the process may use (affinity mask capped by the cgroup quota), %d hardware threads visible"
                      % (steps, B, warm, os.cpu_count() or 0)}


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: re-execute under torch.distributed.run, one rank per GPU,
    rendezvous on 127.0.0.1 and a free port. Rank 0's JSON line goes to this process's stdout."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="nvsm", choices=sorted(PRESETS), help="nvsm = BASELINE configs[1] (the bench line)")
    ap.add_argument("--update-method", default=None, choices=["sgd", "adagrad", "sparse_adam", "dense_adam", "full_adam"])
    ap.add_argument("--batch", type=int, default=None, help="windows per GPU per step (default 51200)")
    ap.add_argument("--num-words", type=int, default=None)
    ap.add_argument("--num-entities", type=int, default=None)
    ap.add_argument("--word-dim", type=int, default=None, help="experiments (row alignment): d_word other than the config's 300")
    ap.add_argument("--strong-scaling", action="store_true", help="make the strong split (51 200 / N windows per rank, SURVEY §8d row 3) "
                    "the headline `value` instead of the weak one; both are always measured and reported when N > 1")
    ap.add_argument("--test-shared-gpu", action="store_true", help="test of the N > 1 control flow on a 1-GPU box: every rank on "
                    "device 0, gloo rendezvous, all-reduces through the host-callback transport (not a measurement)")
    ap.add_argument("--launch-check", action="store_true", help="rendezvous only (gloo, no GPU): proves that the N-rank launch works")
    ap.add_argument("--uniform-words", action="store_true", help="uniform instead of Zipf(1) word ids (worst case for caches)")
    ap.add_argument("--host-batches", action="store_true", help="hand host buffers over each step in the MAIN timed region too")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the read-back / host-batch / strong-scaling legs")
    ap.add_argument("--cpu-steps", type=int, default=30, help="full-size steps of the CPU oracle timed for cpu_baseline (≈0.3 s each on the "
                    "16 CPUs the GPU box grants the process)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sequential", action="store_true", help="compute_cost / compute_gradients / update as separate calls on one stream "
                    "(un-overlapped per-kernel timings) instead of the fused multi-stream nvsm_step")
    ap.add_argument("--no-profile", action="store_true", help="no HIP events at all (no roofline / breakdown in the output)")
    ap.add_argument("--profile-all", action="store_true", help="events around every kernel group inside the timed region (≈5 %% slower)")
    ap.add_argument("--gate-us", type=int, default=0, help="profiling aid: a spin kernel of this many microseconds in front of every "
                    "step, so that the host has queued the step before the GPU starts it (the timeline then shows the GPU-side schedule)")
    ap.add_argument("--read-cost-every", type=int, default=0, help="read the loss back every n steps (0 = never inside the timed region)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))

    import torch
    if args.launch_check:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            t = torch.ones(1)
            dist.all_reduce(t)
            n = int(t.item())
            dist.barrier()
            dist.destroy_process_group()
        else:
            n = 1
        if rank == 0:
            print(json.dumps({"launch_check": n, "n_gpus": world}), flush=True)
        return

    import cunvsm_amd as ca
    from cunvsm_amd.model import comm_unique_id

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if args.test_shared_gpu:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit("--gpus %d but only %d GPU(s) are visible (--test-shared-gpu exercises the N-rank control flow on one)"
                         % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if args.test_shared_gpu else "nccl", rank=rank, world_size=world)

    wl = workload(args)
    method = args.update_method
    cfg = ca.default_config(num_words=wl["num_words"], num_entities=wl["num_entities"], word_repr_size=wl["word_dim"],
                            entity_repr_size=wl["entity_dim"], window_size=wl["window"],
                            num_random_entities=wl["num_random"], batch_normalization=wl["batch_norm"],
                            nonlinearity=wl["nonlinearity"], clip_sigmoid=1,
                            bias_negative_samples=wl["bias_negative_samples"], regularization_lambda=1e-2, update_method=method,
                            max_batch_size=wl["batch"], device=local_rank, sampler=ca.SAMPLER_DEVICE,
                            world_size=world, rank=rank, sync_batch_norm=1)
    model = ca.Model(cfg)
    model.initialize(1)                     # --seed 1 (scripts/functions.sh:393); identical replicas on every rank
    transport, comm_ranks = "single", 0
    if world > 1:
        # the engine's own RCCL communicator (all-reduces on its streams, no host round trip); if it cannot be built on
        # this node, fall back to torch.distributed through the host-callback transport so that the run still completes
        ok = torch.zeros(1, device="cpu" if args.test_shared_gpu else "cuda")
        try:
            if args.test_shared_gpu:
                raise RuntimeError("shared-GPU test: RCCL cannot put two ranks on one device")
            obj = [comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(obj, src=0)
            model.comm_init(obj[0])
            ok += 1
        except Exception as e:            # noqa: BLE001
            sys.stderr.write("rank %d: nvsm_comm_init failed (%s)\n" % (rank, e))
        dist.all_reduce(ok)
        if int(ok.item()) == world:
            transport = "rccl"
            comm_ranks = model.comm_size()      # ncclCommCount
        else:
            from cunvsm_amd import dp
            if args.test_shared_gpu:
                model.set_allreduce_callback(dp.torch_allreduce(dist))
                transport = "torch.distributed(gloo) via host callback"
            else:
                model.set_allreduce_callback(dp.torch_allreduce_device(dist, torch.device("cuda", local_rank)))
                transport = "torch.distributed(nccl) via host callback"

    # synthetic batches, resident in HBM before the timed region
    w = wl["window"]
    dev = torch.device("cuda", local_rank)
    pinned_keep = []

    def make_pool(B, seed, host):
        rs = np.random.RandomState(seed + rank)
        pool = []
        for _ in range(4):
            words = (rs.randint(0, wl["num_words"], B * w).astype(np.int64) if args.uniform_words
                     else zipf_ids(rs, wl["num_words"], B * w))
            labels = rs.randint(0, wl["num_entities"], B).astype(np.int64)
            if host:        # page-locked host buffers, as the trainer's (and the reference's) batches are
                pins = [ca.model.pinned_copy(x) for x in (words, labels, np.ones(B * w, np.float32), np.ones(B, np.float32))]
                pinned_keep.append(pins)
                pool.append(ca.Batch(pins[0].array, pins[1].array, pins[2].array, pins[3].array))
            else:
                pool.append(ca.Batch(torch.from_numpy(words).to(dev), torch.from_numpy(labels).to(dev),
                                     torch.ones(B * w, dtype=torch.float32, device=dev),
                                     torch.ones(B, dtype=torch.float32, device=dev)))
        return pool

    B = wl["batch"]
    pool = make_pool(B, 1234, args.host_batches)
    lr = wl["lr"]

    def sync_all():
        model.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def run_steps(n, batches, read_every=0):
        for s in range(n):
            want = read_every > 0 and (s + 1) % read_every == 0
            if args.gate_us:
                model.debug_delay(args.gate_us)
            if args.sequential:
                model.compute_cost(batches[s % len(batches)])
                model.compute_gradients()
                model.update(lr)
                if want:
                    model.get_cost()
            else:
                model.step(batches[s % len(batches)], lr, want_cost=want)

    def timed(n, batches, read_every=0):
        """EXACTLY n steps between barrier + synchronize on both sides; MAX over ranks."""
        sync_all()
        t0 = time.perf_counter()
        run_steps(n, batches, read_every)
        model.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.test_shared_gpu else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    if world > 1:
        # communicator set-up (connections, first-use kernels) is lazy: two untimed steps take it out of the way even when
        # the caller asks for no warm-up steps
        run_steps(2, pool)
    run_steps(args.warmup, pool)

    # Timed region: HIP events around the two gather kernels only (the document gather + loss kernel and the word
    # gather-mean: four records per step). Events around every kernel group (~50 records per step) cost ≈5 % of the step,
    # so the full per-kernel breakdown comes from a second, untimed pass over the same batches (--profile-all puts it back
    # into the timed region).
    ROOFLINE_KERNEL, GATHER_KERNEL = "loss_fused", "gather_mean_words"
    model.profile_enable(not args.no_profile)
    model.profile_select(None if args.profile_all else ROOFLINE_KERNEL + "," + GATHER_KERNEL)
    model.profile_reset()
    elapsed = timed(args.steps, pool, args.read_cost_every)
    final_cost = model.get_cost()
    prof_timed = model.profile()
    prof = prof_timed
    breakdown_steps = args.steps
    if not args.no_profile and not args.profile_all:
        breakdown_steps = min(args.steps, 20)
        model.profile_select(None)
        model.profile_reset()
        run_steps(breakdown_steps, pool)          # every rank takes part (the collectives are in the step)
        sync_all()
        prof = model.profile()
    model.profile_enable(False)

    # ---- secondary legs, same model, same number of steps, no events -------------------------------------------
    extra = {}
    if not args.no_extra_legs and not args.sequential and not args.gate_us:
        if world == 1:
            # (a) loss read back after EVERY step, as iterate_data does (cpp/main.cu:427-444; SURVEY §8d "with the loss read
            #     back every step")
            run_steps(2, pool, 1)
            dt = timed(args.steps, pool, 1)
            extra["value_readback_every_step"] = round(B * args.steps / dt, 1)
            # (b) page-locked HOST batches handed over each step: PCIe-inclusive (never `value`)
            if not args.host_batches:
                hpool = make_pool(B, 1234, True)
                run_steps(3, hpool)
                dt = timed(args.steps, hpool)
                extra["value_host_batches"] = round(B * args.steps / dt, 1)
        elif B % world == 0:
            # (c) strong split of the same global batch: 51 200 / N windows per rank (BASELINE configs[2])
            Bs = B // world
            spool = make_pool(Bs, 4321, args.host_batches)
            run_steps(3, spool)
            dt = timed(args.steps, spool)
            extra["strong"] = {"value": round(B * args.steps / dt, 1), "unit": "windows/s", "ms_per_step": round(dt * 1e3 / args.steps, 4),
                               "scaling": "strong", "global_batch": B, "batch_per_rank": Bs, "steps": args.steps}

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = B * world * args.steps / elapsed
        weak = {"value": round(value, 1), "unit": "windows/s", "ms_per_step": round(ms_per_step, 4), "scaling": "weak",
                "global_batch": B * world, "batch_per_rank": B, "steps": args.steps}
        scaling = "weak"
        if args.strong_scaling and "strong" in extra:
            value, ms_per_step, scaling = extra["strong"]["value"], extra["strong"]["ms_per_step"], "strong"
        # dominant kernel group and its roofline
        breakdown = {}
        for k, (ms, n) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
            if n == 0:
                continue
            avg = ms / n
            ent = {"avg_ms": round(avg, 4), "launches_per_step": round(n / breakdown_steps, 2)}
            ab = algorithmic_bytes(k, wl, method)
            if ab:
                ent["algorithmic_GBps"] = round(ab / (avg * 1e-3) / 1e9, 1)
            if k.startswith("gemm_"):
                ent["TFLOPs"] = round(gemm_flops(wl) / (avg * 1e-3) / 1e12, 1)
                ent["mfma_frac"] = round(ent["TFLOPs"] / F32_MFMA_PEAK_TFLOPS, 3)      # of the 157.3 TF/s fp32 MFMA peak
            breakdown[k] = ent
        # Roofline kernel: the document-embedding gather + loss kernel — the HBM gather the north star names, and the
        # largest kernel of the step that runs with nothing else next to it but the (tiny) side-stream sorts. In the fused
        # step the documents update / dT GEMM overlap the dx GEMM / words update on a second stream; their event-timed
        # durations (marked "overlapped") include the time they share the chip and are not per-kernel roofline figures.
        for k in (() if args.sequential else ("update_entities", "row_pass_entities", "gemm_bwd_T", "transform_update", "csr_entities", "csr_words")):
            if k in breakdown:
                breakdown[k]["overlapped"] = True
        have = lambda k: prof_timed.get(k, (0, 0))[1] > 0
        roofline = roofline_gather = None
        sig = workload_signature(wl, method, args.uniform_words)
        if have(ROOFLINE_KERNEL):
            dom = ROOFLINE_KERNEL
            ab = algorithmic_bytes(dom, wl, method)
            avg = round(prof_timed[dom][0] / prof_timed[dom][1], 4)      # HIP events inside the timed region
            ach = ab / (avg * 1e-3) / 1e9
            traffic, traffic_src = pmc_traffic(dom, sig)
            roofline = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                        "algorithmic_bytes_per_launch": ab, "avg_launch_ms": avg,
                        "bytes": "document gather B*(k+1)*d_doc*4 + pre/proj/dy 3*B*d_doc*4"}
            if have(GATHER_KERNEL):
                # SURVEY §8d's own figure: gather bytes only — (w*d_word + (k+1)*d_doc)*4 = 29 408 B per window at the NVSM
                # shape — over the two kernels that do the gathering (word gather-mean + document gather/loss)
                R = wl["num_random"] + 1
                gb = B * (w * wl["word_dim"] + R * wl["entity_dim"]) * 4
                t2 = avg + prof_timed[GATHER_KERNEL][0] / prof_timed[GATHER_KERNEL][1]
                ach2 = gb / (t2 * 1e-3) / 1e9
                roofline_gather = {"kernels": [GATHER_KERNEL, dom], "bound": "hbm", "achieved": round(ach2, 1), "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": round(ach2 / HBM_PEAK_GBS, 4), "traffic": None,
                                   "algorithmic_bytes_per_step": gb, "bytes_per_window": gb // B, "avg_ms": round(t2, 4),
                                   "doc_gather_only_frac": round(B * R * wl["entity_dim"] * 4 / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        out = {
            "metric": "n-gram windows/sec (batch=51200, NVSM config)" if args.config == "nvsm" and B == 51200
                      else "n-gram windows/sec (--config %s, batch=%d)" % (args.config, B), "value": round(value, 1), "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s synthetic |V|=%d |D|=%d d_word=%d d_doc=%d window=%d neg=%d batch=%d/GPU "
                                   "%s%s %s lambda=1e-2 lr=%g %s word ids, inputs %s, device negative sampler"
                                   % ("NVSM" if wl["batch_norm"] else "LSE", wl["num_words"], wl["num_entities"], wl["word_dim"],
                                      wl["entity_dim"], w, wl["num_random"], B,
                                      wl["nonlinearity"], "+BN" if wl["batch_norm"] else "", method, wl["lr"],
                                      "uniform" if args.uniform_words else "Zipf(1)",
                                      "handed over as page-locked host buffers" if args.host_batches else "resident in HBM"),
                       "global_batch": B * world if scaling == "weak" else B, "parallelism": "dp%d" % world, "update_method": method,
                       "collectives": transport, "comm_ranks": comm_ranks,
                       "inputs": "host" if args.host_batches else "hbm", "step": "sequential calls" if args.sequential else "fused nvsm_step",
                       "workload_signature": sig},
            "roofline": roofline,
            "roofline_gather": roofline_gather,
            "kernel_breakdown": breakdown,
            "kernel_breakdown_source": ("timed region" if args.profile_all else
                                        "separate untimed pass of %d steps with events around every kernel group" % breakdown_steps),
            "final_cost": round(float(final_cost), 6),
        }
        if world > 1:
            out["weak"] = weak
        out.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, wl, method)
            out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
