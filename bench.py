#!/usr/bin/env python3
"""Benchmark of the NVSM training hot path on MI355X (BASELINE.json metric: n-gram windows/sec, batch = 51 200).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run (one rank per GPU,
rendezvous on 127.0.0.1, a free port), so both call forms work.

One "step" = one pass of the hot path (compute_cost → compute_gradients → update, negatives sampled on
device) over one synthetic batch that is already resident in HBM. Workload = BASELINE.json configs[1]:
|V| = 50k, |D| = 100k, d_word = 300, d_doc = 256, window 10, 16 negatives, batch 51 200, hard_tanh + batch-norm, Adam
(sparse_adam; --update-method selects the others), λ = 1e-2, lr = 1e-3, Zipf(1) word ids, uniform document ids, all
weights 1.

Timing: W warm-up steps, then `--repeats` timed regions of EXACTLY K steps each, every region bracketed by a
barrier + synchronize on both sides and taken as the MAX over ranks; `value` comes from the MEDIAN region (all region
times are in the line: a 20-step region is 20 ms, one region would be a noisy sample).

N = 1: `value` = 51 200-window steps on one GPU. The line also carries
  * `per_rank_shapes` — the per-rank share of the 8-GPU metric (51 200 / N windows per rank) timed on
    this one GPU with an engine created for that batch size, and `strong_projection_8gpu` computed from it (compute only:
    no collective is in it);
  * `secondary` — the reference recipe's `full_adam` (scripts/functions.sh:395) and the uniform-word-id worst case;
  * `value_readback_every_step` (the loss read back after every step, as cpp/main.cu:427-444 does),
    `value_readback_one_step_late` (every step's loss read back, one step behind: cuNVSMTrainModel's protocol) and
    `value_host_batches` (page-locked host batches handed over each step, PCIe inclusive) — never used for `value`.
N > 1: the metric says batch = 51 200, so the headline `value` is the STRONG figure: the 51 200-window batch split
51 200 / N per rank (dense gradients and batch-norm statistics all-reduced over RCCL each step); the weak figure
(51 200 windows per rank) is measured in the same run and reported beside it (`weak`; `--weak-scaling` makes it the headline).

Rank 0 prints ONE JSON line. `roofline` is for the document-embedding gather + loss kernel (the largest HBM
gather of the step), timed with HIP events that ride on the kernel's own launch inside the timed regions;
`kernel_breakdown` comes from a second, untimed pass with events around every kernel group (they cost ≈5 % of a step, so
they stay out of the timed regions); `cpu_baseline` is the CPU oracle (fp32, OpenMP) timed on this box's host cores on a
bounded sample of the same workload (N = 1 only).
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
F32_MFMA_PEAK_TFLOPS = 157.3
BF16_MFMA_PEAK_TFLOPS = 2500.0        # dense (MI355X_MICROARCH.md)


def gemm_split_products(B=None):
    """Partial products of the split-bf16 projection kernels (0: the exact-fp32 MFMA kernels: NVSM_GEMM_SPLIT=0). Every batch size of
    this bench runs them — gemm_split / gemm_dt above 8 192 windows, gemm_rsplit / gemm_dtw at per-rank batches; which product of a
    step takes which kernel is in nvsm_describe (product_arithmetic below)."""
    v = int(os.environ.get("NVSM_GEMM_SPLIT", "6"))
    return v if v in (6, 9) else 0


def product_arithmetic(desc):
    """{gemm_fwd | gemm_bwd_x | gemm_bwd_T: partial products (0 = exact fp32 MFMAs)} from the engine's own description of a step
    (nvsm_describe: "forward <kernel> (...) | backward <kernel> (...) | dT <kernel> (...)")."""
    out = {}
    for key, tag in (("gemm_fwd", "forward "), ("gemm_bwd_x", "backward "), ("gemm_bwd_T", "dT ")):
        seg = [p for p in desc.split(" | ") if p.split(": ")[-1].startswith(tag) or p.startswith(tag)]
        out[key] = gemm_split_products() if seg and "bf16 planes" in seg[0] else 0
    return out


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


def algorithmic_bytes(kernel, wl, method, B, rows=None):
    """Algorithmic HBM bytes per launch of each kernel group (SURVEY.md §8d per-window figures
    x the windows one launch processes). Index/weight traffic (<1 %) is excluded as in the survey.
    rows: {"words": r, "entities": r} — the table rows a row pass reads and writes. A dense pass visits every row of its
    table; a lazily decayed table (kernels.h: the decay of the rows without entries stays pending) only the rows the batch
    touches, so the caller passes the touched-row counts for those."""
    w, dw, de, R = wl["window"], wl["word_dim"], wl["entity_dim"], wl["num_random"] + 1
    rows = rows or {}
    nV, nD = rows.get("words", wl["num_words"]), rows.get("entities", wl["num_entities"])
    F = 4
    word_gather = B * w * dw * F              # 12 000 B / window
    ent_gather = B * R * de * F               # 17 408 B / window
    # [rows][dim] arrays a pass over a table reads AND writes: the table itself + Adam's first (+ full_adam's second) moments; SGD and
    # Adagrad keep the table only (Adagrad's accumulator is one scalar per row)
    state = {"sgd": 1, "adagrad": 1, "sparse_adam": 2, "dense_adam": 2, "full_adam": 3}[method]
    table = {
        "gather_mean_words": word_gather + B * dw * F,
        "loss_fused": ent_gather + 3 * B * de * F,
        # the three projection products: their batch-sized operands and results, once each (the projection matrix is 0.3 MB)
        "gemm_fwd": B * (dw + de) * F,                     # phrase in, pre out
        "gemm_bwd_x": B * (3 * de + dw) * F,               # dy and pre in (batch-norm backward on the way), dx and gphrase out
        "gemm_bwd_T": B * (dw + de) * F,                   # phrase and dx in
        # row passes: gather of the gradient source rows + read/write of the table (+ state) rows
        "row_pass_entities": ent_gather + 2 * nD * de * F * state,
        "row_pass_words": word_gather + 2 * nV * dw * F * state,
        "row_pass_words_mv": word_gather + 2 * nV * dw * F,
        "row_pass_words_u": word_gather + 2 * nV * dw * F,
        "adam_u_words": word_gather + B * dw * F,
        "bn_backward": 3 * B * de * F,
    }
    return table.get(kernel)


def rows_visited(desc, touched, method, B, lr, world=1):
    """{"words": n, "entities": n} for the tables whose update passes visit only the rows a batch touches: lazily decayed tables
    (nvsm_describe says which), and SGD / Adagrad tables whose per-step decay factor 1 - (lambda / B_global) * lr rounds to 1.0f —
    the dense decay is then the identity in the reference's own fp32 arithmetic and the passes skip the rows without entries
    (update.hip: p_always). The LSE leg (lambda 1e-2, lr 1e-2, batch 4 096: 1 - 2.4e-8) is such a case."""
    rows = {}
    sl = np.float32(1e-2) / np.float32(B * max(1, world))
    identity = method in ("sgd", "adagrad") and np.float32(1.0 - float(sl) * float(lr)) == np.float32(1.0)
    for t, name in (("words", "words lazy decay"), ("entities", "documents lazy decay")):
        if touched and (name in desc or identity):
            rows[t] = touched[t]
    return rows


def step_kernel_groups(method):
    """The kernel groups one fused step runs, by update method (model.cpp update_words): what roofline_step adds up."""
    words = {"sparse_adam": ["row_pass_words_mv", "adam_u_words", "row_pass_words_u"]}.get(method, ["row_pass_words"])
    return ["gather_mean_words", "gemm_fwd", "loss_fused", "gemm_bwd_x", "gemm_bwd_T", "row_pass_entities"] + words


def step_roofline(wl, method, B, rows, ms_per_step, counter=None):
    """Step-level HBM figure: Σ algorithmic bytes of the step's kernels (and Σ counter bytes when a PMC pass of this workload is at
    hand) over the measured step time. The kernels of a step overlap on three streams, so this — not any one kernel's in-step
    fraction — is what the memory system delivered."""
    parts = {k: algorithmic_bytes(k, wl, method, B, rows) for k in step_kernel_groups(method)}
    ab = sum(parts.values())
    ach = ab / (ms_per_step * 1e-3) / 1e9
    out = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "ms_per_step": round(ms_per_step, 4), "algorithmic_bytes_per_step": ab,
           "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_bytes_by_kernel": parts,
           "bytes": "sum over the step's kernels of their algorithmic bytes (gathers, the batch-sized tensors each product reads and writes, "
                    "every table / state row the update passes visit); index and weight traffic (< 1 %) excluded"}
    if counter and counter.get("step_bytes"):
        a2 = counter["step_bytes"] / (ms_per_step * 1e-3) / 1e9
        out["counter_bytes"] = {"traffic": int(counter["step_bytes"]), "achieved": round(a2, 1), "frac": round(a2 / HBM_PEAK_GBS, 4),
                                "traffic_source": counter["source"], "traffic_measured_in_run": counter["in_run"],
                                "over_algorithmic": round(counter["step_bytes"] / ab, 3)}
    return out


# kernel group (engine profiler name) -> rocprofv3 kernel name prefix, for the PMC traffic figures kept under profiles/
PMC_KERNEL = {"loss_fused": "loss_rows_kernel", "row_pass_entities": "table_pass_wide_kernel<4, 1, 3",
              "row_pass_words_mv": "table_pass_wide_kernel<4, 0, 2", "row_pass_words_u": "table_pass_wide_kernel<4, 0, 0",
              "gather_mean_words": "gather_mean_kernel", "adam_u_words": "adam_u_kernel"}


# kernel group -> the source file that defines its kernel: a committed PMC summary only speaks for this run's kernel when that
# file is byte-for-byte what it was when the summary was taken (tools/rocprof_summary.py stores the hashes)
KERNEL_SOURCE = {"loss_fused": "loss_bn.hip", "row_pass_entities": "update.hip", "row_pass_words_mv": "update.hip",
                 "row_pass_words_u": "update.hip", "gather_mean_words": "gather_gemm.hip", "adam_u_words": "update.hip"}


def source_sha256(name):
    import hashlib
    path = os.path.join(ROOT, "cunvsm_amd", "csrc", name)
    try:
        with open(path, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()
    except OSError:
        return None


def workload_signature(wl, method, uniform_words, B):
    """What a PMC summary must have been taken on to say anything about this run's kernels."""
    return "V%d_D%d_dw%d_de%d_w%d_k%d_B%d_%s_%s" % (wl["num_words"], wl["num_entities"], wl["word_dim"], wl["entity_dim"],
                                                     wl["window"], wl["num_random"], B, method,
                                                     "uniform" if uniform_words else "zipf")


def pmc_prefix(kernel, walk=None):
    """rocprofv3 kernel-name prefix of a kernel group. The documents update of a table much larger than the batch walks the sorted
    entries (entry_walk_kernel) — table_pass_wide_kernel<4, 1, 3, ...> then only runs the few chunked rows (35 KB a launch: what
    round 5's large-tables line reported as the update's traffic)."""
    if kernel == "row_pass_entities":
        # (table_pass[_wide]_kernel<V, TABLE, KIND, ...>: TABLE 1 = the documents table, whatever the optimiser's row formula)
        return "entry_walk_kernel<4, 1," if walk == "entry_walk" else ("table_pass_wide_kernel<4, 1,", "table_pass_kernel<4, 1,")
    return PMC_KERNEL.get(kernel)


def pmc_in_run(flags, timeout=120):
    """HBM counter bytes measured IN THIS RUN: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (a pass each, with --kernel-trace
    only, as MI355X_MICROARCH.md prescribes) of this very script in its bare form (8 steps of the same workload on this box, no
    events, no extra legs). Returns {"kernels": per-launch table, "step_bytes", "steps", "source", "in_run": True} or None (no
    rocprofv3, a failed pass, or this process is itself running under a profiler)."""
    import shutil
    import subprocess
    import tempfile
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import rocprof_summary
    tmp = tempfile.mkdtemp(prefix="nvsm_pmc_", dir="/tmp")
    try:
        dbs = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "-d", out, "-o", "bench", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "6", "--warmup", "2", "--repeats", "1", "--no-extra-legs", "--no-cpu-baseline", "--no-profile"] + flags
            # (a counter pass of the bare bench takes 8-20 s; one that has not finished in two minutes is given up — subprocess.run kills it —
            #  and the line falls back to the committed summaries: the bench must never hang on its own instrumentation)
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
            found = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith(".db")]
            if r.returncode != 0 or not found:
                sys.stderr.write("in-run PMC pass %s failed (rc %d): %s\n" % (counter, r.returncode, r.stderr[-300:]))
                return None
            dbs[counter] = found[0]
        js = rocprof_summary.pmc_table(dbs["FETCH_SIZE"], dbs["WRITE_SIZE"])
        step_bytes, steps = rocprof_summary.step_traffic(js)
        return {"kernels": js, "step_bytes": step_bytes, "steps": steps, "in_run": True,
                "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (a pass each, --kernel-trace only) of this command's workload, "
                          "launched by this run on this box; FETCH_SIZE doubled (gfx950: MI355X_MICROARCH.md)"}
    except Exception as e:            # noqa: BLE001  (the bench line says "not measured" instead of failing)
        sys.stderr.write("in-run PMC failed: %s\n" % str(e)[:300])
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_committed_step(signature):
    """Step-level counter bytes from the newest committed summary of this workload — only if EVERY kernel source still hashes the same."""
    import glob
    for path in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_pmc.json")))):
        with open(path) as f:
            js = json.load(f)
        if js.get("workload") != signature:
            continue
        hashes = js.get("source_sha256") or {}
        if not hashes or any(source_sha256(n) != h for n, h in hashes.items()):
            return None
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import rocprof_summary
        step_bytes, steps = rocprof_summary.step_traffic(js["kernels"])
        return {"kernels": js["kernels"], "step_bytes": step_bytes, "steps": steps, "in_run": False, "source": os.path.basename(path)}
    return None


def kernel_traffic(counter, kernel, signature, walk=None):
    """(bytes per launch, source, measured in this run) of a kernel group: this run's own counter pass when there is one, else the
    newest committed summary of this workload and this kernel source (pmc_traffic)."""
    pref = pmc_prefix(kernel, walk)
    if counter and counter.get("in_run") and pref:
        for name, e in counter["kernels"].items():
            if name.startswith(pref):
                return int(e["fetch_bytes_corrected"] + e["write_bytes"]), counter["source"], True
    t, src = pmc_traffic(kernel, signature, walk)
    return t, src, False


def pmc_traffic(kernel, signature, walk=None):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary (profiles/*_hbm_pmc.json,
    written by tools/profile_round.sh from separate --pmc FETCH_SIZE / WRITE_SIZE passes of this same command).
    Counters cannot be read from inside the process: the figure is NOT measured in this run (the line says so). None unless
    that summary was taken on exactly this workload (its "workload" key equals `signature`): a figure measured on another
    configuration says nothing about this one."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_pmc.json")))
    pref = pmc_prefix(kernel, walk)
    if not files or not pref:
        return None, None
    for path in reversed(files):                 # the newest summary taken on this workload
        with open(path) as f:
            js = json.load(f)
        if js.get("workload") != signature:
            continue
        # ... and on this kernel: the summary carries the sha256 of every kernel source it was taken with; a summary without
        # hashes (rounds 1-4) or with another hash for the kernel's file is stale and says nothing about this build
        src = KERNEL_SOURCE.get(kernel)
        have = (js.get("source_sha256") or {}).get(src)
        if not have or have != source_sha256(src):
            return None, "stale: %s was taken with another %s" % (os.path.basename(path), src)
        for name, e in js["kernels"].items():
            if name.startswith(pref):
                return int(e["fetch_bytes_corrected"] + e["write_bytes"]), os.path.basename(path)
        return None, None
    return None, None


# ---- what the step's collectives cost on N ranks: a stated MODEL (no multi-GPU box was available to measure on) ----------------
# t(N, bytes) = floor + 2 (N - 1) hops x XGMI_HOP_US + 2 (N - 1) / N x bytes / (XGMI_LINK_GBS x XGMI_RING_EFFICIENCY)
#   floor                 the call's latency on a 1-rank communicator of this GPU (measured in this run: nvsm_comm_latency)
#   XGMI_HOP_US           ASSUMPTION: 1.5 us per ring hop for a latency-bound message (RCCL's LL protocol: a flagged store over one
#                         xGMI link + the receiver's poll); a ring all-reduce is 2 (N - 1) dependent hops
#   XGMI_LINK_GBS         153 GB/s per xGMI link and direction (the task's hardware notes: 7 links x ~153 GB/s per GPU); a ring uses ONE
#   XGMI_RING_EFFICIENCY  ASSUMPTION: half the link rate at these sizes (0.3 MB: far from the bandwidth regime)
XGMI_HOP_US, XGMI_LINK_GBS, XGMI_RING_EFFICIENCY = 1.5, 153.0, 0.5


def collective_model_us(n_ranks, payload_bytes, floor_us):
    hops = 2 * (n_ranks - 1)
    wire = hops / n_ranks * payload_bytes / (XGMI_LINK_GBS * XGMI_RING_EFFICIENCY * 1e3)      # bytes / (GB/s) = ns; -> us
    return floor_us + hops * XGMI_HOP_US + wire


def gemm_flops(wl, B):
    return 2.0 * B * wl["word_dim"] * wl["entity_dim"]


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
    flags = "unknown"
    try:
        with open(os.path.join(ROOT, "oracle", "Makefile")) as f:
            for line in f:
                if line.startswith("CXXFLAGS"):
                    flags = line.split("=", 1)[1].strip()
    except OSError:
        pass
    return {"value": B * steps / t_total, "unit": "windows/s", "cores": orc.lib().orc_num_threads(), "kind": "port",
            "compile_flags": flags,
            "sample": "%d full steps (batch %d) of the fp32 OpenMP oracle after %d warm-up, incl. host negative sampling; OpenMP team = "
                      "the CPUs the process may use (affinity mask capped by the cgroup quota), %d hardware threads visible; built with "
                      "g++ %s (x86-64-v3 = AVX2 + FMA instead of BASELINE.md's -march=native: the .so is built in the CPU container "
                      "and must run on the GPU box's host)" % (steps, B, warm, os.cpu_count() or 0, flags)}


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


class Leg:
    """One engine handle + its pool of synthetic batches: everything a timed region needs."""

    def __init__(self, env, wl, method, B, uniform_words=False, host_batches=False, seed=1234, exact_tables=False, sync_bn=1):
        import cunvsm_amd as ca
        self.env, self.wl, self.method, self.B = env, wl, method, B
        self.uniform_words = uniform_words
        cfg = ca.default_config(num_words=wl["num_words"], num_entities=wl["num_entities"], word_repr_size=wl["word_dim"],
                                entity_repr_size=wl["entity_dim"], window_size=wl["window"],
                                num_random_entities=wl["num_random"], batch_normalization=wl["batch_norm"],
                                nonlinearity=wl["nonlinearity"], clip_sigmoid=1,
                                bias_negative_samples=wl["bias_negative_samples"], regularization_lambda=1e-2, update_method=method,
                                max_batch_size=B, device=env.local_rank, sampler=ca.SAMPLER_DEVICE,
                                world_size=env.world, rank=env.rank, sync_batch_norm=sync_bn,
                                dp_exact_tables=int(bool(exact_tables) and env.world > 1))
        self.model = ca.Model(cfg)
        self.model.initialize(1)                # --seed 1 (scripts/functions.sh:393); identical replicas on every rank
        self.transport, self.comm_ranks = env.connect(self.model)
        self.pool = self.make_pool(seed, host_batches)

    def make_pool(self, seed, host, uniform=None):
        import cunvsm_amd as ca
        import torch
        wl, B, w = self.wl, self.B, self.wl["window"]
        uniform = self.uniform_words if uniform is None else uniform
        rs = np.random.RandomState(seed + self.env.rank)
        pool, self.touched_words = [], []
        for _ in range(4):
            words = (rs.randint(0, wl["num_words"], B * w).astype(np.int64) if uniform else zipf_ids(rs, wl["num_words"], B * w))
            labels = rs.randint(0, wl["num_entities"], B).astype(np.int64)
            self.touched_words.append(int(np.unique(words).size))
            if host:        # page-locked host buffers, as the trainer's (and the reference's) batches are
                pins = [ca.model.pinned_copy(x) for x in (words, labels, np.ones(B * w, np.float32), np.ones(B, np.float32))]
                self.env.keep.append(pins)
                pool.append(ca.Batch(pins[0].array, pins[1].array, pins[2].array, pins[3].array))
            else:
                dev = self.env.device
                pool.append(ca.Batch(torch.from_numpy(words).to(dev), torch.from_numpy(labels).to(dev),
                                     torch.ones(B * w, dtype=torch.float32, device=dev),
                                     torch.ones(B, dtype=torch.float32, device=dev)))
        return pool

    def run_steps(self, n, batches=None, read_every=0, deferred=False):
        args, model, lr = self.env.args, self.model, self.wl["lr"]
        batches = batches or self.pool
        if deferred:
            # EVERY step's loss is read back, one step late: the loss of step s is fetched after step s + 1 has been queued
            # (nvsm_step_deferred / nvsm_deferred_cost — what cuNVSMTrainModel does), so the GPU never waits for the host
            ticket = None
            for s in range(n):
                t = model.step_deferred(batches[s % len(batches)], lr)
                if ticket is not None:
                    model.deferred_cost(ticket)
                ticket = t
            if ticket is not None:
                model.deferred_cost(ticket)
            return
        for s in range(n):
            want = read_every > 0 and (s + 1) % read_every == 0
            if args.gate_us and s % max(1, args.gate_every) == 0:
                model.debug_delay(args.gate_us)
            if args.sequential:
                model.compute_cost(batches[s % len(batches)])
                model.compute_gradients()
                model.update(lr)
                if want:
                    model.get_cost()
            else:
                model.step(batches[s % len(batches)], lr, want_cost=want)

    def timed(self, n, batches=None, read_every=0, deferred=False):
        """EXACTLY n steps between barrier + synchronize on both sides; MAX over ranks. Seconds."""
        import torch
        env = self.env
        env.sync_all(self.model)
        t0 = time.perf_counter()
        self.run_steps(n, batches, read_every, deferred)
        self.enqueue_s.append(time.perf_counter() - t0)      # the host's share: all n steps queued (the GPU may still be running)
        self.model.synchronize()
        torch.cuda.synchronize()
        if env.dist is not None:
            env.dist.barrier()
        dt = time.perf_counter() - t0
        if env.dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if env.args.test_shared_gpu else "cuda")
            env.dist.all_reduce(t, op=env.dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def timed_repeats(self, n, repeats, batches=None, read_every=0, deferred=False):
        self.enqueue_s = []
        return [self.timed(n, batches, read_every, deferred) for _ in range(max(1, repeats))]

    def touched_rows(self):
        """Table rows one batch touches: words from the pool, documents from the ids the device sampler drew last."""
        ids = self.model.get_tensor("entity_ids")
        return {"words": int(round(float(np.mean(self.touched_words)))), "entities": int(np.unique(ids).size)}


class Env:
    """Process-wide state of a run: ranks, device, torch.distributed, the RCCL plumbing of each engine handle."""

    def __init__(self, args, world, rank, local_rank, host_bind):
        import torch
        self.args, self.world, self.rank, self.local_rank = args, world, rank, local_rank
        self.device = torch.device("cuda", local_rank)
        self.dist = None
        self.keep = []
        self.cpu_mask, self.numa_node = host_bind["cpu_mask"], host_bind["numa_node"]      # (main(): bound before the HIP runtime came up)
        self.cpus_bound = len(os.sched_getaffinity(0))
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo" if args.test_shared_gpu else "nccl", rank=rank, world_size=world)
            self.dist = dist

    def connect(self, model):
        """The engine's own RCCL communicator (all-reduces on its streams, no host round trip); if it cannot be built on
        this node, torch.distributed through the host-callback transport so that the run still completes."""
        if self.world <= 1:
            return "single", 0
        import torch
        from cunvsm_amd.model import comm_unique_id
        args, dist = self.args, self.dist
        ok = torch.zeros(1, device="cpu" if args.test_shared_gpu else "cuda")
        try:
            if args.test_shared_gpu:
                raise RuntimeError("shared-GPU test: RCCL cannot put two ranks on one device")
            obj = [comm_unique_id() if self.rank == 0 else None]
            dist.broadcast_object_list(obj, src=0)
            model.comm_init(obj[0])
            ok += 1
        except Exception as e:            # noqa: BLE001
            if not args.test_shared_gpu:
                sys.stderr.write("rank %d: nvsm_comm_init failed (%s)\n" % (self.rank, e))
        dist.all_reduce(ok)
        if int(ok.item()) == self.world:
            return "rccl", model.comm_size()      # ncclCommCount
        from cunvsm_amd import dp
        if args.test_shared_gpu:
            model.set_allreduce_callback(dp.torch_allreduce(dist))
            return "torch.distributed(gloo) via host callback", 0
        model.set_allreduce_callback(dp.torch_allreduce_device(dist, self.device))
        return "torch.distributed(nccl) via host callback", 0

    def sync_all(self, model):
        import torch
        model.synchronize()
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()


def update_roofline(leg, env, wl, method, B, uniform_words, steps, counter=None, timed_in_step=None):
    """Roofline entry of the documents update — by GPU time the largest kernel of a step (table_pass_kernel at the metric's
    shape, entry_walk_kernel where the documents table is much larger than the batch): algorithmic bytes; the kernel's time
    IN the step from an event pair riding on its own launch (a pass with no other records: it runs on side stream 1 next to the
    words chain of the main stream, so this includes what sharing the memory system costs it) and ALONE (compute_cost /
    compute_gradients / update as separate calls: nothing runs beside it); the committed counter bytes when the summary was
    taken on this workload and this update.hip."""
    kernel = "row_pass_entities"
    m, lr = leg.model, leg.wl["lr"]
    desc = m.describe()
    lazy = "documents lazy decay" in desc
    rows = {k: v for k, v in rows_visited(desc, leg.touched_rows(), method, B, lr, env.world).items() if k == "entities"}
    ab = algorithmic_bytes(kernel, wl, method, B, rows)
    out = {"kernel": kernel, "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes_per_launch": ab,
           "table_rows_visited": rows.get("entities", wl["num_entities"]),
           "decay": "lazy" if lazy else ("eager, identity in fp32 (1 - lambda lr / B rounds to 1): rows without entries are skipped" if rows else "eager"),
           "bytes": "gather of proj rows B*(k+1)*d_doc*4 + read and write of E and its first moments for every row the pass visits"}

    def timed(run):
        m.profile_enable(True)
        m.profile_select(kernel)
        m.profile_reset()
        run()
        env.sync_all(m)
        pr = m.profile()
        m.profile_enable(False)
        walk = "entry_walk" if pr.get("entry_walk_entities", (0, 0))[1] > 0 else "row_walk"
        return (pr[kernel][0] / pr[kernel][1] if pr.get(kernel, (0, 0))[1] > 0 else None), walk

    if timed_in_step:      # (ms, walk) from event pairs riding on the kernel's launch INSIDE the timed regions
        in_step, walk = timed_in_step
        out["in_step_timed_by"] = "events riding on the kernel's launch inside the timed regions"
    else:
        in_step, walk = timed(lambda: leg.run_steps(steps))
        out["in_step_timed_by"] = "events riding on the kernel's launch, in a pass of fused steps of its own behind the timed regions"

    def separate():
        for s_ in range(steps):
            b = leg.pool[s_ % len(leg.pool)]
            m.compute_cost(b)
            m.compute_gradients()
            m.update(lr)
    alone, _ = timed(separate)
    out["walk"] = walk
    for name, t in (("in_step", in_step), ("alone", alone)):
        if t:
            ach = ab / (t * 1e-3) / 1e9
            out[name] = {"avg_launch_ms": round(t, 4), "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4)}
    if in_step:
        out["achieved"], out["frac"], out["avg_launch_ms"] = out["in_step"]["achieved"], out["in_step"]["frac"], out["in_step"]["avg_launch_ms"]
    out["traffic"], out["traffic_source"], out["traffic_measured_in_run"] = kernel_traffic(counter, kernel, workload_signature(wl, method, uniform_words, B), walk)
    if out["traffic"]:
        out["traffic_over_algorithmic"] = round(out["traffic"] / ab, 3)
    return out


def secondary_leg(env, args, wl, method):
    """One configuration on an engine of its own: ms per step with no events in the timed regions, then the loss kernel's
    in-step time from a pass of its own with events riding on its launch."""
    B = wl["batch"]
    leg = Leg(env, wl, method, B, uniform_words=args.uniform_words)
    leg.run_steps(max(5, args.warmup))
    med, st = ms_stats(leg.timed_repeats(args.steps, min(args.repeats, 3)), args.steps, leg)
    ent = dict(value=round(B * 1e3 / med, 1), unit="windows/s", ms_per_step=round(med, 4), batch=B, update_method=method, steps_per_region=args.steps,
               workload="|V|=%d |D|=%d d_word=%d d_doc=%d" % (wl["num_words"], wl["num_entities"], wl["word_dim"], wl["entity_dim"]), **st)
    kernel = "loss_fused"
    leg.model.profile_enable(True)
    leg.model.profile_select(kernel)
    leg.model.profile_reset()
    leg.run_steps(min(args.steps, 50))
    env.sync_all(leg.model)
    pr = leg.model.profile()
    leg.model.profile_enable(False)
    if pr.get(kernel, (0, 0))[1] > 0:
        avg = pr[kernel][0] / pr[kernel][1]
        ab = algorithmic_bytes(kernel, wl, method, B)
        ent["roofline"] = {"kernel": kernel, "bound": "hbm", "achieved": round(ab / (avg * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": round(ab / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "avg_launch_ms": round(avg, 4),
                           "algorithmic_bytes_per_launch": ab, "traffic": None,
                           "note": "in-step time of the kernel (events riding on its launch), from a pass of its own behind the timed regions"}
    desc = leg.model.describe()
    touched = leg.touched_rows()
    rows = rows_visited(desc, touched, method, B, wl["lr"])
    # the documents update: by GPU time the largest kernel of every step this bench runs (the main leg's kernel_breakdown; entry walk
    # at configs[4]'s tables) — in the step and alone
    ru = update_roofline(leg, env, wl, method, B, args.uniform_words, min(args.steps, 20))
    leg.model.close()
    # counter bytes of this leg's step: a pass of its own in this run (--leg-pmc), else the committed summary of this workload
    sig = workload_signature(wl, method, args.uniform_words, B)
    counter = None
    if args.leg_pmc:
        flags = ["--config", args.config, "--batch", str(B), "--update-method", method] + (["--uniform-words"] if args.uniform_words else [])
        counter = pmc_in_run(flags)
    counter = counter or pmc_committed_step(sig)
    if ru:
        ru["traffic"], ru["traffic_source"], ru["traffic_measured_in_run"] = kernel_traffic(counter, ru["kernel"], sig, ru["walk"])
        if ru["traffic"]:
            ru["traffic_over_algorithmic"] = round(ru["traffic"] / ru["algorithmic_bytes_per_launch"], 3)
    if "roofline" in ent:      # the loss kernel's record
        ent["roofline"]["traffic"], ent["roofline"]["traffic_source"], ent["roofline"]["traffic_measured_in_run"] = kernel_traffic(counter, kernel, sig)
        ent["roofline_loss"] = ent.pop("roofline")
    if ru and "frac" in ru:
        ent["roofline"] = ru
    ent["roofline_step"] = step_roofline(wl, method, B, rows, med, counter)
    ent["rows_touched_per_batch"] = touched
    return ent


def run_secondary_leg(args, flags, pmc=False, env_extra=None):
    """A secondary leg in a fresh process, as a user would run that configuration: a second engine in a process whose first has
    lived (streams created and destroyed) is mapped onto the runtime's hardware queues differently and runs 2-10 % slower."""
    import subprocess
    # A timed region starts from an idle GPU and ends with a drained one: the fill and drain of the step's three streams (≈ 0.15 ms) are 1 % of
    # twenty 0.88 ms steps but 3 % of twenty 0.25 ms steps. The small-batch legs are steady-state figures (a rank's step in a long
    # run), so their regions are at least 200 steps long; the leg's line says how many.
    small = any(f in ("lse_small",) for f in flags) or any(flags[i] == "--batch" and int(flags[i + 1]) <= 12800 for i in range(len(flags) - 1))
    steps = max(args.steps, 200) if small else args.steps
    cmd = [sys.executable, os.path.abspath(__file__), "--secondary-leg", "--steps", str(steps), "--warmup", str(args.warmup),
           "--repeats", str(min(args.repeats, 3))] + flags + (["--leg-pmc"] if pmc and not args.no_pmc else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env_extra or {})))
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        raise SystemExit("secondary leg %s failed: %s" % (flags, r.stderr[-2000:]))
    return json.loads(lines[-1])


def cranfield_cli_leg(epochs=12):
    """cuNVSMTrainModel on tests/golden/cranfield (BASELINE configs[0]): None when the trainer binary or the fixture is absent."""
    import re
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "cunvsm_amd", "bin", "cuNVSMTrainModel")
    corpus = os.path.join(ROOT, "tests", "golden", "cranfield", "cranfield.trectext")
    if not (os.path.exists(exe) and os.path.exists(corpus)):
        return None
    out = tempfile.mkdtemp(prefix="nvsm_cranfield_")
    try:
        cmd = [exe, "--word_repr_size", "128", "--entity_repr_size", "256", "--window_size", "10", "--num_random_entities", "16",
               "--batch_size", "4096", "--nonlinearity", "tanh", "--bias_negative_samples", "--update_method", "full_adam",
               "--learning_rate", "0.001", "--num_epochs", str(epochs), "--seed", "1", "--sampler", "device", "--v", "1",
               "--output", os.path.join(out, "model"), corpus]
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": r.stderr[-300:]}
        bps = [float(x) for x in re.findall(r"\(([0-9.]+) batches/second\)", r.stderr)]
        wps = [float(x) for x in re.findall(r"([0-9.eE+]+) n-gram windows/second", r.stderr)]
        lists = re.findall(r"Epoch #[0-9]+: duration.*?cost=\[(.*?)\]", r.stderr)      # (the trainer logs the cost history so far)
        costs = [float(x) for x in lists[-1].split(",") if x.strip()] if lists else []
        ent = {"workload": "LSE on the Cranfield collection (1400 documents), batch 4096, tanh, bias_negative_samples, full_adam, device sampler, "
                           "through cuNVSMTrainModel: host batches over PCIe, loss of every step read one step late, async prefetch",
               "epochs": epochs, "wall_s": round(wall, 2)}
        if bps:
            # the reference's log line is CUMULATIVE (batches so far / seconds since the loop began, cpp/main.cu:604-611): epoch k alone
            # took k / bps[k] - (k - 1) / bps[k - 1] epochs-worth of seconds per batch of an epoch
            k = len(bps)
            last = 1.0 / (k / bps[-1] - (k - 1) / bps[-2]) if k >= 2 and bps[-1] > 0 and bps[-2] > 0 else bps[-1]
            ent.update(value=round(last, 1), unit="batches/s", note="batches per second of the LAST epoch alone (steady state, incl. its model dump); "
                       "cumulative_batches_per_s = the reference's own log figure at the end of the run (cpp/main.cu:604-611), which carries "
                       "the first epochs' one-off costs", cumulative_batches_per_s=bps[-1], batches_per_s_by_epoch_cumulative=bps)
            loop = re.findall(r"Training loop: ([0-9]+) batches in ([0-9.eE+-]+) seconds", r.stderr)
            if loop:
                ent["training_loop_s"] = round(float(loop[-1][1]), 3)
                ent["one_off_setup_s"] = round(wall - float(loop[-1][1]), 3)      # process start -> loop start: index build, engine, libhdf5
        if wps:
            ent["windows_per_s_last_epoch"] = wps[-1]
        if costs:
            ent["cost_first_last"] = [costs[0], costs[-1]]
        return ent
    except Exception as e:            # noqa: BLE001
        return {"error": str(e)[:300]}
    finally:
        shutil.rmtree(out, ignore_errors=True)


def ms_stats(times, steps, leg=None, enqueue=None):
    ms = sorted(t * 1e3 / steps for t in times)
    med = statistics.median(ms)
    st = {"repeats": len(ms), "ms_per_step_all": [round(x, 4) for x in ms],
          "spread": round((ms[-1] - ms[0]) / med, 4) if med > 0 else None}
    # how long the HOST took to queue a step (median region): well below ms_per_step = the GPU is the limit, equal = the host is
    enq = enqueue if enqueue is not None else getattr(leg, "enqueue_s", None)
    if enq and len(enq) == len(times):
        st["host_enqueue_ms_per_step"] = round(statistics.median(enq) * 1e3 / steps, 4)
    return med, st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; the median one is reported")
    ap.add_argument("--config", default="nvsm", choices=sorted(PRESETS), help="nvsm = BASELINE configs[1] (the bench line)")
    ap.add_argument("--update-method", default=None, choices=["sgd", "adagrad", "sparse_adam", "dense_adam", "full_adam"])
    ap.add_argument("--batch", type=int, default=None, help="windows per step (default 51200): per GPU at N = 1 and for the weak "
                    "figure, split over the ranks for the strong one")
    ap.add_argument("--num-words", type=int, default=None)
    ap.add_argument("--num-entities", type=int, default=None)
    ap.add_argument("--word-dim", type=int, default=None, help="experiments (row alignment): d_word other than the config's 300")
    ap.add_argument("--weak-scaling", action="store_true", help="N > 1: make the weak figure (51 200 windows per rank) the headline "
                    "`value` instead of the strong split of the metric's 51 200-window batch; both are always measured")
    ap.add_argument("--strong-scaling", action="store_true", help="(the default since round 3; kept for older command lines)")
    ap.add_argument("--exact-tables-leg", action="store_true", help="N > 1: also time the strong split with nvsm_config.dp_exact_tables "
                    "(every rank applies the whole batch's embedding updates: the single-GPU trajectory) -> \"exact_tables\" in the line")
    ap.add_argument("--test-shared-gpu", action="store_true", help="test of the N > 1 control flow on a 1-GPU box: every rank on "
                    "device 0, gloo rendezvous, all-reduces through the host-callback transport (not a measurement)")
    ap.add_argument("--launch-check", action="store_true", help="rendezvous only (gloo, no GPU): proves that the N-rank launch works")
    ap.add_argument("--uniform-words", action="store_true", help="uniform instead of Zipf(1) word ids (worst case for caches)")
    ap.add_argument("--host-batches", action="store_true", help="hand host buffers over each step in the MAIN timed region too")
    ap.add_argument("--secondary-leg", action="store_true", help="child mode of the secondary legs: this configuration on an engine of "
                    "its own in a process of its own — timed with no events, then the loss kernel's roofline from a pass with events; "
                    "one compact JSON line")
    ap.add_argument("--leg-pmc", action="store_true", help="with --secondary-leg: HBM counter bytes of the leg's workload from rocprofv3 --pmc "
                    "passes launched by the leg itself")
    ap.add_argument("--no-pmc", action="store_true", help="no rocprofv3 counter passes launched by this run (traffic then comes from the "
                    "committed summaries under profiles/, or is null)")
    ap.add_argument("--timed-kernels", default="loss_fused,row_pass_entities", help="kernel groups that carry an event pair on their own "
                    "launch inside the timed regions (the roofline kernels)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the read-back / host-batch / per-rank-shape / secondary legs")
    ap.add_argument("--cpu-steps", type=int, default=30, help="full-size steps of the CPU oracle timed for cpu_baseline (≈0.3 s each on the "
                    "16 CPUs the GPU box grants the process)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sequential", action="store_true", help="compute_cost / compute_gradients / update as separate calls on one stream "
                    "(un-overlapped per-kernel timings) instead of the fused multi-stream nvsm_step")
    ap.add_argument("--no-profile", action="store_true", help="no HIP events at all (no roofline / breakdown in the output)")
    ap.add_argument("--profile-all", action="store_true", help="events around every kernel group inside the timed region (≈5 %% slower)")
    ap.add_argument("--gate-us", type=int, default=0, help="profiling aid: a spin kernel of this many microseconds in front of every "
                    "step, so that the host has queued the step before the GPU starts it (the timeline then shows the GPU-side schedule)")
    ap.add_argument("--gate-every", type=int, default=1, help="with --gate-us: the spin kernel in front of every n-th step only, so that "
                    "the steps behind it run back to back as in steady state")
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

    import cunvsm_amd as ca  # noqa: F401  (fails loudly without the HIP library: no CPU fallback)

    # The submitting thread onto the CPUs of the GPU's NUMA node BEFORE anything touches the HIP runtime (nvsm_bind_host_thread finds
    # the device through sysfs then): the runtime's own threads and host allocations land on that node too. The small-batch legs are
    # ~45 launches and event calls per 0.15 ms, which the far socket cannot keep up with; the CPU baseline gets the original mask back.
    host_bind = {"cpu_mask": os.sched_getaffinity(0), "numa_node": None}
    try:
        host_bind["numa_node"] = ca.bind_host_thread(0 if args.test_shared_gpu else local_rank)
    except Exception:      # (no GPU: the check below says so)
        pass

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if args.test_shared_gpu:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit("--gpus %d but only %d GPU(s) are visible (--test-shared-gpu exercises the N-rank control flow on one)"
                         % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    env = Env(args, world, rank, local_rank, host_bind)
    dist = env.dist

    wl = workload(args)
    method = args.update_method
    if args.secondary_leg:
        if world != 1:
            raise SystemExit("--secondary-leg is a single-GPU child mode")
        print(json.dumps(secondary_leg(env, args, wl, method)), flush=True)
        return
    Bg = wl["batch"]                          # the metric's batch: per GPU at N = 1 / weak, global for the strong split
    w = wl["window"]
    quick = args.no_extra_legs or args.sequential or bool(args.gate_us)
    strong_ok = world > 1 and Bg % world == 0
    headline_strong = strong_ok and not args.weak_scaling
    # the leg behind `value` (roofline and kernel breakdown are taken on it) and, for N > 1, the other scaling figure
    B = Bg // world if headline_strong else Bg
    main_leg = Leg(env, wl, method, B, uniform_words=args.uniform_words, host_batches=args.host_batches,
                   seed=4321 if headline_strong else 1234)
    model = main_leg.model
    main_desc = model.describe()      # which kernel each product of a step takes, table modes, switches off their defaults

    if world > 1:
        # communicator set-up (connections, first-use kernels) is lazy: two untimed steps take it out of the way even when
        # the caller asks for no warm-up steps
        main_leg.run_steps(2)
    main_leg.run_steps(args.warmup)

    # Timed regions: HIP events on the roofline kernel only (the document gather + loss kernel; the pair rides on the kernel's
    # own launch). Events around every kernel group (~50 records per step) cost ≈5 % of the step, so the full per-kernel
    # breakdown comes from a second, untimed pass over the same batches (--profile-all puts it back into the timed regions).
    ROOFLINE_KERNEL, GATHER_KERNEL = "loss_fused", "gather_mean_words"
    model.profile_enable(not args.no_profile)
    # (only the roofline kernel carries events inside the timed regions: an event pair riding on a launch is not free — it costs
    #  the stream ~5 us per kernel, 1 % of this step with two kernels timed, 12 % of the batch-4096 LSE step: tools/ab_profile_cost.sh;
    #  the word gather's time for `roofline_gather` comes from the untimed pass below)
    TIMED = [k for k in args.timed_kernels.split(",") if k]
    model.profile_select(None if args.profile_all else ",".join(TIMED))
    model.profile_reset()
    times = main_leg.timed_repeats(args.steps, args.repeats, None, args.read_cost_every)
    main_enqueue = list(main_leg.enqueue_s)
    final_cost = model.get_cost()
    prof_timed = model.profile()
    prof = prof_timed
    breakdown_steps = args.steps * len(times)
    prof_riding = {}
    RIDING = ("gemm_bwd_T", "gemm_bwd_T_reduce", GATHER_KERNEL)
    if not args.no_profile and not args.profile_all:
        breakdown_steps = min(args.steps, 20)
        # kernels timed by events that ride on their own launch (no records around them), in a pass with no other records: the dT
        # product's execution time inside an otherwise undisturbed step. Its whole-CU workgroups and the documents pass become
        # runnable at the same instant; under rocprofv3's kernel trace the product is dispatched ~7 us earlier and takes
        # 0.10-0.11 ms, with records around every group 0.13-0.14, with records on this launch only 0.19 — the STEP takes the same
        # time in all three (its back half is bound by bytes).
        model.profile_select(",".join(RIDING))
        model.profile_reset()
        main_leg.run_steps(breakdown_steps)
        env.sync_all(model)
        prof_riding = {k: v for k, v in model.profile().items() if k in RIDING and v[1] > 0}
        if GATHER_KERNEL in prof_riding:
            prof_timed = dict(prof_timed)
            prof_timed[GATHER_KERNEL] = prof_riding.pop(GATHER_KERNEL)
        model.profile_select(None)
        model.profile_reset()
        main_leg.run_steps(breakdown_steps)          # every rank takes part (the collectives are in the step)
        env.sync_all(model)
        prof = model.profile()
    model.profile_enable(False)
    touched = main_leg.touched_rows() if rank == 0 and not args.no_profile else None
    roofline_update = None
    dt_alone = None
    if world == 1 and not args.no_profile and not args.sequential and not args.gate_us:
        UPD = "row_pass_entities"
        tin = None
        if prof_timed.get(UPD, (0, 0))[1] > 0:
            tin = (prof_timed[UPD][0] / prof_timed[UPD][1], "entry_walk" if prof.get("entry_walk_entities", (0, 0))[1] > 0 else "row_walk")
        roofline_update = update_roofline(main_leg, env, wl, method, B, args.uniform_words, min(args.steps, 20), timed_in_step=tin)
        # the dT product alone (separate calls on one stream): what a kernel trace reports for it, with nothing racing it for CUs
        model.profile_enable(True)
        model.profile_select("gemm_bwd_T")
        model.profile_reset()
        for s_ in range(min(args.steps, 20)):
            model.compute_cost(main_leg.pool[s_ % len(main_leg.pool)])
            model.compute_gradients()
            model.update(wl["lr"])
        env.sync_all(model)
        pr = model.profile()
        model.profile_enable(False)
        if pr.get("gemm_bwd_T", (0, 0))[1] > 0:
            dt_alone = pr["gemm_bwd_T"][0] / pr["gemm_bwd_T"][1]

    def leg_value(leg, batches=None, read_every=0, repeats=None, deferred=False):
        """windows/s and ms per step of a secondary leg: median of `repeats` regions of --steps steps, all ranks' batches"""
        ts = leg.timed_repeats(args.steps, repeats or min(args.repeats, 3), batches, read_every, deferred)
        med, st = ms_stats(ts, args.steps, leg)
        return med, st

    # ---- secondary legs: no events ----------------------------------------------------------------------------------
    extra = {}
    counter = None
    sig_main = workload_signature(wl, method, args.uniform_words, B)
    base_flags = (["--uniform-words"] if args.uniform_words else []) + \
                 [x for k, v in (("--num-words", args.num_words), ("--num-entities", args.num_entities), ("--word-dim", args.word_dim)) if v is not None for x in (k, str(v))]
    main_transport, main_comm_ranks = main_leg.transport, main_leg.comm_ranks
    if not quick:
        if world == 1:
            # (a) loss read back after EVERY step, as iterate_data does (cpp/main.cu:427-444; SURVEY §8d "with the loss read
            #     back every step")
            main_leg.run_steps(2, None, 1)
            med, _ = leg_value(main_leg, None, 1)
            extra["value_readback_every_step"] = round(B * 1e3 / med, 1)
            # (a') every step's loss read back ONE STEP LATE (the trainer's protocol: nothing stalls)
            med, _ = leg_value(main_leg, None, 0, deferred=True)
            extra["value_readback_one_step_late"] = round(B * 1e3 / med, 1)
            # (b) page-locked HOST batches handed over each step: PCIe-inclusive (never `value`)
            if not args.host_batches:
                hpool = main_leg.make_pool(1234, True)
                main_leg.run_steps(3, hpool)
                med, _ = leg_value(main_leg, hpool)
                extra["value_host_batches"] = round(B * 1e3 / med, 1)
                del hpool
            # (c) uniform word ids on the same engine (the worst case for the caches)
            secondary = {}
            if not args.uniform_words:
                upool = main_leg.make_pool(777, args.host_batches, uniform=True)
                main_leg.run_steps(3, upool)
                med, st = leg_value(main_leg, upool)
                secondary["uniform_words"] = dict(value=round(B * 1e3 / med, 1), unit="windows/s", ms_per_step=round(med, 4), **st)
            # The legs below run on engines of their own, each in a process of its own (run_secondary_leg), as a user would run that
            # configuration. The main engine is closed first so that the GPU is theirs.
            main_leg.pool = None
            main_leg.model.close()
            torch.cuda.empty_cache()
            # HBM counter bytes of the headline workload, measured in this run (two rocprofv3 passes of this script's bare form)
            if not args.no_pmc and not args.no_profile:
                pmc_flags = ["--config", args.config, "--batch", str(B), "--update-method", method] + base_flags
                counter = pmc_in_run(pmc_flags)
            base = (["--uniform-words"] if args.uniform_words else [])
            # (d) the reference recipe's optimiser (scripts/functions.sh:395: --update_method full_adam)
            if args.config == "nvsm" and method != "full_adam":
                ent = run_secondary_leg(args, base + ["--config", "nvsm", "--batch", str(B), "--update-method", "full_adam"])
                secondary["full_adam"] = {k: ent[k] for k in ("value", "unit", "ms_per_step", "repeats", "ms_per_step_all", "spread")}
            # (d') the headline step with the projection products on the EXACT-fp32 MFMA kernels (NVSM_GEMM_SPLIT=0: gemm_tstat / tiled /
            #      gemm_panel, v_mfma_f32_*_f32) instead of the three-bf16-plane arithmetic: the apples-to-apples figure against the
            #      reference's sgemm (cpp/params.cu:417,528, cpp/objective.cu:453) — what the bf16-plane products buy is the difference
            if args.config == "nvsm" and gemm_split_products() and "NVSM_GEMM_SPLIT" not in os.environ:
                ent = run_secondary_leg(args, base + ["--config", "nvsm", "--batch", str(B), "--update-method", method], env_extra={"NVSM_GEMM_SPLIT": "0"})
                secondary["exact_fp32_gemm"] = dict({k: ent[k] for k in ("value", "unit", "ms_per_step", "repeats", "ms_per_step_all", "spread")},
                                                    arithmetic="NVSM_GEMM_SPLIT=0: every projection product on exact-fp32 MFMAs (no bf16 planes)")
                extra["value_exact_fp32_gemm"] = ent["value"]
            # (e') the other BASELINE configs that fit one GPU: configs[4]'s tables at the metric's batch (|V| = 500 k, |D| = 2 M: E is
            #      2 GB, i.e. the document gather comes out of HBM proper, not the Infinity Cache — its loss-kernel roofline is
            #      reported here) and configs[3] (LSE, batch 4 096, Adagrad)
            if args.config == "nvsm" and Bg == 51200 and method == "sparse_adam":
                secondary["large_tables"] = run_secondary_leg(args, ["--config", "large_tables"], pmc=True)
                # The batch-4096 step runs in one of two modes per PROCESS (≈ 0.160 and ≈ 0.18 ms: one process in five takes the slow
                # one with any build — which of two co-critical launch chains wins a race that is decided once, when the streams are
                # made; profiles/NOTES_r05.md): three processes, the median one reported, all three in the line.
                runs = [run_secondary_leg(args, ["--config", "lse_small"], pmc=(i == 0)) for i in range(3)]
                pmc_run = runs[0]
                runs.sort(key=lambda e: e["ms_per_step"])
                secondary["lse_small"] = dict(runs[1], processes=3, ms_per_step_by_process=[e["ms_per_step"] for e in runs])
                if runs[1] is not pmc_run:      # (the counter bytes were taken by the first process: the median process's records carry them)
                    for key in ("roofline", "roofline_loss"):
                        if key in pmc_run and key in secondary["lse_small"]:
                            for f in ("traffic", "traffic_source", "traffic_measured_in_run", "traffic_over_algorithmic"):
                                if f in pmc_run[key]:
                                    secondary["lse_small"][key][f] = pmc_run[key][f]
                    if "counter_bytes" in pmc_run.get("roofline_step", {}):
                        cb = dict(pmc_run["roofline_step"]["counter_bytes"])
                        a2 = cb["traffic"] / (secondary["lse_small"]["ms_per_step"] * 1e-3) / 1e9
                        cb.update(achieved=round(a2, 1), frac=round(a2 / HBM_PEAK_GBS, 4))
                        secondary["lse_small"]["roofline_step"]["counter_bytes"] = cb
            # (f) BASELINE configs[0]: the LSE recipe on the Cranfield collection through the cuNVSMTrainModel CLI (host layer + HIP
            #     path end to end: index built from the TREC text, batches over PCIe, every step's loss read one step late, async
            #     prefetch) — batches per second of the last epoch, as the reference's own log line reports it (cpp/main.cu:604-611)
            if args.config == "nvsm" and Bg == 51200:
                cf = cranfield_cli_leg()
                if cf:
                    secondary["lse_cranfield_cli"] = cf
            if secondary:
                extra["secondary"] = secondary
            # (e) the per-rank share of the N-GPU metric (51 200 / N windows per rank) on this one GPU, each
            #     on an engine created for that batch size as rank r of an N-GPU job would create it. Compute only.
            if args.config == "nvsm" and Bg == 51200:
                shapes = {}
                for n in (8, 4, 2):
                    ent = run_secondary_leg(args, base + ["--config", "nvsm", "--batch", str(Bg // n), "--update-method", method])
                    shapes[str(Bg // n)] = dict(ms_per_step=ent["ms_per_step"], ranks=n, steps_per_region=ent["steps_per_region"],
                                                **{k: ent[k] for k in ("repeats", "ms_per_step_all", "spread")})
                extra["per_rank_shapes"] = shapes
        else:
            # the other scaling figure of the same run: weak (51 200 windows per rank) beside a strong headline, or the
            # strong split (BASELINE configs[2]) beside a weak one
            other_B = Bg if headline_strong else (Bg // world if strong_ok else None)
            if other_B:
                leg = Leg(env, wl, method, other_B, uniform_words=args.uniform_words, host_batches=args.host_batches,
                          seed=1234 if headline_strong else 4321)
                leg.run_steps(2 + max(3, args.warmup))
                med, st = leg_value(leg)
                total = other_B * world if headline_strong else Bg
                fig = dict(value=round(total * 1e3 / med, 1), unit="windows/s", ms_per_step=round(med, 4),
                           scaling="weak" if headline_strong else "strong", global_batch=total, batch_per_rank=other_B,
                           steps=args.steps, **st)
                extra["weak" if headline_strong else "strong"] = fig
                leg.model.close()
                del leg
            if strong_ok and wl["batch_norm"]:
                # the strong split with PER-SHARD batch-norm statistics (nvsm_config.sync_batch_norm = 0: every rank normalises
                # with the statistics of its own 51 200 / N windows — not the single-GPU arithmetic, one collective less per step)
                leg = Leg(env, wl, method, Bg // world, uniform_words=args.uniform_words, host_batches=args.host_batches, seed=4321, sync_bn=0)
                leg.run_steps(2 + max(3, args.warmup))
                med, st = leg_value(leg)
                extra["per_shard_batch_norm"] = dict(value=round(Bg * 1e3 / med, 1), unit="windows/s", ms_per_step=round(med, 4), scaling="strong",
                                                     global_batch=Bg, batch_per_rank=Bg // world, steps=args.steps, collectives_per_step=1,
                                                     note="sync_batch_norm=0: per-shard batch statistics — ONE collective per step ([dT | db | loss] in one f32 "
                                                          "all-reduce) instead of three", **st)
                leg.model.close()
                del leg
            if args.exact_tables_leg and strong_ok:
                # the strong split again with exact data-parallel tables: what the single-GPU trajectory costs
                leg = Leg(env, wl, method, Bg // world, uniform_words=args.uniform_words, host_batches=args.host_batches, seed=4321,
                          exact_tables=True)
                leg.run_steps(2 + max(3, args.warmup))
                med, st = leg_value(leg)
                extra["exact_tables"] = dict(value=round(Bg * 1e3 / med, 1), unit="windows/s", ms_per_step=round(med, 4), scaling="strong",
                                             global_batch=Bg, batch_per_rank=Bg // world, steps=args.steps,
                                             note="nvsm_config.dp_exact_tables: every rank applies the table updates of all ranks' windows", **st)
                leg.model.close()
                del leg

    if counter is None and world == 1:
        counter = pmc_committed_step(sig_main)
    if rank == 0:
        ms_per_step, tstats = ms_stats(times, args.steps, enqueue=main_enqueue)
        global_batch = Bg if (headline_strong or world == 1) else Bg * world
        value = global_batch * 1e3 / ms_per_step
        scaling = "single" if world == 1 else ("strong" if headline_strong else "weak")
        this_fig = dict(value=round(value, 1), unit="windows/s", ms_per_step=round(ms_per_step, 4), scaling=scaling,
                        global_batch=global_batch, batch_per_rank=B, steps=args.steps, **tstats)
        # a lazily decayed table's row passes only read and write the rows the batch touches (the engine notes a
        # `lazy_stamp_<table>` launch per update of such a table)
        rows = rows_visited(main_desc, touched, method, B, wl["lr"], world)
        # dominant kernel group and its roofline
        arithmetic = product_arithmetic(main_desc)
        breakdown = {}
        for k, (ms, n) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
            if n == 0:
                continue
            avg = ms / n
            ent = {"avg_ms": round(avg, 4), "launches_per_step": round(n / breakdown_steps, 2)}
            if ms == 0.0:                       # a path note of the engine (which walk a table pass took), not a timing
                breakdown[k] = {"launches_per_step": ent["launches_per_step"], "note": True}
                continue
            ab = algorithmic_bytes(k, wl, method, B, rows)
            if ab:
                ent["algorithmic_GBps"] = round(ab / (avg * 1e-3) / 1e9, 1)
                if ent["algorithmic_GBps"] > HBM_PEAK_GBS:
                    # more algorithmic bytes per second than HBM can deliver: the gathered rows repeat within the batch and
                    # are served by L2 / the Infinity Cache (the PMC summaries under profiles/ show the HBM bytes)
                    ent["served_from_cache"] = True
            if k in prof_riding:
                r_ms, r_n = prof_riding[k]
                ent["avg_ms_no_other_records"] = round(r_ms / r_n, 4)
                ent["timed_by"] = ("events riding on the kernel's launch; avg_ms_no_other_records = the same in a pass with records on "
                                   "this launch only (the product's whole-CU workgroups and the documents pass race for CUs at the same "
                                   "instant: who wins depends on what else is recorded — DESIGN 5.5 item 7)")
            if k == "gemm_bwd_T" and dt_alone:
                # the figure to hold against profiles/: rocprofv3's kernel trace dispatches the product a few microseconds before the
                # documents pass, so it gets its CUs at once and runs as long as it does alone; the in-step event figures above are
                # outcomes of that race and are kept for the record only
                ent["in_step_event_ms"] = ent["avg_ms"]
                ent["avg_ms"] = round(dt_alone, 4)
                ent["avg_ms_is"] = "the kernel alone (separate calls on one stream, events riding on its launch): agrees with the rocprofv3 kernel trace under profiles/"
            if k.startswith("gemm_") and not k.endswith("_reduce"):
                # useful (fp32) flops per second. Large batches run the split-bf16 kernels (gemm_split.hip / gemm_dt.hip): every
                # fp32 operand cut exactly into three bf16 pieces, 6 (or 9) bf16 MFMAs per product, fp32 accumulation — the
                # matrix pipe then issues `products` times the useful flops at the bf16 rate
                ent["TFLOPs"] = round(gemm_flops(wl, B) / (avg * 1e-3) / 1e12, 1)
                ent["over_f32_mfma_peak"] = round(ent["TFLOPs"] / F32_MFMA_PEAK_TFLOPS, 3)      # 157.3 TF/s: what exact-fp32 MFMAs could do at best
                nprod = arithmetic.get(k, 0)      # (which kernel a product of this batch size takes: nvsm_describe)
                if nprod:
                    ent["arithmetic"] = "f32 as 3 bf16 planes, %d of 9 partial products, f32 accumulation" % nprod
                    ent["bf16_mfma_frac"] = round(ent["TFLOPs"] * nprod / BF16_MFMA_PEAK_TFLOPS, 3)
                else:
                    ent["mfma_frac"] = ent["over_f32_mfma_peak"]
            breakdown[k] = ent
        # Roofline kernel: the document-embedding gather + loss kernel — the HBM gather the north star names, and the
        # largest kernel of the step that runs with nothing else next to it but the (tiny) side-stream sorts. In the fused
        # step the documents update / dT GEMM overlap the dx GEMM / words update on a second stream; their event-timed
        # durations (marked "overlapped") include the time they share the chip and are not per-kernel roofline figures.
        # (gemm_bwd_T and its slab reduce are timed by events riding on their launches: the kernels' own execution times)
        for k in (() if args.sequential else ("update_entities", "row_pass_entities", "transform_update", "csr_entities", "csr_words")):
            if k in breakdown:
                breakdown[k]["overlapped"] = True
        have = lambda k: prof_timed.get(k, (0, 0))[1] > 0
        roofline = roofline_gather = None
        sig = workload_signature(wl, method, args.uniform_words, B)
        if have(ROOFLINE_KERNEL):
            dom = ROOFLINE_KERNEL
            ab = algorithmic_bytes(dom, wl, method, B)
            avg = round(prof_timed[dom][0] / prof_timed[dom][1], 4)      # HIP events inside the timed regions
            ach = ab / (avg * 1e-3) / 1e9
            traffic, traffic_src, traffic_live = kernel_traffic(counter, dom, sig)
            roofline = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                        # counters cannot be read in-process: this run's own rocprofv3 passes of the workload (true), a committed
                        # summary of this workload and this kernel source (false), or null
                        "traffic_measured_in_run": traffic_live,
                        "algorithmic_bytes_per_launch": ab, "avg_launch_ms": avg,
                        "timed_by": "events riding on the kernel's launch inside the timed regions",
                        "bytes": "document gather B*(k+1)*d_doc*4 + pre/proj/dy 3*B*d_doc*4"}
            if traffic:
                roofline["traffic_over_algorithmic"] = round(traffic / ab, 3)
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
                # the same two kernels on COUNTER bytes (what HBM actually delivered: the word rows repeat within a batch and
                # mostly come out of L2): committed PMC summary of this workload over this run's kernel times
                tw, _, tw_live = kernel_traffic(counter, GATHER_KERNEL, sig)
                if traffic is not None and tw is not None:
                    ach3 = (traffic + tw) / (t2 * 1e-3) / 1e9
                    roofline_gather["counter_bytes"] = {"traffic": traffic + tw, "achieved": round(ach3, 1),
                                                        "frac": round(ach3 / HBM_PEAK_GBS, 4), "traffic_source": traffic_src,
                                                        "traffic_measured_in_run": bool(traffic_live and tw_live)}
        # `roofline` is the kernel with the largest share of the step's GPU time (kernel_breakdown: one launch per step each): at
        # every shape of this bench the documents update (update_roofline: in the step — next to the words chain — and alone);
        # the document gather + loss kernel, the north star's named gather, stays as `roofline_loss`
        single = {k: e["avg_ms"] * e["launches_per_step"] for k, e in breakdown.items()
                  if not e.get("note") and (("algorithmic_GBps" in e) or ("TFLOPs" in e))}
        gpu_time = sum(single.values())
        roofline_loss = roofline
        dominant = max(single, key=single.get) if single else None
        if roofline_update:
            roofline_update["traffic"], roofline_update["traffic_source"], roofline_update["traffic_measured_in_run"] = \
                kernel_traffic(counter, roofline_update["kernel"], sig, roofline_update.get("walk"))
            if roofline_update["traffic"]:
                roofline_update["traffic_over_algorithmic"] = round(roofline_update["traffic"] / roofline_update["algorithmic_bytes_per_launch"], 3)
        if dominant == "row_pass_entities" and roofline_update and "frac" in roofline_update:
            roofline = dict(roofline_update)
        elif dominant and dominant != ROOFLINE_KERNEL and "algorithmic_GBps" in breakdown.get(dominant, {}):
            e = breakdown[dominant]
            ab = algorithmic_bytes(dominant, wl, method, B, rows)
            roofline = {"kernel": dominant, "bound": "hbm", "achieved": e["algorithmic_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(e["algorithmic_GBps"] / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": ab, "avg_launch_ms": e["avg_ms"],
                        "timed_by": "record pair around the group in the untimed breakdown pass"}
            roofline["traffic"], roofline["traffic_source"], roofline["traffic_measured_in_run"] = kernel_traffic(counter, dominant, sig)
        if roofline and dominant:
            roofline["share_of_gpu_time"] = round(single.get(roofline["kernel"], 0.0) / gpu_time, 4) if gpu_time else None
            roofline["dominant_by"] = "largest avg_ms x launches among kernel_breakdown's single-kernel groups: " + dominant
        roofline_step = step_roofline(wl, method, B, rows, ms_per_step, counter) if world == 1 else None
        out = {
            "metric": "n-gram windows/sec (batch=51200, NVSM config)" if args.config == "nvsm" and Bg == 51200
                      else "n-gram windows/sec (--config %s, batch=%d)" % (args.config, Bg), "value": round(value, 1), "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
            "dtype_detail": ("f32 storage and accumulation; the three projection products multiply f32 operands as 3 exact bf16 planes, "
                             "%d of 9 partial products, on the bf16 matrix pipe (DESIGN 4.1)" % gemm_split_products()) if all(arithmetic.values())
                            else ("f32 throughout (exact-f32 MFMA kernels)" if not any(arithmetic.values()) else
                                  "f32 storage and accumulation; products on bf16 planes: %s" % arithmetic),
            "engine": main_desc,
            "data": "synthetic",
            "config": {"workload": "%s synthetic |V|=%d |D|=%d d_word=%d d_doc=%d window=%d neg=%d global batch=%d (%d/GPU) "
                                   "%s%s %s lambda=1e-2 lr=%g %s word ids, inputs %s, device negative sampler"
                                   % ("NVSM" if wl["batch_norm"] else "LSE", wl["num_words"], wl["num_entities"], wl["word_dim"],
                                      wl["entity_dim"], w, wl["num_random"], global_batch, B,
                                      wl["nonlinearity"], "+BN" if wl["batch_norm"] else "", method, wl["lr"],
                                      "uniform" if args.uniform_words else "Zipf(1)",
                                      "handed over as page-locked host buffers" if args.host_batches else "resident in HBM"),
                       "global_batch": global_batch, "batch_per_rank": B, "parallelism": "dp%d" % world, "update_method": method,
                       "collectives": main_transport, "comm_ranks": main_comm_ranks,
                       # per step and rank under data parallelism: [Σx | Σx²] (f64, forward, batch-norm only), [loss | Σdy | Σdy·x̂]
                       # (f64, backward), dT (f32, 307 KB) — DESIGN.md §6
                       # (round 6: without synchronised batch-norm statistics [db | loss] ride behind dT: one collective)
                       "collectives_per_step": 0 if world == 1 else (3 if wl["batch_norm"] else 1),
                       "inputs": "host" if args.host_batches else "hbm", "step": "sequential calls" if args.sequential else "fused nvsm_step",
                       "workload_signature": sig,
                       "host_thread": {"gpu_numa_node": env.numa_node, "cpus_bound": env.cpus_bound, "cpus_given": len(env.cpu_mask),
                                       "note": "nvsm_bind_host_thread: the submitting thread on the CPUs of the GPU's NUMA node; "
                                               "timing.host_enqueue_ms_per_step = the host's time to queue one step"}},
            "timing": tstats,
            "roofline": roofline,
            "roofline_loss": roofline_loss,
            "roofline_update": roofline_update,
            "roofline_gather": roofline_gather,
            "roofline_step": roofline_step,
            "kernel_breakdown": breakdown,
            "kernel_breakdown_source": ("timed regions" if args.profile_all else
                                        "separate untimed pass of %d steps with events around every kernel group" % breakdown_steps),
            "rows_touched_per_batch": touched,
            "final_cost": round(float(final_cost), 6),
        }
        if world > 1:
            out[scaling] = this_fig
        out.update(extra)
        if "per_rank_shapes" in extra and "6400" in extra["per_rank_shapes"]:
            # what 8 ranks would deliver on the metric's 51 200-window batch if the per-rank step were all there is (three
            # latency-bound all-reduces per step come on top: DESIGN.md §6): 8 x 6 400 windows per per-rank step
            ms8 = extra["per_rank_shapes"]["6400"]["ms_per_step"]
            out["strong_projection_8gpu"] = {"value": round(51200 * 1e3 / ms8, 1), "unit": "windows/s",
                                             "speedup_over_1gpu": round(ms_per_step / ms8, 2),
                                             "basis": "per_rank_shapes[6400] on one GPU, collectives excluded"}
        if world == 1 and not quick:
            # What a rank of the N-GPU job adds to the per-rank step: the step's collectives, their payloads, and the
            # latency of each on a 1-rank RCCL communicator of this GPU — the floor a call costs before any wire time (over xGMI a
            # ring all-reduce of these sizes is latency-bound: 2 (N - 1) hops). Bounds the 8-GPU step from per_rank_shapes.
            try:
                import ctypes
                us = (ctypes.c_float * 3)()
                nb = (ctypes.c_int64 * 3)()
                ca._lib.check(ca.lib().nvsm_comm_latency(local_rank, wl["entity_dim"], wl["word_dim"], 200, us, nb))
                names = ["allreduce f64 [sum x | sum x^2] (forward, batch-norm)", "allreduce f64 [loss | sum dy | sum dy*xhat] (backward)",
                         "allreduce f32 dT (projection gradient)"]
                keep = [0, 1, 2] if wl["batch_norm"] else [2]
                folded_bytes = int(nb[2]) + 4 * (wl["entity_dim"] + 2)      # [dT | db | loss hi | loss lo], f32
                N8 = 8
                calls = [{"what": names[i], "payload_bytes": int(nb[i]), "rccl_1rank_latency_us": round(float(us[i]), 2),
                          "model_8rank_us": round(collective_model_us(N8, int(nb[i]), float(us[i])), 1)} for i in keep]
                one = {"what": "allreduce f32 [dT | db | loss hi | loss lo] (the only collective of a step without synchronised statistics)",
                       "payload_bytes": folded_bytes, "rccl_1rank_latency_us": round(float(us[2]), 2),
                       "model_8rank_us": round(collective_model_us(N8, folded_bytes, float(us[2])), 1)}
                if not wl["batch_norm"]:
                    calls = [one]
                out["config"]["collectives_dp"] = {
                    "per_step": len(calls), "per_step_per_shard_batch_norm": 1, "per_step_without_batch_norm": 1,
                    "calls": calls, "call_per_shard_batch_norm": one,
                    "rccl_1rank_latency_us_per_step": round(float(sum(us[i] for i in keep)), 2),
                    "model": {"formula": "t(N, bytes) = rccl_1rank_latency + 2 (N - 1) x hop_us + 2 (N - 1) / N x bytes / (link_GBps x ring_efficiency)",
                              "hop_us": XGMI_HOP_US, "link_GBps": XGMI_LINK_GBS, "ring_efficiency": XGMI_RING_EFFICIENCY,
                              "assumptions": "hop_us and ring_efficiency are ASSUMED (no multi-GPU box to measure on; RCCL's latency-bound LL ring: "
                                             "2 (N - 1) dependent hops of a flagged store over one xGMI link); the floor is measured in this run",
                              "model_8rank_us_per_step": round(sum(c["model_8rank_us"] for c in calls), 1)},
                    "note": "each call out of place on a 1-rank communicator of this GPU, 200 calls back to back on one stream (RCCL's launch + "
                            "copy kernel: the floor a collective starts from). With synchronised batch-norm statistics a step has three "
                            "collectives, each behind the kernel that needs the previous one's sums; with per-shard statistics "
                            "(sync_batch_norm=0) or without batch-norm, one"}
                if "per_rank_shapes" in extra and "6400" in extra["per_rank_shapes"]:
                    per_rank = extra["per_rank_shapes"]["6400"]["ms_per_step"]
                    sp = out["strong_projection_8gpu"]
                    sp["compute_only"] = {"ms_per_step": per_rank, "speedup_over_1gpu": sp["speedup_over_1gpu"]}
                    # every collective of a step sits on a chain the next kernel waits for (the two f64 ones on the main stream; the dT
                    # one on the chain the next forward product joins — at per-rank batches that chain is co-critical, NOTES_r06 §1):
                    # the model adds all of them to the per-rank step
                    for key, cl in (("with_collectives_model", calls), ("per_shard_batch_norm_model", [one])):
                        ms8 = per_rank + sum(c["model_8rank_us"] for c in cl) * 1e-3
                        sp[key] = {"ms_per_step": round(ms8, 4), "value": round(51200 * 1e3 / ms8, 1), "speedup_over_1gpu": round(ms_per_step / ms8, 2),
                                   "collectives_per_step": len(cl),
                                   "basis": "per_rank_shapes[6400] + the step's collectives by config.collectives_dp.model (8 ranks), none of them hidden"}
                    sp["speedup_over_1gpu_compute_only"] = sp["speedup_over_1gpu"]
                    sp["speedup_over_1gpu"] = sp["with_collectives_model"]["speedup_over_1gpu"] if wl["batch_norm"] else sp["per_shard_batch_norm_model"]["speedup_over_1gpu"]
                    sp["basis"] = "per_rank_shapes[6400] on one GPU + the modelled collectives (speedup_over_1gpu_compute_only: without them)"
                    # weak scaling (51 200 windows per rank, the design's natural use): this run's own step + the same collectives
                    wk = {}
                    for key, cl in (("synchronised_batch_norm", calls), ("per_shard_batch_norm", [one])):
                        msw = ms_per_step + sum(c["model_8rank_us"] for c in cl) * 1e-3
                        wk[key] = {"ms_per_step": round(msw, 4), "value": round(8 * B * 1e3 / msw, 1), "speedup_over_1gpu": round(8 * ms_per_step / msw, 2)}
                    out["weak_projection_8gpu"] = dict(wk, basis="this run's step (51 200 windows per rank) + the step's collectives by config.collectives_dp.model, "
                                                                 "none of them hidden; tables rank-local (DESIGN.md 6)")
            except Exception as e:            # noqa: BLE001  (no librccl: the line says so instead of failing the bench)
                out["config"]["collectives_dp"] = {"error": str(e)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            os.sched_setaffinity(0, env.cpu_mask)      # every CPU the process was given, not only the GPU's node
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
