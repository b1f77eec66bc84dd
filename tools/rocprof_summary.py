#!/usr/bin/env python3
"""Condenses rocprofv3 rocpd (.db) outputs into the small text summaries committed under profiles/.

  tools/rocprof_summary.py stats  <stats.db>                 -> per-kernel calls / total / average / %
  tools/rocprof_summary.py pmc    <fetch.db> <write.db> [out.json [workload signature]] -> per-kernel FETCH_SIZE / WRITE_SIZE per launch
  tools/rocprof_summary.py mfma   <mfma.db> [out.json]      -> per-kernel counter-based MFMA utilisation
  tools/rocprof_summary.py timeline <stats.db> [marker]     -> start / duration / queue of every kernel of one step
FETCH_SIZE on gfx950 under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md §HBM); the pmc summary
prints both the raw counter and the doubled ("corrected") read bytes.
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::trampoline_kernel<rocprim::ROCPRIM_\d+_NS::detail::wrapped_(\w+?)_config.*",
                  r"rocprim::\1", name)
    name = name.replace("void ", "").replace("cunvsm::", "")
    name = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", name)
    return name[:90]


def stats(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name").fetchall()
    agg = {}
    for name, n, tot, avg, mn, mx in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
        a[0] += n; a[1] += tot; a[2] = min(a[2], mn); a[3] = max(a[3], mx)
    total = sum(a[1] for a in agg.values())
    print("%-92s %8s %12s %10s %10s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-92s %8d %12.1f %10.2f %10.2f %10.2f %7.2f" % (k, a[0], a[1] / 1e3, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100 * a[1] / total))
    print("%-92s %8d %12.1f" % ("TOTAL", sum(a[0] for a in agg.values()), total / 1e3))


def mfma(path, json_path=None, cu_num=256, xcds=8):
    """Counter-based MFMA utilisation per kernel: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32
    GRBM_GUI_ACTIVE. MfmaUtil = 100 * SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * CU_NUM * 4) is derived_counters.xml's
    (gfx94x) formula; on gfx950 rocprofv3 reports GRBM_GUI_ACTIVE SUMMED over the 8 XCDs (1 998 687 "cycles" for a 103.75 us
    kernel = 8 x 2.4 GHz x 103.75 us), so the per-device busy window is GRBM_GUI_ACTIVE / 8:
    MfmaUtil = 100 * BUSY / ((GUI_ACTIVE / 8) * 256 CUs * 4 SIMDs). Cross-check: SQ_VALU_MFMA_BUSY_CYCLES is exactly
    (number of v_mfma_f32_32x32x2 instructions) x 64 cycles, and flops = SQ_INSTS_VALU_MFMA_MOPS_F32 * 512."""
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    ncol = next((c for c in ("counter_name", "name") if c in cols), None)
    if ncol is None:
        print("columns:", cols); return
    agg = {}
    for kname, cname, n, val, dur in db.execute(
            "select kernel_name, %s, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, %s" % (ncol, ncol)):
        e = agg.setdefault(short(kname), {"launches": n, "avg_us": dur / 1e3})
        e[cname] = e.get(cname, 0.0) + val
    rows = []
    for k, e in agg.items():
        busy, gui = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), e.get("GRBM_GUI_ACTIVE", 0.0)
        # (the split-bf16 GEMMs issue bf16 MFMAs: six or nine per fp32 product; one MOP = 512 flops for either type)
        mops_f32, mops_bf16 = e.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0), e.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
        mops = mops_f32 + mops_bf16
        if mops <= 0:
            continue
        e["mfma_type"] = "bf16" if mops_bf16 > mops_f32 else "f32"
        e["mfma_util_pct"] = 100.0 * busy / ((gui / xcds) * cu_num * 4) if gui else None
        e["mfma_util_pct_uncorrected_formula"] = 100.0 * busy / (gui * cu_num * 4) if gui else None
        e["mfma_flops"] = mops * 512
        e["TFLOPs_under_pmc"] = mops * 512 / (e["avg_us"] * 1e-6) / 1e12 if e["avg_us"] else None
        rows.append((k, e))
    print("%-70s %7s %10s %5s %14s %14s %12s" % ("kernel", "calls", "avg_us", "type", "MFMA GFLOP", "MfmaUtil %", "TF/s (pmc run)"))
    for k, e in sorted(rows, key=lambda kv: -kv[1]["mfma_flops"]):
        print("%-70s %7d %10.2f %5s %14.3f %14.1f %12.1f" % (k[:70], e["launches"], e["avg_us"], e["mfma_type"], e["mfma_flops"] / 1e9,
                                                             e["mfma_util_pct"] or 0.0, e["TFLOPs_under_pmc"] or 0.0))
    if json_path:
        import json
        with open(json_path, "w") as f:
            json.dump({"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE; "
                               "MfmaUtil = 100*BUSY/((GUI_ACTIVE/8 XCDs)*256 CUs*4 SIMDs) — GRBM_GUI_ACTIVE is reported summed over the 8 XCDs; "
                               "the fp32 MFMA peak is 157.3 TF/s; kernels of type bf16 are the split-bf16 GEMMs (issued bf16 MFMA flops: 6 or 9 per fp32 flop, bf16 peak 2.5 PF/s)",
                       "kernels": {k: e for k, e in rows}}, f, indent=1, sort_keys=True)


def pmc_merge(fetch_path, write_path):
    """kernel (short name) -> [launches, avg FETCH_SIZE (KB), avg WRITE_SIZE (KB), avg duration (ns), launches of the write pass]"""
    out = {}
    for path, col in ((fetch_path, 0), (write_path, 1)):
        db = sqlite3.connect(path)
        for name, n, val, dur in db.execute("select kernel_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name"):
            k = short(name)
            e = out.setdefault(k, [0, 0.0, 0.0, 0.0, 0])
            # weighted merge of kernels that share a short name
            if col == 0:
                e[1] = (e[1] * e[0] + val * n) / (e[0] + n); e[3] = (e[3] * e[0] + dur * n) / (e[0] + n); e[0] += n
            else:
                e[2] = (e[2] * e[4] + val * n) / (e[4] + n); e[4] += n
    return out


def pmc_table(fetch_path, write_path):
    """Per-launch HBM-side bytes per kernel, FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads: MI355X_MICROARCH.md
    §HBM), WRITE_SIZE as reported — what bench.py's in-run counter passes and the committed summaries both hold."""
    return {k: {"launches": e[0], "fetch_bytes_corrected": 2 * e[1] * 1024, "write_bytes": e[2] * 1024, "avg_us": e[3] / 1e3}
            for k, e in pmc_merge(fetch_path, write_path).items()}


def step_traffic(js):
    """HBM bytes of ONE step from a per-kernel table: every kernel's bytes x its launches over the number of steps of the run
    (= launches of the once-per-step prologue kernel). None when the run had no prologue launches (host-sampler runs)."""
    steps = sum(e["launches"] for k, e in js.items() if k.startswith("step_prologue_kernel"))
    if not steps:
        return None, 0
    total = sum((e["fetch_bytes_corrected"] + e["write_bytes"]) * e["launches"] for e in js.values())
    return total / steps, steps


def pmc(fetch_path, write_path, json_path=None, workload=None):
    out = pmc_merge(fetch_path, write_path)
    if json_path:
        import json
        js = pmc_table(fetch_path, write_path)
        import glob, hashlib, os
        csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cunvsm_amd", "csrc")
        hashes = {os.path.basename(p): hashlib.sha256(open(p, "rb").read()).hexdigest() for p in sorted(glob.glob(os.path.join(csrc, "*.hip")))}
        with open(json_path, "w") as f:
            per_step, steps = step_traffic(js)
            json.dump({"workload": workload, "steps": steps, "step_bytes": per_step,
                       # bench.py's pmc_traffic() only cites a summary whose kernel source is byte-for-byte this build's
                       "source_sha256": hashes,
                       "note": "per-launch HBM-side bytes from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); "
                               "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads), "
                               "WRITE_SIZE as reported", "kernels": js}, f, indent=1, sort_keys=True)
    print("%-70s %7s %14s %16s %14s %10s %12s" % ("kernel", "calls", "FETCH_KB/launch", "FETCHx2_MB(corr)", "WRITE_KB/launch", "avg_us", "HBM_GB/s(corr)"))
    for k, e in sorted(out.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        rd = 2 * e[1] * 1024
        wr = e[2] * 1024
        gbs = (rd + wr) / (e[3] * 1e-9) / 1e9 if e[3] else 0
        print("%-70s %7d %14.1f %16.2f %14.1f %10.2f %12.1f" % (k[:70], e[0], e[1], rd / 1e6, e[2], e[3] / 1e3, gbs))


def timeline(path, marker="sample_entities_kernel", which=-2):
    """Start offset / duration / queue of every kernel of one steady-state step (from one `marker` launch to the next)."""
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = next((c for c in ("stream_id", "queue_id", "queue", "stream") if c in cols), None)
    rows = db.execute("select name, start, end, %s from kernels order by start" % (qcol or "0")).fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < 3:
        print("columns:", cols); return
    a, b = marks[which], marks[which + 1]
    t0 = rows[a][1]
    print("step of %.1f us (%s -> next %s); columns: start_us dur_us end_us queue kernel" % ((rows[b][1] - t0) / 1e3, marker, marker))
    for name, st, en, q in rows[a:b]:
        print("%9.1f %8.1f %9.1f  q%-4s %s" % ((st - t0) / 1e3, (en - st) / 1e3, (en - t0) / 1e3, q, short(name)[:70]))


if __name__ == "__main__":
    if sys.argv[1] == "timeline":
        timeline(sys.argv[2], *(sys.argv[3:4]))
    elif sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "mfma":
        mfma(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None, sys.argv[5] if len(sys.argv) > 5 else None)
