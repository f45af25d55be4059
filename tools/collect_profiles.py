#!/usr/bin/env python3
"""Copies the round's measurement artefacts from gpurun_out/ (scratch) into profiles/ (tracked):
bench lines, rocprofv3 kernel-stats summaries, PMC HBM traffic and SQ counter summaries.
usage: python tools/collect_profiles.py [round tag, default r01]   (after tools/final_profiles.sh,
tools/pmc_traffic.sh <prec> gpurun_out/pmc_traffic_<prec> and tools/pmc_sq.sh fp16x3 gpurun_out/pmc_sq)"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def short(name):
    m = re.search(r"(k_[a-z0-9_]+?)(?:_f16_v2|_f16|_f32|IL|<|\(|E)", name)
    base = m.group(1) if m else name
    return "k_srp_gemm" if base.startswith("k_srp_gemm") else base


for prec in ("fp16x3", "fp16", "fp32"):
    log = os.path.join(G, "final", "bench_%s.log" % prec)
    if os.path.exists(log):
        lines = [l for l in open(log) if l.startswith("{")]
        if lines:
            open(os.path.join(P, "%s_bench_%s.json" % (tag, prec)), "w").write(lines[-1])
    ks = os.path.join(G, "final", "kernel_stats_%s.csv" % prec)
    if os.path.exists(ks):
        shutil.copy(ks, os.path.join(P, "%s_kernel_stats_%s.csv" % (tag, prec)))
    tj = os.path.join(G, "pmc_traffic_%s" % prec, "traffic_%s.json" % prec)
    if os.path.exists(tj):
        raw = json.load(open(tj))
        out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) of `python3 bench.py "
                         "--steps 3 --warmup 1 --cpu-frames 0 --precision %s` (tools/pmc_traffic.sh), MI355X" % prec,
               "correction": "gfx950: FETCH_SIZE counts exactly half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM "
                             "section): hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
               "workload": "8 arrays x 4096 frames per launch", "kernels": {}}
        for k, v in raw.items():
            out["kernels"][short(k)] = {"kernel": k,
                                        "FETCH_SIZE_KB_per_launch": v["FETCH_SIZE"]["sum"] / v["FETCH_SIZE"]["dispatches"],
                                        "WRITE_SIZE_KB_per_launch": v["WRITE_SIZE"]["sum"] / v["WRITE_SIZE"]["dispatches"]}
        json.dump(out, open(os.path.join(P, "%s_pmc_traffic_%s.json" % (tag, prec)), "w"), indent=1)
ks = os.path.join(G, "final", "kernel_stats_mvdr.csv")
if os.path.exists(ks):
    shutil.copy(ks, os.path.join(P, "%s_kernel_stats_mvdr.csv" % tag))
log = os.path.join(G, "final", "bench_mvdr.log")
if os.path.exists(log):
    lines = [l for l in open(log) if l.startswith("{")]
    if lines:
        open(os.path.join(P, "%s_bench_mvdr.json" % tag), "w").write(lines[-1])
sq = os.path.join(G, "pmc_sq", "sq_fp16x3.json")
if os.path.exists(sq):
    raw = json.load(open(sq))
    out = {"source": "rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes, tools/pmc_sq.sh) of `python3 bench.py --steps 2 "
                     "--warmup 1 --cpu-frames 0 --precision fp16x3`, MI355X, values per dispatch (8 arrays x 4096 frames)",
           "notes": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_BUSY_CYCLES is summed over "
                    "the 32 shader engines; SQ_INSTS_* are wave-instructions.", "kernels": raw}
    json.dump(out, open(os.path.join(P, "%s_pmc_sq_fp16x3.json" % tag), "w"), indent=1)
print(sorted(os.listdir(P)))
