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
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def short(name):
    m = re.search(r"(k_[a-z0-9_]+?)(?:_f16_v2|_f16|_f32|IL|<|\(|E)", name)
    base = m.group(1) if m else name
    if base.startswith("k_srp_gemm_repair"):
        return "k_srp_gemm_repair"
    if base.startswith("k_mvdr_analyse"):
        return "k_mvdr_analyse"
    return "k_srp_gemm" if base.startswith("k_srp_gemm") else base


for prec in ("adaptive", "fp16x3", "fp16", "fp32"):
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
               "correction": "gfx950: FETCH_SIZE counts exactly half of a WIDE coalesced read stream (16 B per lane: MI355X_MICROARCH.md, HBM "
                             "section) -- the contraction (LDS-DMA, 16 B per lane), the scan and repair kernels: hbm_bytes = (2*FETCH_SIZE + "
                             "WRITE_SIZE) * 1024.  The wave-per-run kernels of round 3 (k_stft_phat_wave, k_beamform_wave) load 4 B per "
                             "lane; calibrated on their known input -- 0.537 GB of PCM per step, every sample fetched from HBM once, the "
                             "overlapping half frame served by L2 -- FETCH_SIZE reads 0.52-0.55 GB there, i.e. the bytes themselves: "
                             "factor 1 for those two (fetch_factor below)",
               "workload": "8 arrays x 4096 frames per step; values are per STEP (a kernel that runs twice per step -- k_stft_phat_wave "
                           "in the adaptive mode: all frames, then the listed repair groups -- has both launches added up)", "kernels": {}}
        # launches per counter pass = steps of that bench run: every per-step kernel but the list-mode analysis runs once per step
        # (k_bf_table runs once per context and is left out of the count)
        steps = {c: max(1, min(v[c]["dispatches"] for k, v in raw.items() if c in v and "k_bf_table" not in k)) for c in ("FETCH_SIZE", "WRITE_SIZE")}
        tot = 0.0
        for k, v in raw.items():
            name = short(k)
            factor = 1.0 if name in ("k_stft_phat_wave", "k_beamform_wave", "k_bf_table") else 2.0
            e = out["kernels"].setdefault(name, {"kernel": name, "instantiations": [], "dispatches_per_step": 0.0, "FETCH_SIZE_KB_per_launch": 0.0,
                                                 "WRITE_SIZE_KB_per_launch": 0.0, "fetch_factor": factor, "hbm_bytes_per_step": 0.0})
            e["instantiations"].append(k)
            e["dispatches_per_step"] += v["FETCH_SIZE"]["dispatches"] / steps["FETCH_SIZE"]
            f_kb, w_kb = v["FETCH_SIZE"]["sum"] / steps["FETCH_SIZE"], v["WRITE_SIZE"]["sum"] / steps["WRITE_SIZE"]
            e["FETCH_SIZE_KB_per_launch"] += f_kb          # (per step, all launches of the step added up)
            e["WRITE_SIZE_KB_per_launch"] += w_kb
            b = (factor * f_kb + w_kb) * 1024.0
            e["hbm_bytes_per_step"] += b
            tot += b
        out["total_hbm_bytes_per_step"] = tot
        out["algorithmic_bytes_per_step"] = 18440 * 32768
        json.dump(out, open(os.path.join(P, "%s_pmc_traffic_%s.json" % (tag, prec)), "w"), indent=1)
ks = os.path.join(G, "final", "kernel_stats_mvdr.csv")
if os.path.exists(ks):
    shutil.copy(ks, os.path.join(P, "%s_kernel_stats_mvdr.csv" % tag))
log = os.path.join(G, "final", "bench_mvdr.log")
if os.path.exists(log):
    lines = [l for l in open(log) if l.startswith("{")]
    if lines:
        open(os.path.join(P, "%s_bench_mvdr.json" % tag), "w").write(lines[-1])
for name in ("adaptive_check.json", "precision_report.json", "host_path.log", "bench_128x256.json", "bench_single_stream.json", "shapes.log", "gputest.log", "fallback.log",
             "bench_driver_cmd.json", "bench_driver_cmd_detail.json", "bench_warmup0.json", "bench_warmup1.json", "kernel_stats_driver_cmd.csv", "stream_latency.log",
             "repair_breakdown.log", "timeline.log"):
    src = os.path.join(G, "final", name)
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copy(src, os.path.join(P, "%s_%s" % (tag, name)))
# MVDR (BASELINE configs[3]): tools/pmc_traffic.sh adaptive gpurun_out/pmc_traffic_mvdr "--config mvdr", tools/pmc_sq.sh adaptive gpurun_out/pmc_sq_mvdr "--config mvdr"
tj = os.path.join(G, "pmc_traffic_mvdr", "traffic_adaptive.json")
if os.path.exists(tj):
    raw = json.load(open(tj))
    steps = {c: max(1, min(v[c]["dispatches"] for k, v in raw.items() if c in v)) for c in ("FETCH_SIZE", "WRITE_SIZE")}
    S_, F_, Mm, Kb, hop = 256, 64, 16, 513, 512
    known = {"k_mvdr_analyse": {"read": S_ * Mm * (F_ + 1) * hop * 4, "write": S_ * F_ * Kb * Mm * 8},
             "k_mvdr_solve": {"read": S_ * F_ * Kb * Mm * 8 + S_ * Kb * Mm * Mm * 8, "write": S_ * F_ * Kb * 8 + S_ * Kb * Mm * Mm * 8},
             "k_mvdr_synth": {"read": S_ * F_ * Kb * 8, "write": S_ * F_ * hop * 4}}
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) of `python3 bench.py --config mvdr --steps 3 --warmup 1 "
                     "--cpu-frames 0` (tools/pmc_traffic.sh adaptive <dir> \"--config mvdr\"), MI355X; 256 streams x 64 frames x 16 microphones per step",
           "correction": "gfx950: FETCH_SIZE counts half of a WIDE coalesced read stream (MI355X_MICROARCH.md, HBM section).  The factor of each kernel is chosen "
                         "against the bytes its launch is KNOWN to read (expected_read_bytes: PCM once / the [bin][mic] spectra + the covariance state / the beamformed "
                         "bins): 1 if FETCH_SIZE itself is nearer to them, 2 if twice FETCH_SIZE is",
           "kernels": {}}
    tot = 0.0
    for k, v in raw.items():
        name = short(k)
        f_kb, w_kb = v["FETCH_SIZE"]["sum"] / steps["FETCH_SIZE"], v["WRITE_SIZE"]["sum"] / steps["WRITE_SIZE"]
        exp = known.get(name)
        factor = 2.0
        if exp and abs(f_kb * 1024.0 - exp["read"]) < abs(2.0 * f_kb * 1024.0 - exp["read"]):
            factor = 1.0
        e = out["kernels"].setdefault(name, {"kernel": name, "instantiations": [], "FETCH_SIZE_KB_per_step": 0.0, "WRITE_SIZE_KB_per_step": 0.0, "fetch_factor": factor,
                                             "hbm_bytes_per_step": 0.0, "expected_read_bytes": exp["read"] if exp else None, "expected_write_bytes": exp["write"] if exp else None})
        e["instantiations"].append(k)
        e["FETCH_SIZE_KB_per_step"] += f_kb
        e["WRITE_SIZE_KB_per_step"] += w_kb
        b = (factor * f_kb + w_kb) * 1024.0
        e["hbm_bytes_per_step"] += b
        tot += b
    out["total_hbm_bytes_per_step"] = tot
    out["algorithmic_bytes_per_step"] = (Mm * hop * 4 + hop * 4) * S_ * F_
    json.dump(out, open(os.path.join(P, "%s_pmc_traffic_mvdr.json" % tag), "w"), indent=1)
sqm = os.path.join(G, "pmc_sq_mvdr", "sq_adaptive.json")
if os.path.exists(sqm):
    json.dump({"source": "rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes, tools/pmc_sq.sh adaptive <dir> \"--config mvdr\") of `python3 bench.py --config mvdr "
                         "--steps 2 --warmup 1 --cpu-frames 0`, MI355X, values per dispatch (256 streams x 64 frames)",
               "notes": "as the adaptive file; clock_ghz = SQ_BUSY_CYCLES / 32 / avg_duration_ns_under_pmc", "kernels": json.load(open(sqm))},
              open(os.path.join(P, "%s_pmc_sq_mvdr.json" % tag), "w"), indent=1)
sq = os.path.join(G, "pmc_sq", "sq_adaptive.json")
if os.path.exists(sq):
    raw = json.load(open(sq))
    out = {"source": "rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes, tools/pmc_sq.sh) of `python3 bench.py --steps 2 "
                     "--warmup 1 --cpu-frames 0 --precision adaptive`, MI355X, values per dispatch (8 arrays x 4096 frames)",
           "notes": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_BUSY_CYCLES is summed over "
                    "the 32 shader engines; SQ_INSTS_* are wave-instructions; avg_duration_ns_under_pmc is the kernel's average duration in counter pass 1 and "
                    "clock_ghz = SQ_BUSY_CYCLES / 32 / that duration (the shader clock bench.py prices its issue-rate roofline with).", "kernels": raw}
    json.dump(out, open(os.path.join(P, "%s_pmc_sq_adaptive.json" % tag), "w"), indent=1)
print(sorted(os.listdir(P)))
