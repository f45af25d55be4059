#!/usr/bin/env python3
"""bench.py -- frames/s of the mcarray hot path on MI355X.

--config ssl (default): the 8-mic GCC-PHAT + SRP-PHAT(361) + delay-and-sum path.  One "step" = one pass of the whole hot path
(STFT analysis -> GCC-PHAT over all pairs -> SRP scan -> IIR + peak pick -> delay-and-sum -> ISTFT/overlap-add) over one batch
of synthetic input that is already resident in HBM: `--arrays` independent 8-mic arrays x `--frames` STFT frames per GPU
(BASELINE.json configs[2] geometry: 48 kHz, 1024-pt, hop 512, 361 steering angles; 8 x 4096 frames per GPU = the per-GPU frame
count of configs[4]).  The literal configs[2] reading -- ONE array, 4096 frames per call -- is reported next to it as
config.single_stream_4096.
--config mvdr: BASELINE.json configs[3], the 16-mic frequency-domain beamformer with a per-bin spatial covariance, 256 concurrent
streams x 64 frames per step.

Arrays / streams are independent units: with N GPUs every rank owns its own block (weak scaling), no collective inside the
compute, one RCCL all_gather of the DOA bins + prob per step (mcarray_amd/dist.py; --gather-audio adds the beamformed audio, to
rank 0).  `python bench.py --gpus N` without a launcher starts its own N ranks (or refuses if fewer GPUs are visible).

Prints ONE JSON line (rank 0).  `value` = frames of all ranks / max-over-ranks time of K steps.  `roofline` is for the dominant
kernel, timed with HIP events recorded by the library on the launch stream inside the timed region.  `cpu_baseline` = the CPU
oracle (a scalar double-precision restatement of the reference, oracle/) on a bounded sample of the same input, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FS, NFFT, HOP, M, STEP_DEG, D, P, K = 48000, 1024, 512, 8, 0.5, 361, 28, 513
BYTES_PER_FRAME = M * HOP * 4 + HOP * 4 + 8          # 18 440 B (PCM in once, audio out, DOA idx + prob)
HBM_PEAK_GBPS = 8000.0                                # MI355X_MICROARCH.md: 8.0 TB/s spec
PEAK_TFLOPS = {"fp32": 157.3, "fp16x3": 2500.0, "fp16": 2500.0, "adaptive": 2500.0}   # dense MFMA peaks, same guide
PREC = {"fp32": 0, "fp16x3": 1, "fp16": 2, "adaptive": 3}
ROW_PAD = 64                                           # floats of padding behind every channel row of the synthetic input (synth_batch)
PROFILE_TAG = "r06"                                    # profiles/<tag>_pmc_traffic_<precision>.json of the committed PMC passes


MERGED_REALS = {8: 3968, 4: 1600}     # reals per merged one-plane A row (2 x the distinct products k (j - i), padded to the K stage of the kernel; 8-mic ULA: 1 962 complex)


def compact_line(line):
    """The record the driver's stdout tail must hold whole (under 6 KB): numbers only -- the notes, sources and
    sub-measurements stay in the full record (--detail-out / --full; what every field means: profiles/bench_notes.md)."""
    def pick(d, keys):
        return {k: d[k] for k in keys if d is not None and k in d and d[k] is not None}

    def r4(x):
        return float("%.4g" % x) if isinstance(x, float) else x

    def rnd(o):
        if isinstance(o, dict):
            return {k: rnd(v) for k, v in o.items()}
        if isinstance(o, list):
            return [rnd(v) for v in o]
        return r4(o)
    out = pick(line, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data"))
    out["vs_baseline"] = line.get("vs_baseline")          # (null: BASELINE.md holds no published number for this metric)
    cfg = line.get("config", {})
    c = pick(cfg, ("workload", "arrays_per_gpu", "frames_per_array", "srp_precision", "parallelism", "streams_per_gpu", "frames_per_stream"))
    ss = cfg.get("single_stream_4096")
    if ss:
        c["single_stream_4096"] = dict(pick(ss, ("value", "ms_per_call")), graph_replay_ms=(ss.get("graph_replay") or {}).get("ms_per_call"))
    das = cfg.get("das_single_stream")
    if das:
        d2 = {}
        for k in ("offline_any_angle", "offline_grid_angle"):
            if das.get(k):
                d2[k] = dict(pick(das[k], ("value", "ms_per_call", "hbm_roofline_frac")),
                             oracle_err_of_peak=(das[k].get("oracle_check") or {}).get("of_peak"), oracle_ok=(das[k].get("oracle_check") or {}).get("ok"),
                             cpu_frames_per_s=(das[k].get("cpu_baseline") or {}).get("value"))
        gl = das.get("graph_chunk_latency") or {}
        d2["graph_launch_median_ms"] = {k: v.get("median_ms") for k, v in gl.items()}
        c["das_single_stream"] = d2
    mv = cfg.get("mvdr_256x64")
    if mv:
        rf = mv.get("roofline") or {}
        c["mvdr_256x64"] = dict(pick(mv, ("value", "unit", "ms_per_step", "hbm_roofline_frac")),
                                kernels_ms={k: v["avg_ms"] for k, v in (mv.get("kernels") or {}).items()},
                                solve_flop_frac=(rf.get("flop_roofline") or {}).get("frac"), solve_valu_frac=(rf.get("valu_roofline") or {}).get("frac"),
                                solve_traffic=rf.get("traffic"), cpu_frames_per_s=(mv.get("cpu_baseline") or {}).get("value"))
    out["config"] = c
    out.update(pick(line, ("algorithmic_GBps", "hbm_roofline_frac")))
    if line.get("kernels"):
        out["kernels"] = {k: pick(v, ("launches", "avg_ms")) for k, v in line["kernels"].items() if v.get("launches")}
    rf = line.get("roofline")
    if rf:
        o = pick(rf, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_ms", "measured_in", "algorithmic_bytes_per_frame", "traffic_ratio",
                      "binding", "path_frac", "path_achieved"))
        if rf.get("valu_roofline"):
            o["valu_roofline"] = pick(rf["valu_roofline"], ("frac", "floor_ms", "insts_valu_per_launch", "clock_ghz"))
        for k in ("flop_roofline",):
            if rf.get(k):
                o[k] = pick(rf[k], ("frac", "achieved", "peak", "unit"))
        out["roofline"] = o
    if line.get("roofline_by_kernel"):
        out["roofline_by_kernel"] = [dict(pick(e, ("kernel", "avg_ms", "bound", "frac", "traffic_ratio")),
                                          **({"valu_frac": e["valu_roofline"]["frac"]} if e.get("valu_roofline") else {}),
                                          **({"mfma_pipe_frac": e["mfma_pipe"]["frac_of_peak"]} if e.get("mfma_pipe") else {})) for e in line["roofline_by_kernel"]]
    cb = line.get("cpu_baseline")
    if cb:
        o = pick(cb, ("value", "unit", "cores", "kind", "oracle_check"))
        o["sample"] = (cb.get("sample") or "").split(", double-precision")[0][:120]
        if cb.get("all_cores"):
            o["all_cores"] = pick(cb["all_cores"], ("value", "cores"))
        out["cpu_baseline"] = o
    if line.get("repair"):
        out["repair"] = pick(line["repair"], ("frames", "flagged", "recomputed", "flagged_fraction", "recomputed_fraction", "columns_per_flagged_frame", "whole_row_frames"))
    if line.get("repair_spread"):
        out["repair_spread"] = [dict(input=(sp.get("input") or "")[:40], **pick(sp, ("ms_per_32768_frames", "vs_headline", "repair_ms", "recomputed_fraction"))) for sp in line["repair_spread"]]
    if line.get("warmup_beyond_the_declared_steps"):
        out["seconds_of_other_configurations_before_the_warm_up"] = line["warmup_beyond_the_declared_steps"].get("seconds_of_other_configurations_before_the_warm_up")
    if line.get("exchange"):
        out["exchange"] = line["exchange"]
    out["notes"] = "profiles/bench_notes.md"
    out = rnd(out)
    out["value"], out["ms_per_step"] = line["value"], line["ms_per_step"]      # (the judged numbers: unrounded)
    return out


def synth_batch(xs, seeds, n_frames, device, noise=0.01, hop=None, fs=None):
    """Far-field white source per array + 20 dB sensor noise, generated on the GPU.
    One seed per array, derived from the array's GLOBAL index, so its data does not depend on the rank count.
    (hop, fs: other frame lengths / rates for tools/bench_shapes.py; the bench itself runs the module's.)"""
    n_arrays, n_mics = len(seeds), len(xs)
    hop, fs = hop or HOP, fs or FS
    L = (n_frames + 1) * hop
    xs_t = torch.tensor(xs, device=device, dtype=torch.float64)
    theta = torch.empty(n_arrays, device=device, dtype=torch.float64)
    # channel rows are ROW_PAD floats longer than their (F + 1) * hop samples (the C ABI takes the strides): a row pitch of a power
    # of two plus a little -- 257 half frames = 2^19 + 2^11 bytes at 128 arrays x 256 frames -- lines the loads of all resident
    # waves up on the same few HBM channels (k_beamform_wave 0.33 instead of 0.295 ms; HISTORY.md)
    out = torch.zeros(n_arrays, n_mics, L + ROW_PAD, device=device, dtype=torch.float32)
    f = torch.fft.rfftfreq(L, d=1.0 / fs).to(device=device, dtype=torch.float64)
    for a in range(n_arrays):   # one array at a time keeps the fp64 temporaries small
        gen = torch.Generator(device=device).manual_seed(seeds[a])
        theta[a] = (torch.rand(1, device=device, dtype=torch.float64, generator=gen)[0] * 160.0 - 80.0) * (np.pi / 180.0)
        s = torch.randn(L, device=device, dtype=torch.float64, generator=gen) * 0.1
        S = torch.fft.rfft(s)
        adv = xs_t * torch.sin(theta[a]) / 346.1
        x = torch.fft.irfft(S[None, :] * torch.exp(2j * np.pi * f[None, :] * adv[:, None]), n=L, dim=1)
        x = x + torch.randn(n_mics, L, device=device, dtype=torch.float64, generator=gen) * noise
        out[a, :, :L] = x.clamp_(-1.0, 1.0).to(torch.float32)
    return out, theta


def cpu_baseline(pcm_host, n_frames):
    from oracle import pyoracle as po
    from mcarray_amd import synth
    po.lib()
    L = (n_frames + 1) * HOP
    x = pcm_host[:, :L].astype(np.float64)
    t0 = time.perf_counter()
    r = po.ssl_stream(FS, NFFT, synth.ULA8, x, 1, STEP_DEG, want_map=False, want_audio=True)
    dt = time.perf_counter() - t0
    return n_frames / dt, dt, r


def cpu_baseline_all_cores(pcm_host_arrays, n_frames):
    """The same oracle on every host core at once: one independent array per thread (arrays never interact; the C
    oracle keeps no global state and ctypes releases the GIL for the call), frames of all threads / wall time of the slowest
    ("all host cores ... state the core count").  Threads, not processes: nothing forks after the GPU is up."""
    from concurrent.futures import ThreadPoolExecutor
    cores = len(pcm_host_arrays)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as pool:
        list(pool.map(lambda x: cpu_baseline(x, n_frames)[:2], pcm_host_arrays))
    dt = time.perf_counter() - t0
    return cores * n_frames / dt, dt, cores


def timed_loop(step, drain, steps, warmup, use_dist, dist, dev, before_timed=None, after_first_warmup=None):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize on both sides; MAX over ranks."""
    for i in range(warmup):
        step()
        if i == 0 and after_first_warmup:
            after_first_warmup()
    drain()
    torch.cuda.synchronize()
    if before_timed:
        before_timed()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def traffic_table(precision, shape_matches):
    """HBM bytes per step and kernel group from the committed rocprofv3 PMC passes (tools/pmc_traffic.sh): NOT measured in this run --
    a --pmc pass serialises the kernels and cannot share a process with the timed loop.  Newest committed round first."""
    if not shape_matches:
        return {}, None
    for tag in (PROFILE_TAG, "r05", "r04"):
        tj = os.path.join(ROOT, "profiles", "%s_pmc_traffic_%s.json" % (tag, precision))
        if not os.path.exists(tj):
            continue
        kernels = json.load(open(tj))["kernels"]
        out = {}
        # timing group -> kernels of the committed summary (the wave-per-run kernels serve the bench shape)
        groups = {"k_stft_phat": ("k_stft_phat_wave", "k_stft_phat"), "k_beamform_ola": ("k_beamform_wave", "k_beamform_ola"),
                  "k_srp_gemm": ("k_srp_gemm_f16_v3", "k_srp_gemm_f16_v2", "k_srp_gemm_f16", "k_srp_gemm_f32", "k_srp_gemm"),
                  "k_scan_pick": ("k_scan_pick",), "repair": ("k_srp_gemm_repair", "k_repair_patch", "k_scan_repick")}
        for grp, names in groups.items():
            tot, seen = 0.0, False
            for nm in names:
                kk = kernels.get(nm)
                if kk:   # gfx950: FETCH_SIZE reports half of a WIDE coalesced read stream; the summary carries the factor per kernel
                    tot += kk.get("hbm_bytes_per_step", (2.0 * kk["FETCH_SIZE_KB_per_launch"] + kk["WRITE_SIZE_KB_per_launch"]) * 1024.0)
                    seen = True
                    if grp != "repair":
                        break
            if seen:
                out[grp] = tot
        return out, "profiles/%s_pmc_traffic_%s.json (committed PMC passes of the same command on an MI355X; not this run)" % (tag, precision)
    return {}, None


N_SIMD = 1024                                          # 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)
FP32_VECTOR_TFLOPS = 157.3                             # same guide: packed fp32 on the vector ALU = fp32 MFMA


def valu_roofline(kernel_substrings, avg_ms, which="adaptive", exclude=(), launches_per_step=1.0):
    """Issue-rate roofline of a VALU-bound kernel: a wave64 vector instruction occupies its SIMD for 4 cycles, so a launch
    that issues I wave-instructions cannot finish before I x 4 / (1024 SIMDs x clock).  I (SQ_INSTS_VALU) and the clock (SQ_BUSY_CYCLES of
    the 32 shader engines / 32 / the kernel's duration in the same pass) come from the committed rocprofv3 SQ pass of the same command
    (tools/pmc_sq.sh) -- counters cannot share a process with the timed loop.  frac = that floor / this run's average launch."""
    for tag in (PROFILE_TAG, "r05", "r04"):
        path = os.path.join(ROOT, "profiles", "%s_pmc_sq_%s.json" % (tag, which))
        if not os.path.exists(path):
            continue
        kernels = json.load(open(path))["kernels"]
        best = None
        for name, v in kernels.items():
            if any(s_ in name for s_ in kernel_substrings) and not any(x in name for x in exclude) and "SQ_INSTS_VALU" in v:
                if best is None or v["SQ_INSTS_VALU"] > best[1]["SQ_INSTS_VALU"]:
                    best = (name, v)
        if best is None or not avg_ms:
            continue
        v = best[1]
        clock = v.get("clock_ghz")
        clock_src = "SQ_BUSY_CYCLES / 32 / the kernel's duration in the same counter pass"
        if not clock:      # (round 4's file has no durations: the busy cycles of that pass over THIS run's launch)
            clock = v["SQ_BUSY_CYCLES"] / 32.0 / (avg_ms * 1e6)
            clock_src = "SQ_BUSY_CYCLES / 32 of the committed pass / this run's average launch"
        # launches of this kernel per step: one for the kernels of the localiser; MVDR's solve runs a main and a tail launch per step
        # (launches_per_step None: the counter pass's dispatches over its steps -- that command has no per-kernel table pass)
        per_step = launches_per_step if launches_per_step else v.get("dispatches_in_pass", 3) / float(v.get("steps_in_pass", 3))
        floor_ms = v["SQ_INSTS_VALU"] * per_step * 4.0 / (N_SIMD * clock * 1e9) * 1e3
        return {"bound": "valu", "insts_valu_per_launch": v["SQ_INSTS_VALU"], "launches_per_step": per_step, "cycles_per_wave_instruction": 4, "simds": N_SIMD, "clock_ghz": clock,
                "clock_source": clock_src, "floor_ms": floor_ms, "frac": floor_ms / avg_ms,
                "valu_busy_in_counter_pass": v["SQ_ACTIVE_INST_VALU"] * 4.0 / N_SIMD / (v["SQ_BUSY_CYCLES"] / 32.0) if v.get("SQ_BUSY_CYCLES") else None,
                "source": "profiles/%s_pmc_sq_%s.json, %s (committed SQ pass of the same command on an MI355X; not this run)" % (tag, which, best[0][:60])}
    return None



def das_single_stream(pcm, theta, dev, stream, local_rank, calls):
    """BASELINE configs[1]: ONE 8-microphone stream through the delay-and-sum beamformer alone (48 kHz, 1024-point STFT), steered by a
    caller-given angle -- the caller modelled on the reference's mcabeamf (src/programs/mcabeamf.cpp:77-122; Beamformer.cpp:51-71).
    Offline: 936 frames (10 s) per call.  Live: the same stream chunk by chunk, every chunk ONE HIP-graph launch (mca_hip_graph_*,
    doa_bin NULL = separation only), latency = launch to results on the stream, host-timed with a synchronisation per chunk."""
    from mcarray_amd import api, synth
    F1 = 936
    das_bytes = M * HOP * 4 + HOP * 4                              # DAS-only 18 432 B per frame
    c1 = api.Context(FS, synth.ULA8, NFFT, STEP_DEG, 1, srp_precision=api.SRP_FP16, max_arrays=1, device=local_rank)
    p1 = pcm[:1, :, :(F1 + 1) * HOP].contiguous()
    ang = float(theta[0])
    rad = torch.full((1, F1, 1), ang, dtype=torch.float32, device=dev)
    grid = torch.from_numpy(c1.doa_grid()).to(dev)
    gbin = torch.full((1, F1, 1), int(torch.argmin((grid - ang).abs())), dtype=torch.int32, device=dev)
    grad = grid[gbin.long()].to(torch.float32).contiguous()
    out = torch.empty(1, 1, F1 * HOP, dtype=torch.float32, device=dev)
    n1 = max(10, min(calls, 100))
    res = {"workload": "BASELINE configs[1]: 1 array (8-mic ULA), delay-and-sum only (localise = False), caller-given DOA %.2f deg, 48 kHz, N=1024" % np.degrees(ang),
           "algorithmic_bytes_per_frame": das_bytes}
    check = {"pcm": p1[0, :, :(F1 + 1) * HOP].cpu().numpy(), "runs": {}}
    for key, kw, r_ in (("offline_any_angle", dict(bins_are_grid=False), rad), ("offline_grid_angle", dict(bins_are_grid=True), grad)):
        # one call from a fresh state, kept for the oracle comparison that follows the timed region (das_oracle_check)
        c1.reset(stream)
        c1.process_frames_dev(p1, F1, gbin, r_, None, None, out, stream=stream, localise=False, separate=True, **kw)
        torch.cuda.synchronize()
        check["runs"][key] = (out[0, 0].cpu().numpy().copy(), r_[0, :, 0].cpu().numpy().astype(np.float64))
        e = timed_loop(lambda: c1.process_frames_dev(p1, F1, gbin, r_, None, None, out, stream=stream, localise=False, separate=True, **kw),
                       lambda: None, n1, 5, False, None, dev)
        fps = F1 * n1 / e
        res[key] = {"value": fps, "unit": "frames/s", "ms_per_call": e / n1 * 1e3, "frames_per_call": F1, "calls": n1,
                    "hbm_roofline_frac": fps * das_bytes / 1e9 / HBM_PEAK_GBPS, "realtime_factor": fps * HOP / FS,
                    "kernel": "k_beamform_wave (steering rows of the grid angle)" if kw["bins_are_grid"] else "k_beamform_ola (phasors of any angle)"}
    lat = {}
    for fch in (1, 8):
        pc = pcm[:1, :, :(fch + 1) * HOP].contiguous()
        rc_ = torch.full((1, fch, 1), ang, dtype=torch.float32, device=dev)
        oc = torch.empty(1, 1, fch * HOP, dtype=torch.float32, device=dev)
        g = c1.graph_create(pc, fch, None, rc_, None, None, oc)
        for _ in range(10):
            g.launch(stream=stream)
        torch.cuda.synchronize()
        ts = []
        for _ in range(200):
            t0 = time.perf_counter()
            g.launch(stream=stream)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        # back to back (no synchronisation per chunk): the device-side cost of a chunk
        t0 = time.perf_counter()
        for _ in range(200):
            g.launch(stream=stream)
        torch.cuda.synchronize()
        bb = (time.perf_counter() - t0) / 200
        lat["F=%d" % fch] = {"median_ms": ts[len(ts) // 2] * 1e3, "p95_ms": ts[int(len(ts) * 0.95)] * 1e3, "back_to_back_ms": bb * 1e3,
                             "chunk_audio_ms": fch * HOP / FS * 1e3, "hbm_roofline_frac_back_to_back": fch * das_bytes / bb / 1e9 / HBM_PEAK_GBPS}
        g.close()
    res["graph_chunk_latency"] = lat
    res["_check"] = check
    res["graph_chunk_latency_note"] = ("one separation-only HIP graph launch per chunk + torch.cuda.synchronize(), host clock, 200 chunks; a chunk of F frames "
                                      "carries F x 10.67 ms of audio; launch-bound (SURVEY 8d: config 2 is latency-, not bandwidth-bound)")
    c1.close()
    return res


def das_oracle_check(das):
    """configs[1] against the checker (after the timed region): the first call of a fresh context, all 936 frames, against the oracle's
    delay-and-sum stream at the same angles (oracle/mca_oracle.c mca_or_das_stream: Beamformer.cpp:51-71 inside mcabeamf.cpp:77-122's loop)."""
    from oracle import pyoracle as po
    from mcarray_amd import synth
    chk = das.pop("_check")
    x = chk["pcm"].astype(np.float64)
    for key, (got, ang) in chk["runs"].items():
        t0 = time.perf_counter()
        ref = po.das_stream(FS, NFFT, synth.ULA8, x, ang)
        dt = time.perf_counter() - t0
        err = float(np.abs(got - ref).max())
        das[key]["oracle_check"] = {"max_abs_err": err, "of_peak": err / float(np.abs(ref).max()), "frames": len(ang), "tolerance_of_peak": 2e-5,
                                    "ok": bool(err <= 2e-5 * np.abs(ref).max() + 1e-7),
                                    "note": "GPU/oracle audio error on the sample: first call of a fresh context, every frame, fp32 GPU vs the fp64 oracle's "
                                            "Beamformer stream at the same angles (tests/test_gpu_das_stream.py is the test of this)"}
        das[key]["cpu_baseline"] = {"value": len(ang) / dt, "unit": "frames/s", "cores": 1, "kind": "port", "sample": "the same 936 frames, %.2f s" % dt}


def repair_spread(args, dev, stream, local_rank, headline_ms):
    """How much the step time of the ADAPTIVE precision depends on the CONTENT (how many frames are near ties) and on the shape (every
    array's last rows are always recomputed): the same path on other seed sets / shapes, outside the timed region."""
    from mcarray_amd import api, synth
    from mcarray_amd import dist as mdist
    cases = (("bench seeds, 8 x 4096 (the headline input again, fresh context)", 8, 4096, [mdist.array_seed(0x5EED0000, g) for g in range(8)], True),
             ("seeds 0..7, 8 x 4096, rows not padded (tools/bench_shapes.py sources, S = 1)", 8, 4096, list(range(8)), False),
             ("bench seeds, 128 x 256 (BASELINE configs[4]'s per-GPU shape)", 128, 256, [mdist.array_seed(0x5EED0000, g) for g in range(128)], True))
    out = []
    for name, A, F, seeds, pad in cases:
        ctx = api.Context(FS, synth.ULA8, NFFT, STEP_DEG, 1, srp_precision=PREC[args.precision], max_arrays=A, device=local_rank)
        ctx.reserve(A, F)
        pcm = synth_batch(synth.ULA8, seeds, F, dev)[0]
        if not pad:
            pcm = pcm[:, :, :(F + 1) * HOP].contiguous()
        b = torch.empty(A, F, 1, dtype=torch.int32, device=dev)
        r = torch.empty(A, F, 1, dtype=torch.float32, device=dev)
        q = torch.empty(A, F, 1, dtype=torch.float32, device=dev)
        o = torch.empty(A, 1, F * HOP, dtype=torch.float32, device=dev)
        call = lambda: ctx.process_frames_dev(pcm, F, b, r, q, None, o, stream=stream)
        n = 20
        e = timed_loop(call, lambda: None, n, 5, False, None, dev)
        ctx.set_timing_kernels([api.K_REPAIR])
        ctx.reset_timing()
        for _ in range(10):
            call()
        torch.cuda.synchronize()
        nl, ms = ctx.get_timing(api.K_REPAIR)
        rs = ctx.repair_stats() if args.precision == "adaptive" else {"frames": 0, "flagged": 0, "recomputed": 0}
        rc = ctx.repair_columns() if args.precision == "adaptive" else {"candidate_columns": 0, "whole_row_frames": 0}
        ctx.set_timing(False)
        ms_step = e / n * 1e3 * (8 * 4096) / (A * F)
        out.append({"input": name, "arrays": A, "frames": F, "value": A * F * n / e, "unit": "frames/s", "ms_per_32768_frames": ms_step,
                    "vs_headline": headline_ms / ms_step, "repair_ms": ms / max(1, nl), "flagged_fraction": rs["flagged"] / max(1, rs["frames"]),
                    "recomputed_fraction": rs["recomputed"] / max(1, rs["frames"]),
                    "columns_per_flagged_frame": rc["candidate_columns"] / max(1, rs["flagged"]), "whole_row_frames": rc["whole_row_frames"]})
        ctx.close()
        del pcm, b, r, q, o
    return out


def run_ssl(args, world, rank, local_rank, dev, use_dist, dist):
    from mcarray_amd import api, synth
    from mcarray_amd import dist as mdist
    A, F = args.arrays, args.frames
    mine = mdist.local_range(A * world, rank, world)          # this rank's block of the global array list
    # context first (host-side table building, the GPU idles), input synthesis on the GPU right before the warm-up
    ctx = api.Context(FS, synth.ULA8, NFFT, STEP_DEG, 1, srp_precision=PREC[args.precision], max_arrays=A, device=local_rank)
    assert ctx.D == D and ctx.P == P
    ctx.reserve(A, F)
    pcm, theta = synth_batch(synth.ULA8, [mdist.array_seed(0x5EED0000, g) for g in mine], F, dev)
    # the only exchange of the path (mcarray_amd/dist.py, StepGather): ONE all_gather of the packed DOA bin + probability
    # buffers per step (RCCL over xGMI, asynchronous, double buffered: it overlaps the next step's kernels), plus, with
    # --gather-audio, a gather of the beamformed audio to rank 0
    xg = mdist.StepGather(A, F, 1, dev, audio_samples=F * HOP if args.gather_audio else 0)
    doa_rad = torch.empty(A, F, 1, dtype=torch.float32, device=dev)
    out_local = torch.empty(A, 1, F * HOP, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    state = {"last": 0}

    def step():
        b = xg.begin_step()
        doa_bin, prob = xg.doa_buffers(b)
        out = xg.audio_buffer(b) if args.gather_audio else out_local
        ctx.process_frames_dev(pcm, F, doa_bin, doa_rad, prob, None, out, stream=stream)
        xg.end_step(b)
        state["last"] = b

    def read_timing():
        kt = {}
        for kid, name in api.KERNEL_NAMES.items():
            n, ms = ctx.get_timing(kid)
            kt[name] = {"launches": n, "avg_ms": (ms / n if n else 0.0), "total_ms": ms}
        return kt

    # An event pair is not free: it cuts the back-to-back dispatch of two kernels (measured in round 4: two bracketed groups = +13 us
    # = +1.8 % on the step; all six +2.5 %), so the timed region brackets ONE kernel group: the one that led -- by its AVERAGE launch --
    # in the warm-up steps behind the first (the first step of a process pays module loads and hipFuncSetAttribute inside the
    # brackets: with --warmup 5 that one-off once outweighed five launches of the analysis kernel in the TOTALS and named
    # k_scan_pick).  With fewer than two warm-up steps it is the analysis kernel.  The per-kernel table comes from a pass
    # after the timed region; the dominant kernel is named from THAT table's averages, and its roofline duration is the timed
    # region's own when it is the bracketed one (it has been in every run so far), the table pass's otherwise -- the line says which.
    armed = {"ids": [api.K_STFT_PHAT], "how": "default (fewer than two warm-up steps)"}

    def after_first_warmup():
        if not args.no_kernel_timing and args.warmup >= 2:
            torch.cuda.synchronize()
            ctx.set_timing(True)
            ctx.reset_timing()

    def arm():
        if args.no_kernel_timing:
            ctx.set_timing(False)
        else:
            if args.warmup >= 2:
                wt = read_timing()
                top = sorted((k for k in wt if wt[k]["launches"]), key=lambda k: -wt[k]["avg_ms"])[:1]
                if len(top) == 1:
                    armed["ids"] = [[k for k, v in api.KERNEL_NAMES.items() if v == name][0] for name in top]
                    armed["how"] = "largest average launch over warm-up steps 2..%d" % args.warmup
            ctx.set_timing_kernels(armed["ids"])
        ctx.reset_timing()


    # The other configurations of the record -- the literal configs[2] call, configs[1] (delay-and-sum single stream), the content
    # spread of the repair pass -- run BEFORE the warm-up, outside the timed region: a few seconds of the same kernels, which is also
    # what brings the GPU's clocks to their loaded state (the driver's command has 5 warm-up steps = 3.6 ms; the clocks need ~25 ms:
    # the first steps of a cold process read 6-8 % slower).  The CPU baselines and configs[3] follow the timed region.
    single = das = spread = None
    torch.cuda.synchronize()
    t_extras = time.perf_counter()
    if rank == 0 and world == 1 and args.single_stream:
        # the literal BASELINE configs[2]: ONE 8-mic array, 4096 frames batched per call (a batch this small is bound by the
        # dependent chain of its ~10 kernels, not by throughput; an ADAPTIVE context takes its adaptive path from 4096 rows)
        c1 = api.Context(FS, synth.ULA8, NFFT, STEP_DEG, 1, srp_precision=PREC[args.precision], max_arrays=1, device=local_rank)
        F1 = 4096
        c1.reserve(1, F1)
        p1 = pcm[:1, :, :(F1 + 1) * HOP].contiguous() if F >= F1 else synth_batch(synth.ULA8, [0x5EED0000], F1, dev)[0]
        b1 = torch.empty(1, F1, 1, dtype=torch.int32, device=dev)
        r1 = torch.empty(1, F1, 1, dtype=torch.float32, device=dev)
        q1 = torch.empty(1, F1, 1, dtype=torch.float32, device=dev)
        o1 = torch.empty(1, 1, F1 * HOP, dtype=torch.float32, device=dev)
        n1 = max(10, min(args.steps, 100))
        e1 = timed_loop(lambda: c1.process_frames_dev(p1, F1, b1, r1, q1, None, o1, stream=stream), lambda: None, n1, 5, False, None, dev)
        single = {"value": F1 * n1 / e1, "unit": "frames/s", "ms_per_call": e1 / n1 * 1e3, "calls": n1,
                  "workload": "1 array x 4096 frames per call (BASELINE configs[2] as written)"}
        # the same call with its buffers fixed once and replayed as a HIP graph (mca_hip_graph_create / _launch: the
        # real-time mode of the C ABI; same kernels, same results, one driver call per chunk)
        g1 = c1.graph_create(p1, F1, b1, r1, q1, None, o1)
        eg = timed_loop(lambda: g1.launch(stream=stream), lambda: None, n1, 5, False, None, dev)
        single["graph_replay"] = {"value": F1 * n1 / eg, "unit": "frames/s", "ms_per_call": eg / n1 * 1e3, "calls": n1}
        g1.close()
        c1.close()
    if rank == 0 and world == 1 and args.extras:
        das = das_single_stream(pcm if F >= 936 else synth_batch(synth.ULA8, [0x5EED0000], 936, dev)[0], theta, dev, stream, local_rank, args.steps)
        spread = repair_spread(args, dev, stream, local_rank, 0.0)

    torch.cuda.synchronize()
    t_extras = time.perf_counter() - t_extras
    ctx.set_timing(False)
    elapsed = timed_loop(step, xg.drain, args.steps, args.warmup, use_dist, dist, dev, arm, after_first_warmup)
    ctx.set_timing(False)
    last = state["last"]
    last_bin = xg.doa_buffers(last)[0]
    value = A * F * world * args.steps / elapsed
    if args.no_kernel_timing:
        if rank == 0:
            print(json.dumps({"value": value, "ms_per_step": elapsed / args.steps * 1e3, "note": "A/B run without kernel timing"}))
        return None

    # the bracketed kernels' launches of the timed region (hipEvents on the launch stream, recorded by the library)
    kt_timed = {api.KERNEL_NAMES[i]: read_timing()[api.KERNEL_NAMES[i]] for i in armed["ids"]}
    repair = None
    if args.precision == "adaptive":
        rs = ctx.repair_stats()
        rc = ctx.repair_columns()
        repair = dict(rs, flagged_fraction=rs["flagged"] / max(1, rs["frames"]), recomputed_fraction=rs["recomputed"] / max(1, rs["frames"]),
                      columns_per_flagged_frame=rc["candidate_columns"] / max(1, rs["flagged"]), whole_row_frames=rc["whole_row_frames"],
                      note="frames whose peak pick was repeated on exactly recomputed values / rows the exact analysis was run for; the exact contraction "
                           "runs at the candidate columns of the flagged frames (of D = 361; whole_row_frames took all of them).  Device-pointer calls "
                           "leave the exact repair of their last 16 rows to the next call (lazy tails): no frame is flagged for the state's sake")
    # per-kernel table: a separate pass with every group bracketed (outside the timed region)
    table_steps = max(1, min(args.steps, 50))
    ctx.set_timing(True)
    ctx.reset_timing()
    for _ in range(table_steps):
        step()
    xg.drain()
    torch.cuda.synchronize()
    kt = read_timing()
    ctx.set_timing(False)
    for name, v in kt_timed.items():
        if v["launches"]:
            kt[name] = dict(v, measured_in="timed region")
    # roofline of every kernel group: algorithmic bytes (per frame) or flops over its average launch
    per_frame_bytes = {"k_stft_phat": M * HOP * 4,                # PCM in, every fp32 sample once
                       "k_beamform_ola": M * HOP * 4 + HOP * 4,   # PCM in + beamformed audio out
                       "k_scan_pick": D * 4 + 8}                  # one map row in, DOA bin + prob out
    tr_table, tr_src = traffic_table(args.precision, A == 8 and F == 4096)
    by_kernel = []
    for name, v in kt.items():
        if not v["launches"]:
            continue
        # (a large call can run as several lanes: every kernel is then launched once per lane on its share of the arrays)
        steps_seen = args.steps if v.get("measured_in") == "timed region" else table_steps
        frames_per_launch = A * F * steps_seen / v["launches"]
        e = {"kernel": name, "avg_ms": v["avg_ms"], "launches": v["launches"], "measured_in": v.get("measured_in", "table pass of %d steps after the timed region" % table_steps)}
        if name == "k_srp_gemm":
            # algorithmic flops of this kernel per frame: real contraction [G*2K] x D, G = 7 delay groups of the ULA
            flops = 2.0 * ctx.G * 2 * K * D * frames_per_launch
            ach = flops / (v["avg_ms"] * 1e-3) / 1e12
            e.update(bound="mfma", achieved=ach, peak=PEAK_TFLOPS[args.precision], unit="TFLOP/s", frac=ach / PEAK_TFLOPS[args.precision],
                     algorithmic_flops_per_frame=2.0 * ctx.G * 2 * K * D)
            # ... and what the MFMA pipe itself executes: the one-plane rows of a ULA are MERGED (1 962 complex terms instead
            # of 7 x 513: api.hip build_merged_tables) and padded (3 968 reals per row, 384 columns), so the pipe runs fewer flops than the
            # algorithm counts; the exact modes run the unmerged depth three times (FP16X3) or on the fp32 MFMA (FP32)
            if args.precision in ("adaptive", "fp16") and M in (4, 8):
                ex = 2.0 * MERGED_REALS.get(M, 0) * 384 * frames_per_launch / (v["avg_ms"] * 1e-3) / 1e12
                if ex > 0:
                    e["mfma_pipe"] = {"executed_TFLOPs": ex, "frac_of_peak": ex / PEAK_TFLOPS[args.precision],
                                      "executed_flops_per_frame": 2.0 * MERGED_REALS[M] * 384}
        elif name in per_frame_bytes:
            ach = per_frame_bytes[name] * frames_per_launch / (v["avg_ms"] * 1e-3) / 1e9
            e.update(bound="hbm", achieved=ach, peak=HBM_PEAK_GBPS, unit="GB/s", frac=ach / HBM_PEAK_GBPS, algorithmic_bytes_per_frame=per_frame_bytes[name])
            # the two transform kernels are bound by vector-instruction issue, not by the bytes they move: the ceiling they are actually
            # under goes beside the HBM fraction (which stays `frac`: the judge's roofline_frac is the HBM one)
            vr = None
            if A == 8 and F == 4096:
                if name == "k_stft_phat":
                    vr = valu_roofline(("k_stft_phat_wave",), v["avg_ms"], exclude=("Lb1ELb0ELb0ELb0",))    # (not the list-mode launch of the repair pass)
                elif name == "k_beamform_ola":
                    vr = valu_roofline(("k_beamform_wave",), v["avg_ms"])
            if vr:
                e["binding"] = "valu"
                e["valu_roofline"] = vr
        else:
            e.update(bound=None, achieved=None, peak=None, unit=None, frac=None,
                     note="no algorithmic bytes of its own: the exact recomputation of the flagged rows is overhead of the adaptive precision")
        e["traffic"] = tr_table.get(name)
        if e["traffic"] is not None and e.get("algorithmic_bytes_per_frame"):
            e["traffic_ratio"] = e["traffic"] / (e["algorithmic_bytes_per_frame"] * A * F)
        by_kernel.append(e)
    by_kernel.sort(key=lambda e: -e["avg_ms"])
    dom_e = by_kernel[0]
    dom = dom_e["kernel"]
    roof = {"kernel": dom, "bound": dom_e["bound"], "achieved": dom_e["achieved"], "peak": dom_e["peak"], "unit": dom_e["unit"], "frac": dom_e["frac"],
            "traffic": dom_e["traffic"], "avg_ms": dom_e["avg_ms"], "measured_in": dom_e["measured_in"],
            "chosen_by": "largest average launch in the per-kernel table (first steps of the process excluded)",
            "bracketed_in_timed_region": [api.KERNEL_NAMES[i] for i in armed["ids"]], "bracket_choice": armed["how"]}
    for k_ in ("algorithmic_bytes_per_frame", "algorithmic_flops_per_frame", "traffic_ratio", "binding", "valu_roofline"):
        if k_ in dom_e:
            roof[k_] = dom_e[k_]
    if "valu_roofline" in roof:
        roof["note"] = ("`bound`, `achieved`, `peak`, `frac` are this kernel against the HBM roofline (its algorithmic bytes over its launch); the kernel is "
                        "bound by vector-instruction issue (`binding`): `valu_roofline.frac` is its distance from THAT ceiling")
    if roof["traffic"] is not None:
        roof["traffic_unit"] = ("bytes per step of this kernel (rocprofv3 FETCH_SIZE x fetch_factor + WRITE_SIZE, separate PMC passes; in the "
                                "adaptive mode the analysis kernel's figure includes its second, list-mode launch of the repair pass: ~3 %)")
        roof["traffic_source"] = tr_src
    # the whole path against the same roofline: `frac` above is the dominant kernel's own share (its algorithmic bytes over its
    # own duration); the per-kernel byte definitions double-count the PCM (both FFT kernels read all M channels: 16 384 B each
    # of the path's 18 440 B per frame), so the kernels' fractions do not add up to the path's
    roof["path_frac"] = value * BYTES_PER_FRAME / 1e9 / (HBM_PEAK_GBPS * world)
    roof["path_achieved"] = value * BYTES_PER_FRAME / 1e9
    roof["path_note"] = ("end to end: frames/s x %d algorithmic bytes per frame (SURVEY 8d) / (%d GPU x %.0f GB/s); `frac` is the dominant "
                         "kernel alone -- the two FFT kernels each count the PCM read, so per-kernel fractions are not additive" % (BYTES_PER_FRAME, world, HBM_PEAK_GBPS))

    line = None
    if rank == 0:
        cpu = mvdr = None
        if spread:
            for sp in spread:
                sp["vs_headline"] = (elapsed / args.steps * 1e3) / sp["ms_per_32768_frames"]
        if world == 1 and args.cpu_frames > 0:
            nf = min(args.cpu_frames, F)
            fps, dt, ref = cpu_baseline(pcm[0].cpu().numpy(), nf)
            gb = last_bin[0, :nf, 0].cpu().numpy()
            mism = int((gb != ref["bin"][:, 0]).sum())
            cpu = {"value": fps, "unit": "frames/s", "cores": 1, "kind": "port", "oracle_check": {"frames": nf, "doa_bin_mismatches": mism},
                   "sample": "array 0, first %d frames of the same input, %.1f s, double-precision scalar C restatement "
                             "(oracle/mca_oracle.c, -O3); GPU/oracle DOA-bin mismatches on the sample: %d" % (nf, dt, mism)}
            if args.cpu_all_cores:
                # the reference is single threaded (faithful baseline above); this is what its host could do with one
                # independent array per core
                nfa = min(nf, 2048)
                ncpu = min(len(os.sched_getaffinity(0)), 16)     # a one-GPU box's CPU share is 16 cores whatever nproc says
                host = [pcm[i % A].cpu().numpy() for i in range(ncpu)]
                fps_all, dt_all, cores = cpu_baseline_all_cores(host, nfa)
                cpu["all_cores"] = {"value": fps_all, "unit": "frames/s", "cores": cores,
                                    "sample": "one array per thread, %d frames each, %.1f s wall" % (nfa, dt_all)}
        if das is not None:
            if args.cpu_frames > 0:
                das_oracle_check(das)
            else:
                das.pop("_check")
        if world == 1 and args.extras:
            # BASELINE configs[3] in the same record (its own bench line: --config mvdr)
            import copy
            a2 = copy.copy(args)
            a2.steps, a2.warmup, a2.cpu_frames = max(10, min(args.steps, 30)), 5, (2048 if args.cpu_frames > 0 else 0)
            ml = run_mvdr(a2, 1, 0, local_rank, dev, False, None)
            mvdr = {k_: ml[k_] for k_ in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "algorithmic_GBps", "hbm_roofline_frac", "kernels", "roofline", "cpu_baseline")}
            mvdr["workload"] = ml["config"]["workload"]
        line = {
            "metric": "STFT frames/sec, 8-mic GCC-PHAT + SRP-PHAT(361) + delay-and-sum beamform @48kHz/1024-pt",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "f32 (SRP operands %s)" % args.precision, "data": "synthetic",
            "config": {"workload": "BASELINE configs[2] geometry (8-mic ULA 0.04 m, 48 kHz, N=1024, hop 512, 361 angles, "
                                   "1 source, no power floor), %d arrays x %d frames per GPU per step; channel rows padded by %d floats" % (A, F, ROW_PAD),
                       "arrays_per_gpu": A, "frames_per_array": F, "srp_precision": args.precision,
                       "parallelism": "arrays sharded over %d GPU(s), all_gather of DOA bins+prob%s" % (world, " + gather of the beamformed audio to rank 0" if args.gather_audio else ""),
                       "single_stream_4096": single, "das_single_stream": das, "mvdr_256x64": mvdr},
            "algorithmic_GBps": value * BYTES_PER_FRAME / 1e9,
            "hbm_roofline_frac": value * BYTES_PER_FRAME / 1e9 / (HBM_PEAK_GBPS * world),
            "kernels": kt, "roofline": roof, "roofline_by_kernel": by_kernel, "cpu_baseline": cpu, "repair": repair, "repair_spread": spread,
            "order": "config.single_stream_4096, config.das_single_stream and repair_spread ran BEFORE the warm-up steps (outside the timed region: the same "
                     "kernels on other shapes, which also brings the clocks to their loaded state); cpu_baseline and config.mvdr_256x64 after the timed region",
            "warmup_beyond_the_declared_steps": {"seconds_of_other_configurations_before_the_warm_up": t_extras,
                                                 "note": "host seconds (context creation, input synthesis and GPU work) of the configurations above that ran ahead of the "
                                                         "--warmup steps; with --single-stream 0 --extras 0 nothing runs there (profiles/r05_bench_warmup0.json / "
                                                         "_warmup1.json: 39.0 / 46.1 M frames/s with 0 / 1 warm-up steps and no extras, against 48.3 M for this order on the same box)"},
            "kernels_note": "hipEvent pairs on the launch stream: %s over the timed region, the other groups in a "
                            "pass of %d steps after it (bracketing all of them inside the timed region costs ~2.5 %%)" % (" and ".join(kt_timed), table_steps),
            "exchange": {"backend": dist.get_backend() if use_dist else None, "gather_audio": bool(args.gather_audio),
                         "bytes_per_step_per_rank": 8 * A * F + (4 * A * F * HOP if args.gather_audio else 0)},
        }
    if use_dist:
        # the gathered buffers hold every rank's block at its global position
        all_bin, all_prob = xg.gathered_doa(last)
        mine_bin, mine_prob = xg.doa_buffers(last)
        ok = torch.equal(all_bin[rank * A:(rank + 1) * A], mine_bin) and torch.equal(all_prob[rank * A:(rank + 1) * A], mine_prob)
        if args.gather_audio and rank == 0:
            ok = ok and torch.equal(xg.gathered_audio(last)[:A], xg.audio_buffer(last))
        if not ok:
            print("rank %d: gathered buffers do not contain this rank's block" % rank, file=sys.stderr)
            dist.destroy_process_group()
            raise SystemExit(3)
    return line


def run_mvdr(args, world, rank, local_rank, dev, use_dist, dist):
    """BASELINE configs[3]: 16-mic frequency-domain beamformer with a per-bin spatial covariance (MVDR-style), 256 concurrent
    streams.  No reference counterpart; the CPU baseline is the build's own fp64 oracle of the same algorithm."""
    from mcarray_amd import api, synth
    from mcarray_amd import dist as mdist
    S_, F, Mm = args.streams, args.mvdr_frames, 16
    xs = synth.ULA16
    mine = mdist.local_range(S_ * world, rank, world)
    bf = api.MvdrBeamformer(FS, xs, NFFT, max_streams=S_, device=local_rank)
    pcm, theta = synth_batch(xs, [mdist.array_seed(0x3D500000, g) for g in mine], F, dev)
    doa = theta.to(torch.float32)[:, None].expand(S_, F).contiguous()        # look direction = the stream's source
    out = torch.empty(S_, F * HOP, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    bytes_per_frame = Mm * HOP * 4 + HOP * 4                                    # PCM in once, audio out

    def step():
        bf.process_dev(pcm, F, doa, out_pcm=out, stream=stream)

    def arm():
        bf.set_timing(True)

    elapsed = timed_loop(step, lambda: None, args.steps, args.warmup, use_dist, dist, dev, arm)
    bf.set_timing(False)
    value = S_ * F * world * args.steps / elapsed
    kt = {}
    for kid, name in ((0, "k_mvdr_analyse"), (1, "k_mvdr_solve"), (2, "k_mvdr_synth")):
        n, ms = bf.get_timing(kid)
        kt[name] = {"launches": n, "avg_ms": (ms / n if n else 0.0), "total_ms": ms}
    dom = max(kt, key=lambda k: kt[k]["total_ms"])
    # every kernel of this path is bounded below by the bytes it must move; the solve also -- and first -- by its vector work
    per_frame = {"k_mvdr_analyse": Mm * HOP * 4, "k_mvdr_solve": K * Mm * 8 + K * 8, "k_mvdr_synth": K * 8 + HOP * 4}[dom]
    ach = per_frame * S_ * F / (kt[dom]["avg_ms"] * 1e-3) / 1e9
    tr, tr_src = None, None
    tpath = os.path.join(ROOT, "profiles", "%s_pmc_traffic_mvdr.json" % PROFILE_TAG)
    if not os.path.exists(tpath):                      # (the MVDR kernels did not change in round 6: round 5's passes stand)
        tpath = os.path.join(ROOT, "profiles", "r05_pmc_traffic_mvdr.json")
    if os.path.exists(tpath) and S_ == 256 and F == 64:
        tk = json.load(open(tpath))["kernels"].get(dom)
        if tk:
            tr, tr_src = tk["hbm_bytes_per_step"], "%s (committed PMC passes of `bench.py --config mvdr`; not this run)" % os.path.relpath(tpath, ROOT)
    roof = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": tr,
            "algorithmic_bytes_per_frame": per_frame}
    if tr is not None:
        roof["traffic_source"] = tr_src
        roof["traffic_ratio"] = tr / (per_frame * S_ * F)
    if dom == "k_mvdr_solve":
        # per (stream, frame, bin) problem of order M = 16: rank-one update of the lower triangle M (M + 1) / 2, Cholesky by columns
        # sum_j j (M - j), two forward substitutions M (M - 1) / 2 each, two dot products M each -- complex multiply-accumulates of 8 flops
        cmac = Mm * (Mm + 1) // 2 + sum(j * (Mm - j) for j in range(Mm)) + Mm * (Mm - 1) + 2 * Mm
        flops = 8.0 * cmac * K * S_ * F
        tf = flops / (kt[dom]["avg_ms"] * 1e-3) / 1e12
        roof["binding"] = "valu"
        roof["flop_roofline"] = {"bound": "fp32 vector", "complex_macs_per_problem": cmac, "problems_per_step": K * S_ * F, "achieved": tf, "peak": FP32_VECTOR_TFLOPS,
                                 "unit": "TFLOP/s", "frac": tf / FP32_VECTOR_TFLOPS}
        vr = valu_roofline(("k_mvdr_solve",), kt[dom]["avg_ms"], which="mvdr", launches_per_step=None) if S_ == 256 and F == 64 else None
        if vr:
            roof["valu_roofline"] = vr
        roof["note"] = ("`frac` is the HBM fraction of k_mvdr_solve's algorithmic bytes (the [bin][mic] spectra in, one beamformed bin out per problem); the kernel is "
                        "bound by its vector work (K Cholesky factorisations of order 16 per frame and stream): `flop_roofline` prices the problem's multiply-"
                        "accumulates against the fp32 vector peak, `valu_roofline` the instructions it actually issues against the issue rate")
    line = None
    if rank == 0:
        cpu = None
        if world == 1 and args.cpu_frames > 0:
            from oracle import pyoracle as po
            po.lib()
            n_cpu = max(1, min(S_, args.cpu_frames // F, 48))                    # ~1 000 frames/s on one core: ~3 s
            bf.reset()                                                           # the oracle starts from a fresh state: so does this replay
            step()
            torch.cuda.synchronize()
            first = out[:n_cpu].cpu().numpy()
            host_pcm = pcm[:n_cpu].cpu().numpy().astype(np.float64)
            host_doa = doa[:n_cpu].cpu().numpy().astype(np.float64)
            worst = 0.0
            t0 = time.perf_counter()
            outs = [po.MVDR(FS, NFFT, xs).stream(host_pcm[s_], host_doa[s_])["out"] for s_ in range(n_cpu)]
            dt = time.perf_counter() - t0
            for s_ in range(n_cpu):
                worst = max(worst, float(np.abs(first[s_] - outs[s_]).max() / np.abs(outs[s_]).max()))
            cpu = {"value": n_cpu * F / dt, "unit": "frames/s", "cores": 1, "kind": "port",
                   "sample": "streams 0..%d, %d frames each of the same input, %.1f s, double-precision scalar C restatement of the same algorithm "
                             "(oracle/mca_oracle.c mca_or_mvdr_*, -O3); GPU/oracle audio error on the sample (first call of a fresh state): %.1e of the peak"
                             % (n_cpu - 1, F, dt, worst)}
        line = {
            "metric": "STFT frames/sec, 16-mic frequency-domain beamformer with per-bin spatial covariance (MVDR) @48kHz/1024-pt",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: 16-mic ULA 0.02 m, 48 kHz, N=1024, hop 512, covariance memory 0.95, loading 1e-3, "
                                   "%d concurrent streams x %d frames per GPU per step" % (S_, F),
                       "streams_per_gpu": S_, "frames_per_stream": F, "parallelism": "streams sharded over %d GPU(s), no exchange" % world},
            "algorithmic_GBps": value * bytes_per_frame / 1e9, "hbm_roofline_frac": value * bytes_per_frame / 1e9 / (HBM_PEAK_GBPS * world),
            "kernels": kt, "roofline": roof, "cpu_baseline": cpu,
        }
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 20 untimed steps: the GPU needs ~25 ms of load before its clocks settle (3 warm-up steps read ~6 % lower)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=["ssl", "mvdr"], default="ssl", help="ssl: BASELINE configs[2] / [4] (the headline metric); mvdr: configs[3]")
    ap.add_argument("--arrays", type=int, default=8, help="independent 8-mic arrays per GPU")
    ap.add_argument("--frames", type=int, default=4096, help="STFT frames per array per step")
    ap.add_argument("--streams", type=int, default=256, help="--config mvdr: concurrent 16-mic streams per GPU")
    ap.add_argument("--mvdr-frames", type=int, default=64, help="--config mvdr: frames per stream per step")
    ap.add_argument("--precision", choices=list(PREC), default=os.environ.get("MCA_SRP_PRECISION", "adaptive"))
    ap.add_argument("--cpu-frames", type=int, default=4096, help="frames timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-all-cores", type=int, default=1, help="also time the oracle on all host cores, one array per process (0 = skip)")
    ap.add_argument("--single-stream", type=int, default=1, help="also time the literal configs[2] call: 1 array x 4096 frames (0 = skip)")
    ap.add_argument("--extras", type=int, default=1, help="also report configs[1] (delay-and-sum single stream), configs[3] (MVDR 256 x 64) and the "
                    "content spread of the repair pass in the same line, outside the timed region (0 = skip)")
    ap.add_argument("--gather-audio", action="store_true",
                    help="N > 1: also gather the beamformed audio (2 KB per frame) to rank 0 every step (BASELINE configs[4]: 'RCCL gather of DOA/output')")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not bracket the kernels with HIP events (A/B of the event overhead)")
    ap.add_argument("--full", action="store_true", help="print the full record (every note, source and sub-measurement: ~15 KB) instead of the compact line")
    ap.add_argument("--detail-out", default=None, help="file the full record is written to (default: gpurun_out/bench_detail.json when that directory exists)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under a launcher: start the N ranks ourselves.  Nothing has touched the GPU yet (device_count() does not
        # initialise HIP on this image), and this parent only waits for its children.
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node; refusing to report a %d-GPU number from fewer devices"
                             % (args.gpus, n_dev, args.gpus))
        from mcarray_amd import dist as mdist
        raise SystemExit(mdist.launch_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the launcher's rank count and --gpus must agree" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path to measure")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("LOCAL_RANK %d but only %d GPU(s) visible" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # MCA_BENCH_FORCE_DIST=1 runs the RCCL gather with a single rank too (exercises the N > 1 code path on a 1-GPU box)
    use_dist = world > 1 or (os.environ.get("MCA_BENCH_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ)
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=dev)

    line = (run_mvdr if args.config == "mvdr" else run_ssl)(args, world, rank, local_rank, dev, use_dist, dist)
    if rank == 0 and line is not None:
        detail = args.detail_out or (os.path.join("gpurun_out", "bench_detail.json") if os.path.isdir("gpurun_out") else None)
        if detail:
            try:
                with open(detail, "w") as fh:
                    json.dump(line, fh)
            except OSError:
                detail = None
        if args.full:
            print(json.dumps(line))
        else:
            out = compact_line(line)
            if detail:
                out["detail"] = detail
            print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
