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
PROFILE_TAG = "r03"                                    # profiles/<tag>_pmc_traffic_<precision>.json of the committed PMC passes


def synth_batch(xs, seeds, n_frames, device, noise=0.01):
    """Far-field white source per array + 20 dB sensor noise, generated on the GPU.
    One seed per array, derived from the array's GLOBAL index, so its data does not depend on the rank count."""
    n_arrays, n_mics = len(seeds), len(xs)
    L = (n_frames + 1) * HOP
    xs_t = torch.tensor(xs, device=device, dtype=torch.float64)
    theta = torch.empty(n_arrays, device=device, dtype=torch.float64)
    # channel rows are ROW_PAD floats longer than their (F + 1) * hop samples (the C ABI takes the strides): a row pitch of a power
    # of two plus a little -- 257 half frames = 2^19 + 2^11 bytes at 128 arrays x 256 frames -- lines the loads of all resident
    # waves up on the same few HBM channels (k_beamform_wave 0.33 instead of 0.295 ms)
    out = torch.zeros(n_arrays, n_mics, L + ROW_PAD, device=device, dtype=torch.float32)
    f = torch.fft.rfftfreq(L, d=1.0 / FS).to(device=device, dtype=torch.float64)
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


def timed_loop(step, drain, steps, warmup, use_dist, dist, dev, before_timed=None):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize on both sides; MAX over ranks."""
    for _ in range(warmup):
        step()
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


def traffic_from_profiles(roof, dom, precision, shape_matches):
    """HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes (tools/pmc_traffic.sh): NOT measured in this
    run -- a --pmc pass serialises the kernels and cannot share a process with the timed loop."""
    tj = os.path.join(ROOT, "profiles", "%s_pmc_traffic_%s.json" % (PROFILE_TAG, precision))
    if os.path.exists(tj) and shape_matches:
        kernels = json.load(open(tj))["kernels"]
        # timing group -> kernel of the committed summary (round 3: the wave-per-run kernels serve the bench shape)
        kk = kernels.get({"k_stft_phat": "k_stft_phat_wave", "k_beamform_ola": "k_beamform_wave"}.get(dom, dom)) or kernels.get(dom)
        if kk:   # gfx950: FETCH_SIZE reports half of a WIDE coalesced read stream; the summary carries the factor per kernel
            roof["traffic"] = kk.get("hbm_bytes_per_step", (2.0 * kk["FETCH_SIZE_KB_per_launch"] + kk["WRITE_SIZE_KB_per_launch"]) * 1024.0)
            roof["traffic_unit"] = ("bytes per step of this kernel (rocprofv3 FETCH_SIZE x fetch_factor + WRITE_SIZE, separate PMC passes; in the "
                                    "adaptive mode the analysis kernel's figure includes its second, list-mode launch of the repair pass: ~3 %)")
            roof["traffic_source"] = "profiles/%s_pmc_traffic_%s.json (committed PMC passes of the same command on an MI355X; not this run)" % (PROFILE_TAG, precision)


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

    # Every event pair costs the stream ~1.5 us (measured: all six kernel groups bracketed = +2.5 % on the step).  The warm-up
    # steps run with all groups bracketed and name the dominant kernel; the timed region brackets that kernel only (the
    # roofline's duration is measured live inside the timed region); the per-kernel table comes from a pass after it.
    dom_state = {"id": api.K_STFT_PHAT}

    def arm():
        if args.no_kernel_timing:
            ctx.set_timing(False)
        else:
            if args.warmup > 0:
                wt = read_timing()
                name = max(wt, key=lambda k: wt[k]["total_ms"])
                dom_state["id"] = [k for k, v in api.KERNEL_NAMES.items() if v == name][0]
            ctx.set_timing_kernels([dom_state["id"]])
        ctx.reset_timing()

    ctx.set_timing(not args.no_kernel_timing)
    elapsed = timed_loop(step, xg.drain, args.steps, args.warmup, use_dist, dist, dev, arm)
    ctx.set_timing(False)
    last = state["last"]
    last_bin = xg.doa_buffers(last)[0]
    value = A * F * world * args.steps / elapsed
    if args.no_kernel_timing:
        if rank == 0:
            print(json.dumps({"value": value, "ms_per_step": elapsed / args.steps * 1e3, "note": "A/B run without kernel timing"}))
        return None

    # the dominant kernel's launches of the timed region, bracketed with hipEvents on the launch stream by the library
    dom = api.KERNEL_NAMES[dom_state["id"]]
    kt_dom = read_timing()[dom]
    repair = None
    if args.precision == "adaptive":
        rs = ctx.repair_stats()
        repair = dict(rs, flagged_fraction=rs["flagged"] / max(1, rs["frames"]), recomputed_fraction=rs["recomputed"] / max(1, rs["frames"]),
                      note="frames whose peak pick was repeated on exactly recomputed rows / rows recomputed (includes the last frame of every array and call)")
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
    kt[dom] = kt_dom
    # (a large call can run as several lanes: every kernel is then launched once per lane on its share of the arrays)
    frames_per_launch = A * F * args.steps / max(1, kt[dom]["launches"])
    if dom == "k_srp_gemm":
        # algorithmic flops of this kernel per frame: real contraction [G*2K] x D, G = 7 delay groups of the ULA
        flops = 2.0 * ctx.G * 2 * K * D * frames_per_launch
        ach = flops / (kt[dom]["avg_ms"] * 1e-3) / 1e12
        roof = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": PEAK_TFLOPS[args.precision], "unit": "TFLOP/s",
                "frac": ach / PEAK_TFLOPS[args.precision], "traffic": None}
    else:
        per_frame = {"k_stft_phat": M * HOP * 4, "k_beamform_ola": M * HOP * 4 + HOP * 4, "k_scan_pick": D * 4 + 8}.get(dom, D * 4 + 8)
        ach = per_frame * frames_per_launch / (kt[dom]["avg_ms"] * 1e-3) / 1e9
        roof = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBPS, "traffic": None, "algorithmic_bytes_per_frame": per_frame}
    traffic_from_profiles(roof, dom, args.precision, A == 8 and F == 4096)
    # the whole path against the same roofline: `frac` above is the dominant kernel's own share (its algorithmic bytes over its
    # own duration); the per-kernel byte definitions double-count the PCM (both FFT kernels read all M channels: 16 384 B each
    # of the path's 18 440 B per frame), so the kernels' fractions do not add up to the path's
    roof["path_frac"] = value * BYTES_PER_FRAME / 1e9 / (HBM_PEAK_GBPS * world)
    roof["path_achieved"] = value * BYTES_PER_FRAME / 1e9
    roof["path_note"] = ("end to end: frames/s x %d algorithmic bytes per frame (SURVEY 8d) / (%d GPU x %.0f GB/s); `frac` is the dominant "
                         "kernel alone -- the two FFT kernels each count the PCM read, so per-kernel fractions are not additive" % (BYTES_PER_FRAME, world, HBM_PEAK_GBPS))

    line = None
    if rank == 0:
        cpu = single = None
        if world == 1 and args.cpu_frames > 0:
            nf = min(args.cpu_frames, F)
            fps, dt, ref = cpu_baseline(pcm[0].cpu().numpy(), nf)
            gb = last_bin[0, :nf, 0].cpu().numpy()
            mism = int((gb != ref["bin"][:, 0]).sum())
            cpu = {"value": fps, "unit": "frames/s", "cores": 1, "kind": "port",
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
        if world == 1 and args.single_stream:
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
        line = {
            "metric": "STFT frames/sec, 8-mic GCC-PHAT + SRP-PHAT(361) + delay-and-sum beamform @48kHz/1024-pt",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "f32 (SRP operands %s)" % args.precision, "data": "synthetic",
            "config": {"workload": "BASELINE configs[2] geometry (8-mic ULA 0.04 m, 48 kHz, N=1024, hop 512, 361 angles, "
                                   "1 source, no power floor), %d arrays x %d frames per GPU per step; channel rows padded by %d floats" % (A, F, ROW_PAD),
                       "arrays_per_gpu": A, "frames_per_array": F, "srp_precision": args.precision,
                       "parallelism": "arrays sharded over %d GPU(s), all_gather of DOA bins+prob%s" % (world, " + gather of the beamformed audio to rank 0" if args.gather_audio else ""),
                       "single_stream_4096": single},
            "algorithmic_GBps": value * BYTES_PER_FRAME / 1e9,
            "hbm_roofline_frac": value * BYTES_PER_FRAME / 1e9 / (HBM_PEAK_GBPS * world),
            "kernels": kt, "roofline": roof, "cpu_baseline": cpu, "repair": repair,
            "kernels_note": "hipEvent pairs on the launch stream: %s (the roofline kernel) over the timed region, the other groups in a "
                            "pass of %d steps after it (bracketing all of them inside the timed region costs ~2.5 %%)" % (dom, table_steps),
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
    # every kernel of this path is bounded below by the bytes it must move; the solve also by its VALU work
    per_frame = {"k_mvdr_analyse": Mm * HOP * 4, "k_mvdr_solve": K * Mm * 8 + K * 8, "k_mvdr_synth": K * 8 + HOP * 4}[dom]
    ach = per_frame * S_ * F / (kt[dom]["avg_ms"] * 1e-3) / 1e9
    roof = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": None,
            "algorithmic_bytes_per_frame": per_frame,
            "note": "k_mvdr_solve reads the [bin][mic] spectra and writes one beamformed bin per (stream, frame, bin); it is VALU-issue bound (K Cholesky "
                    "factorisations of order 16 per frame), which the HBM fraction makes visible"}
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
    ap.add_argument("--gather-audio", action="store_true",
                    help="N > 1: also gather the beamformed audio (2 KB per frame) to rank 0 every step (BASELINE configs[4]: 'RCCL gather of DOA/output')")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not bracket the kernels with HIP events (A/B of the event overhead)")
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
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
