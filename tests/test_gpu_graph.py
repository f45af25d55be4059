"""Real-time mode (mca_hip_graph_*, include/mcarray_hip.h): a live stream handed over chunk by chunk, the kernels of every
chunk replayed as one HIP graph.  The replay must be bit-identical to the eager stream calls on the same chunks (the same
kernels on the same buffers), continue the module state (E_prev of SteeringBeamforming.h:69, overlap-add tails, the power
gate of BeamformingSeparationAndLocalisation.cpp:55-87) across launches, and mix with eager calls."""
import numpy as np
import pytest

from mcarray_amd import api, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _bufs(A, M, F, hop, S, D, dev):
    return dict(pcm=torch.zeros((A, M, (F + 1) * hop), device=dev), bin=torch.zeros((A, F, S), dtype=torch.int32, device=dev),
                rad=torch.zeros((A, F, S), device=dev), prob=torch.zeros((A, F, S), device=dev),
                energy=torch.zeros((A, F, D), device=dev), out=torch.zeros((A, S, F * hop), device=dev))


@pytest.mark.parametrize("F,floor,audio", [(1, False, True), (8, False, True), (4, True, True), (8, False, False)])
def test_graph_replay_equals_eager_chunks(F, floor, audio):
    fs, N, A, S, n_chunks = 48000, 1024, 2, 1, 9
    hop = N // 2
    xs = synth.ULA8
    dev = torch.device("cuda:0")
    total = n_chunks * F
    pcm = np.stack([synth.noise_source_stream(xs, np.deg2rad(25.0 - 60 * a), fs, (total + 1) * hop, 70 + a) for a in range(A)])
    if floor:
        pcm[:, :, : 3 * F * hop] *= 1e-3            # a quiet lead-in for the floor estimation
    eager = api.Context(fs, xs, N, 0.5, S, use_power_floor=floor, max_arrays=A)
    graph = api.Context(fs, xs, N, 0.5, S, use_power_floor=floor, max_arrays=A)
    D = eager.D
    be, bg = _bufs(A, 8, F, hop, S, D, dev), _bufs(A, 8, F, hop, S, D, dev)
    g = graph.graph_create(bg["pcm"], F, bg["bin"], bg["rad"], bg["prob"], bg["energy"], bg["out"] if audio else None)
    st = torch.cuda.current_stream().cuda_stream
    for i in range(n_chunks):
        chunk = torch.from_numpy(pcm[:, :, i * F * hop:(i * F + F + 1) * hop].copy()).to(dev)
        be["pcm"].copy_(chunk); bg["pcm"].copy_(chunk)
        eager.process_frames_dev(be["pcm"], F, be["bin"], be["rad"], be["prob"], be["energy"], be["out"] if audio else None, stream=st)
        if i == 5:      # an eager call in between moves the state parity; the next launch records the other graph
            graph.process_frames_dev(bg["pcm"], F, bg["bin"], bg["rad"], bg["prob"], bg["energy"], bg["out"] if audio else None, stream=st)
        else:
            g.launch(st)
        torch.cuda.synchronize()
        for k in ("bin", "rad", "prob", "energy") + (("out",) if audio else ()):
            assert torch.equal(be[k], bg[k]), (i, k)
    if not floor:
        assert int(be["bin"].max()) > 1          # a real localisation happened (with the floor, 3 s of estimation gate everything)
    else:
        assert int(be["bin"].max()) == -1
    g.close()


def test_graph_rejects_bad_arguments():
    ctx = api.Context(48000, synth.ULA8, 1024, 5.0, 1)
    dev = torch.device("cuda:0")
    b = _bufs(1, 8, 2, 512, 1, ctx.D, dev)
    with pytest.raises(api.MCArrayHipError):
        ctx.graph_create(b["pcm"], 5, b["bin"], b["rad"], b["prob"])          # pcm too short for 5 frames
    with pytest.raises(api.MCArrayHipError):
        ctx.graph_create(b["pcm"], 2, b["bin"], None, b["prob"], None, b["out"])   # separation needs doa_rad


def test_graph_survives_workspace_growth_and_context_destruction():
    """A recorded graph bakes in the workspace pointers.  A later eager call with more arrays / frames reallocates the
    workspace: the next launch must notice (workspace generation counter), re-record and still equal the eager result.
    A context destroyed before its graph orphans it: launch fails cleanly, destroy is safe."""
    fs, N, A, F = 48000, 1024, 2, 4
    hop = N // 2
    xs = synth.ULA8
    dev = torch.device("cuda:0")
    n_chunks = 4
    pcm = np.stack([synth.noise_source_stream(xs, np.deg2rad(-35.0 + 50 * a), fs, (n_chunks * F + 1) * hop, 170 + a) for a in range(A)])
    eager = api.Context(fs, xs, N, 0.5, 1, max_arrays=A)
    graph = api.Context(fs, xs, N, 0.5, 1, max_arrays=A)
    be, bg = _bufs(A, 8, F, hop, 1, eager.D, dev), _bufs(A, 8, F, hop, 1, eager.D, dev)
    g = graph.graph_create(bg["pcm"], F, bg["bin"], bg["rad"], bg["prob"], bg["energy"], bg["out"])
    st = torch.cuda.current_stream().cuda_stream
    big = _bufs(A, 8, 700, hop, 1, eager.D, dev)
    big["pcm"].normal_(0, 0.1)
    for i in range(n_chunks):
        chunk = torch.from_numpy(pcm[:, :, i * F * hop:(i * F + F + 1) * hop].copy()).to(dev)
        be["pcm"].copy_(chunk); bg["pcm"].copy_(chunk)
        eager.process_frames_dev(be["pcm"], F, be["bin"], be["rad"], be["prob"], be["energy"], be["out"], stream=st)
        if i == 2:
            # grow the workspace of the graph's context behind the graph's back (a throw-away state: saved and restored)
            blob = graph.state_save()
            graph.process_frames_dev(big["pcm"], 700, big["bin"], big["rad"], big["prob"], big["energy"], big["out"], stream=st)
            torch.cuda.synchronize()
            graph.state_load(blob)
        g.launch(st)
        torch.cuda.synchronize()
        for k in ("bin", "rad", "prob", "energy", "out"):
            assert torch.equal(be[k], bg[k]), (i, k)
    graph.close()                                   # the context goes first
    with pytest.raises(api.MCArrayHipError):
        g.launch(st)
    g.close()
    eager.close()


def test_separation_only_graph_equals_the_eager_delay_and_sum_call():
    """doa_bin NULL (round 4): the graph records mca_hip_separate_frames_dev alone -- the delay-and-sum stream of the reference's
    mcabeamf (src/programs/mcabeamf.cpp:77-122, Beamformer.cpp:51-71) chunk by chunk at a caller-given angle; bit-identical to the
    eager calls, overlap-add carried from chunk to chunk, and equal to the ORACLE's delay-and-sum stream at that angle."""
    import torch
    from mcarray_amd import api, synth
    fs, N, Fc, n_chunks = 48000, 1024, 8, 5
    xs = synth.ULA8
    dev = torch.device("cuda:0")
    pcm = torch.from_numpy(synth.noise_source_stream(xs, np.deg2rad(-37.0), fs, (Fc * n_chunks + 1) * 512, 21)[None]).to(dev)
    ang = float(np.float32(np.deg2rad(-36.3)))          # not a grid angle (the float the GPU call receives)
    outs = {}
    for mode in ("eager", "graph"):
        ctx = api.Context(fs, xs, N, 0.5, 1, srp_precision=api.SRP_FP16, max_arrays=1)
        buf = torch.zeros(1, len(xs), (Fc + 1) * 512, dtype=torch.float32, device=dev)
        rad = torch.full((1, Fc, 1), ang, dtype=torch.float32, device=dev)
        o = torch.empty(1, 1, Fc * 512, dtype=torch.float32, device=dev)
        g = ctx.graph_create(buf, Fc, None, rad, None, None, o) if mode == "graph" else None
        got = []
        for i in range(n_chunks):
            buf.copy_(pcm[:, :, i * Fc * 512:((i + 1) * Fc + 1) * 512])
            if g:
                g.launch()
            else:
                ctx.process_frames_dev(buf, Fc, None, rad, None, None, o, localise=False, separate=True)
            torch.cuda.synchronize()
            got.append(o.cpu().numpy().copy())
        outs[mode] = np.concatenate(got, axis=2)
        if g:
            g.close()
        ctx.close()
    assert np.array_equal(outs["eager"], outs["graph"])
    # the oracle's Beamformer stream at the same angle (mca_or_das_stream: Beamformer.cpp:51-71 inside the loop of mcabeamf.cpp:77-122)
    from oracle import pyoracle as po
    ref = po.das_stream(fs, N, xs, pcm[0].cpu().numpy().astype(np.float64), ang)
    assert np.abs(outs["graph"][0, 0] - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-7
    # and one whole call over the same stream: the run boundaries differ, fp32 rounding only
    whole_ctx = api.Context(fs, xs, N, 0.5, 1, srp_precision=api.SRP_FP16, max_arrays=1)
    whole = torch.empty(1, 1, Fc * n_chunks * 512, dtype=torch.float32, device=dev)
    whole_ctx.process_frames_dev(pcm.contiguous(), Fc * n_chunks, None, torch.full((1, Fc * n_chunks, 1), ang, dtype=torch.float32, device=dev), None, None, whole,
                                 localise=False, separate=True)
    torch.cuda.synchronize()
    assert np.abs(whole.cpu().numpy() - outs["graph"]).max() <= 2e-6 * np.abs(outs["graph"]).max() + 1e-9
    assert np.abs(whole.cpu().numpy()[0, 0] - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-7
    whole_ctx.close()
