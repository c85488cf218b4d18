#!/usr/bin/env python
"""bench.py -- streaming-inference benchmark of the MI355X-native RVC engine.

Metric (BASELINE.json): audio frames/s (+ p99 per-chunk latency), 160 ms chunks @16 kHz.
A "step" is one 160 ms chunk of every stream of this rank through the whole per-chunk hot path
(ContentVec -> RMVPE f0 -> [retrieval] -> NSF-HiFiGAN); one chunk = 16 new 10 ms frames.

Headline (`value`): BASELINE configs[1] on every GPU -- 1 stream, v2/768 ContentVec + RMVPE + v2-48k synthesizer, retrieval off,
inputs resident in HBM, every chunk synchronised (per-chunk latency is the product).  Streams are independent, so with N GPUs every
rank runs the same per-GPU work (weak scaling, no per-chunk collective).  The same run also measures, as `sub_configs`:
  index100k  BASELINE configs[2]: + 100k x 768 flat-L2 retrieval (k = 4, rate 0.75); with N > 1 the index reaches every rank through
             the engine's own RCCL broadcast (rvc_index_broadcast)
  streams64  BASELINE configs[3] (N = 1) / configs[4] (N > 1: 64 streams per GPU, stream s of the job on rank s mod N, shared index
             broadcast over RCCL; also emitted as the top-level key `config4`): throughput mode, every stage batched over the streams
  streams64_index100k (N = 1)  one rank's work of configs[4] on one GPU: 64 streams + the 100k x 768 index
  streams{2,4,8,16,32}         where the chip saturates (ms / step, frames / s)
  v1_256     the literal reading of configs[1] ("ContentVec-256"): v1 models, 256-d features from layer 9 (enums.rs:10-23)
and the host-buffer boundary of the reference (`latency_ms_host_buffer`: rvc_infer with H2D + D2H inside the call = SURVEY 8d's
latency definition).  `latency_ms` of the headline comes from a soak of >= 1000 synchronised chunks inside the run (p50 / p99 / p99.9).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--streams S] [--index] [--only-headline] [--no-cpu] [--dry-launch]

The ONE JSON line on stdout is compact (it must survive a tail of ~8 KB): the contract keys, `roofline`, `cpu_baseline`, the box's own
peaks measured in-run (`peak_measured`: rvc_calibrate at the start and at the end of the run), one short record per sub-configuration
(ms per step, frames/s, p50 / p99, GPU ms, effective shader clock and socket power WHILE that leg ran, roofline fractions against the nominal
and against the measured peak) and -- as the LAST key -- `summary_ms`.  The verbose records (notes, kernel lists, sources of the traffic
figures, per-rank values) go to gpurun_out/bench_full.json (`full_record`) and, with --verbose-line, to stderr.

N > 1: either started by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / LOCAL_RANK / WORLD_SIZE
in the environment) or, without those variables, bench.py spawns the N ranks itself (one process per GPU, 127.0.0.1 rendezvous).
--dry-launch runs the launcher, the rendezvous (gloo), the stream sharding and the reduction of the per-rank results without
touching a GPU (CPU test of the multi-rank plumbing).
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

# hardware queues of the HIP runtime: must be in the environment before the runtime initialises (engine.hip, rvc_runtime_defaults)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
HBM_PEAK_GBS = 8000.0
FRAMES_PER_CHUNK = 16              # 160 ms chunk = 16 hops of 10 ms


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--streams", type=int, default=1, help="streams per GPU of the HEADLINE leg (default 1 = BASELINE configs[1])")
    ap.add_argument("--index", action="store_true", help="headline leg with the 100k x 768 index (BASELINE configs[2])")
    ap.add_argument("--only-headline", action="store_true", help="skip the sub_configs / informational legs")
    ap.add_argument("--graph", action="store_true", help="replay each chunk from a hipGraph (default: eager launches with interleaved branch submission, measured faster)")
    ap.add_argument("--no-graph", action="store_true", help="accepted for compatibility: eager launches are the default")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--dry-launch", action="store_true", help="multi-rank plumbing only (gloo, no GPU work)")
    ap.add_argument("--legs", default="", help="measurement aid: run only these sub_configs legs (comma-separated names; default: all)")
    ap.add_argument("--serial-branches", action="store_true", help="profiling aid: issue the f0 and ContentVec branches one after the other on the main stream "
                    "(test hook RVC_SERIAL_BRANCHES through rvc_debug_option; the product reads no such variable from the environment), so that a "
                    "rocprofv3 kernel trace of many streams shows every kernel's own duration")
    ap.add_argument("--preset", default="full")
    ap.add_argument("--hook", action="append", default=[], metavar="NAME=VALUE", help="measurement aid: set a test hook of the library (rvc_debug_option) before any engine exists, "
                    "e.g. --hook RVC_G32L_PANEL=0 for an A/B under rocprofv3; recorded in the line as `hooks`")
    ap.add_argument("--no-autotune", action="store_true", help="measurement aid: plans by the planner's rules only (rvc_set_plan_autotune(e, 0))")
    ap.add_argument("--verbose-line", action="store_true", help="also print the verbose record (gpurun_out/bench_full.json) to stderr")
    ap.add_argument("--no-calibration", action="store_true", help="skip rvc_calibrate and the per-leg clock probes")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------ launcher
def spawn_ranks(args, argv):
    """`bench.py --gpus N` without a torchrun environment: one child process per GPU, rank r on device r.  Every rank's stdout and
    stderr go to files (gpurun_out/bench_ranks/rank<r>.{out,err}); rank 0's stdout is replayed here (its last line is the JSON line);
    if a rank fails the run fails with that rank's last lines."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    logdir = os.environ.get("RVC_BENCH_LOGDIR") or os.path.join(ROOT, "gpurun_out", "bench_ranks")
    os.makedirs(logdir, exist_ok=True)
    procs, files = [], []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), RVC_BENCH_CHILD="1")
        fo = open(os.path.join(logdir, "rank%d.out" % r), "wb"); fe = open(os.path.join(logdir, "rank%d.err" % r), "wb")
        files += [fo, fe]
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=fo, stderr=fe))
    rcs = [p.wait() for p in procs]
    for f in files:
        f.close()
    sys.stdout.write(open(os.path.join(logdir, "rank0.out"), "rb").read().decode(errors="replace"))
    sys.stdout.flush()
    if any(rcs):
        for r, rc in enumerate(rcs):
            if rc:
                tail = open(os.path.join(logdir, "rank%d.err" % r), "rb").read().decode(errors="replace").splitlines()[-25:]
                print("bench.py: rank %d exited with %d; its last lines (%s):\n  %s" % (r, rc, os.path.join(logdir, "rank%d.err" % r), "\n  ".join(tail)), file=sys.stderr)
        raise SystemExit("bench.py: rank exit codes %s" % rcs)


# ------------------------------------------------------------------------------------------------------------ helpers
def pct(lat, q):
    return round(float(np.percentile(lat, q)) * 1e3, 4)



# ------------------------------------------------------------------------------------------------------------ the box
class BoxProbe:
    """Socket power / temperature / clock of THIS rank's GPU while a leg runs.  First choice: the amdsmi Python binding of the ROCm image (the same SMU
    metrics rocm-smi prints: current socket power, gfx clock, hotspot temperature, power cap), device matched by PCI bus id.  Fallback: sysfs hwmon of
    the card with that bus id (power1_input "PPT", freq1_input, temp2_input).  NB sysfs lists every GPU of the NODE even when the container sees one --
    card0 is not "the" GPU (tests/tools/power_series.py, box_probe.sh) -- hence the bus-id match.  Everything is optional: a missing source is a missing
    key, never an error.  The leg's CLOCK of record is the in-kernel monitor's (rvc_clock_monitor_*), these are the box's own view next to it."""

    def __init__(self, local_rank=0, bdf=None):
        self.smi = self.h = self.hwmon = self.uuid = None
        self.bdf = (bdf or "").lower()
        self.source = None
        if bdf == "none":
            return
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            pick = None
            for h in hs:
                try:
                    if self.bdf and amdsmi.amdsmi_get_gpu_device_bdf(h).lower().endswith(self.bdf[-7:]):
                        pick = h
                except Exception:
                    pass
            if pick is None and hs:
                pick = hs[local_rank] if local_rank < len(hs) else hs[0]
            if pick is not None:
                self.smi, self.h = amdsmi, pick
                try:
                    self.uuid = str(amdsmi.amdsmi_get_gpu_device_uuid(pick))
                except Exception:
                    pass
        except Exception:
            pass
        if self.smi is None:
            import glob
            for hw in sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*")):
                dev = os.path.basename(os.path.realpath(os.path.dirname(os.path.dirname(hw))))
                if self.bdf and dev.lower().endswith(self.bdf[-7:]):
                    self.hwmon = hw
        self.source = "amdsmi" if self.smi else ("sysfs " + self.hwmon if self.hwmon else None)

    @staticmethod
    def _int(path):
        try:
            with open(path) as fh:
                return int(fh.read().strip())
        except Exception:
            return None

    def sample(self):
        out = {}
        if self.smi is not None:
            A, h = self.smi, self.h
            try:
                pi = A.amdsmi_get_power_info(h)
                v = pi.get("current_socket_power")
                if not isinstance(v, (int, float)):
                    v = pi.get("average_socket_power")
                if isinstance(v, (int, float)):
                    out["power_w"] = float(v)
            except Exception:
                pass
            try:
                ci = A.amdsmi_get_clock_info(h, A.AmdSmiClkType.GFX)
                v = ci.get("clk", ci.get("cur_clk"))
                if isinstance(v, (int, float)):
                    out["sclk_mhz"] = float(v)
            except Exception:
                pass
            for tt in ("HOTSPOT", "JUNCTION", "EDGE"):
                try:
                    v = A.amdsmi_get_temp_metric(h, getattr(A.AmdSmiTemperatureType, tt), A.AmdSmiTemperatureMetric.CURRENT)
                    if isinstance(v, (int, float)):
                        out["temp_c"] = float(v)
                        break
                except Exception:
                    pass
            try:
                pc = A.amdsmi_get_power_cap_info(h).get("power_cap")
                if isinstance(pc, (int, float)) and pc > 0:
                    out["power_cap_w"] = round(pc / 1e6) if pc > 1e5 else round(pc)
            except Exception:
                pass
            return out
        if self.hwmon:
            for k, f, scale, nd in (("sclk_mhz", "freq1_input", 1e-6, 0), ("temp_c", "temp2_input", 1e-3, 1)):
                v = self._int(os.path.join(self.hwmon, f))
                if v is not None:
                    out[k] = round(v * scale, nd)
            v = self._int(os.path.join(self.hwmon, "power1_input"))
            if v is not None:
                out["power_w"] = round(v / 1e6, 1)
            cap = self._int(os.path.join(self.hwmon, "power1_cap"))
            if cap:
                out["power_cap_w"] = round(cap / 1e6)
        return out


class Sampler:
    """polls BoxProbe.sample() every `period` seconds from a thread -> mean / max of what it saw"""

    def __init__(self, probe, period=0.1):
        import threading
        self.probe, self.period, self.rows, self._stop = probe, period, [], threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            r = self.probe.sample()
            if r:
                self.rows.append(r)
            self._stop.wait(self.period)

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set(); self._t.join(timeout=2.0)

    def summary(self):
        out = {}
        for k in ("sclk_mhz", "power_w", "temp_c"):
            v = [r[k] for r in self.rows if k in r]
            if v:
                out[k] = round(float(np.mean(v)), 1)
                if k in ("power_w", "temp_c"):
                    out[k + "_max"] = round(float(np.max(v)), 1)
        if self.rows and "power_cap_w" in self.rows[0]:
            out["power_cap_w"] = self.rows[0]["power_cap_w"]
        out["samples"] = len(self.rows)
        return out


def calibrate(job, when):
    """rvc_calibrate on this rank's GPU: what the box sustains right now (bare fp32-MFMA stream + its shader clock, HBM read stream)"""
    from obs_rvc_amd import _native
    try:
        c = _native.calibrate(job.local_rank)
    except Exception as ex:
        return {"when": when, "error": str(ex)}
    return {"when": when, "mfma_f32_tflops": round(c["mfma_f32_tflops"], 2), "mfma_sclk_mhz": round(c["mfma_sclk_mhz"], 1),
            "hbm_read_tbs": round(c["hbm_read_tbs"], 3), "hbm_sclk_mhz": round(c["hbm_sclk_mhz"], 1), "ms": round(c["ms_total"], 1), "cus": c["compute_units"]}


def clock_probe(job, step, ms_per_step, box):
    """The leg's own load for >= 0.6 s with (a) the in-kernel clock monitor of the library (eight sleeping waves, one per XCD, counting shader cycles
    against the 100 MHz real-time counter: rvc_clock_monitor_*) and (b) the box sensors (amdsmi / hwmon) sampled from a thread.  Outside the timed region and the latency soak."""
    from obs_rvc_amd import _native
    n = int(min(400, max(8, np.ceil(600.0 / max(ms_per_step, 1e-3)))))
    rec = {"steps": n}
    mon = False
    try:
        _native.clock_monitor_start(job.local_rank); mon = True
    except Exception as ex:
        rec["monitor_error"] = str(ex)
    t0 = time.perf_counter()
    with Sampler(box) as sm:
        for i in range(n):
            step(i)
    rec["ms_per_step"] = round((time.perf_counter() - t0) / n * 1e3, 4)
    if mon:
        try:
            m = _native.clock_monitor_stop(job.local_rank)
            rec["sclk_mhz"] = round(m["sclk_mhz_mean"], 1); rec["sclk_mhz_min_xcd"] = round(m["sclk_mhz_min"], 1)
        except Exception as ex:
            rec["monitor_error"] = str(ex)
    rec["sensors"] = sm.summary()
    return rec


class Job:
    """Rank / world bookkeeping + the two reductions the bench needs (max of a time, concatenation of latency samples)."""

    def __init__(self, dry):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dry = dry
        if not dry:
            if not torch.cuda.is_available():
                raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
            torch.cuda.set_device(self.local_rank)
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if dry:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.local_rank))
        self.dev = "cpu" if dry else "cuda"

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        if not self.dry:
            self.torch.cuda.synchronize()

    def max_over_ranks(self, v):
        if self.world == 1:
            return float(v)
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_ok(self, ok):
        """True only if every rank reports success (one tiny all-reduce): ranks then skip a failed leg TOGETHER instead of parting ways
        in front of the next barrier."""
        if self.world == 1:
            return bool(ok)
        t = self.torch.tensor([1.0 if ok else 0.0], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    def gather(self, values):
        """every rank's list of floats (equal lengths) -> list of per-rank arrays, on every rank"""
        t = self.torch.tensor(list(values), dtype=self.torch.float64, device=self.dev)
        if self.world == 1:
            return [t.cpu().numpy()]
        gl = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(gl, t)
        return [g.cpu().numpy() for g in gl]

    def close(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()


def make_rings(S, stream_ids, n_rings, g):
    from common import chunk_stream, voice_signal
    L, chunk = g.input_buffer_16k_size, g.sample_frame_16k
    rings = np.zeros((n_rings, S, L), np.float32)
    for s in range(S):
        audio = voice_signal(chunk * (n_rings + 14), seed=stream_ids[s])
        rs = list(chunk_stream(audio, L, chunk))[-n_rings:]
        for i in range(n_rings):
            rings[i, s] = rs[i]
    return rings


def timed_leg(job, eng, d_rings, d_out, g, steps, warmup):
    """W warm-up steps, then exactly K steps bracketed by barrier + synchronize on both sides; every step synchronised so that the
    per-chunk latency is observed.  -> (elapsed seconds = max over ranks, this rank's per-step latencies)"""
    L, chunk, N = g.input_buffer_16k_size, g.sample_frame_16k, g.model_return_size
    n_rings = d_rings.shape[0]

    def step(i):
        eng.infer_device(d_rings[i % n_rings].data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, d_out.data_ptr(), N, sync=True)

    for i in range(warmup):
        step(i)
    # The interpreter's cyclic collector is the measuring host's business, not the path's: with torch imported a full collection walks ~10^6 objects --
    # 10 ms pauses that landed inside 20 timed steps of 2 ms twice in this round's driver-style runs (p99 12 ms on a leg whose p50 was 2.04).  Everything alive
    # now is frozen out of the collector's reach and the collector stays off while the timed region and the latency soak behind it run (run_config turns it
    # back on).
    gc.collect(); gc.freeze(); gc.disable()
    job.barrier()
    lat = []
    t0 = time.perf_counter()
    for i in range(steps):
        t1 = time.perf_counter()
        step(i)
        lat.append(time.perf_counter() - t1)
    job.torch.cuda.synchronize()
    job.barrier()
    elapsed = job.max_over_ranks(time.perf_counter() - t0)
    return elapsed, lat, step


def committed_traffic(S, with_index=False, version=2, preset="full"):
    """HBM traffic of the dominant kernel class from the committed rocprofv3 --pmc pass of THIS build AND THIS configuration
    (tests/tools/profile_round.sh -> profiles/<round>_pmc_traffic*.json; PMC counters need their own rocprofv3 run, so they cannot be taken
    inside this process).  A file whose build hash differs from the loaded library's, or whose recorded configuration (streams, index,
    model version, preset) is not the running one, is refused: -> (bytes per launch | None, bytes per launch of the retrieval scan | None, note)"""
    from obs_rvc_amd import _native
    have = _native.binary_hash()
    import glob
    want_cfg = {"streams": int(S), "index": bool(with_index), "version": int(version), "preset": preset}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True)
    cands = []
    for f in files:
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("config") == want_cfg:
            cands.append((f, d))
    if not cands:
        return None, None, {"traffic_source": None, "traffic_note": "no committed PMC pass of this configuration (%s) under profiles/" % json.dumps(want_cfg, sort_keys=True)}
    match = [c for c in cands if c[1].get("build") == have]
    f, d = (match or cands)[0]
    src = {"traffic_source": os.path.relpath(f, ROOT) + " (rocprofv3 --pmc FETCH_SIZE, x2 on gfx950, separate pass of the same configuration)",
           "traffic_config": d.get("config"), "traffic_build": d.get("build"), "library_build": have}
    if d.get("build") != have:
        src["traffic_note"] = "REFUSED: the committed pass was taken on build %s, this library is %s -- rerun tests/tools/profile_round.sh" % (d.get("build"), have)
        return None, None, src
    ig = d.get("igemm_all_instantiations") or {}
    kd = d.get("knn_scan_select_kernel") or d.get("knn_dot_kernel") or {}
    src["traffic_launches_per_step"] = ig.get("launches_per_chunk")
    src["traffic_bytes_per_step"] = ig.get("hbm_read_bytes_per_chunk")
    src["algorithmic_weight_bytes_per_step"] = ig.get("algorithmic_weight_bytes_per_chunk")
    src["algorithmic_bytes_per_step_estimate"] = ig.get("algorithmic_bytes_per_chunk_estimate")
    return ig.get("hbm_read_bytes_per_launch"), kd.get("hbm_read_bytes_per_launch"), src


def committed_serial_pass(S, sum_kernel_ms):
    """Validation of the HIP-event figure, not a gate: profiles/<round>_serial_<S>streams.json records, for a serial-branch pass on the builder's box,
    the rocprofv3 kernel-trace sum of the implicit-GEMM class per step (tests/tools/serial_pass.py).  This run's `frac` is its own HIP-event figure; the
    event sum of this (unprofiled) run over that rocprofv3 sum is quoted next to it: 1.00 on the same build and box, the box's speed ratio elsewhere.  -> record"""
    from obs_rvc_amd import _native
    import glob
    have = _native.binary_hash()
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_serial_%dstreams.json" % S)), reverse=True)
    if not files:
        return {"serial_pass": None}
    f = files[0]
    try:
        d = json.load(open(f))
    except Exception as ex:
        return {"serial_pass": os.path.relpath(f, ROOT), "serial_pass_note": "unreadable: %s" % ex}
    rec = {"serial_pass": os.path.relpath(f, ROOT), "serial_pass_csv": d.get("csv"), "serial_pass_build": d.get("build"), "library_build": have,
           "serial_pass_same_build": d.get("build") == have,
           "serial_pass_rocprof_ms_per_step": d.get("sum_igemm_ms_per_step"), "serial_pass_events_ms_per_step": d.get("events_sum_igemm_ms_per_step")}
    # (the pass's own event sum was taken while rocprofv3 traced the process -- the tool adds ~4 us to every event pair -- so it is kept for the record only;
    #  the validation of the event method is THIS unprofiled run's event sum over the pass's rocprofv3 sum, below: 1.00 on the same build and box)
    if d.get("sum_igemm_ms_per_step"):
        rec["this_run_events_over_committed_rocprof"] = round(sum_kernel_ms / float(d["sum_igemm_ms_per_step"]), 4)
    return rec


KERNEL_CLASS = "implicit-GEMM class: rvc::igemm2 / igemm2w / conv_tile / igemm32 / igemm32l / conv32s(_buf) / igemm_lds / rm_block kernels, all instantiations"


def roofline_of(eng, step, S, reps=5, with_index=False, version=2, preset="full"):
    """Dominant kernel class (implicit GEMM on the fp32 matrix cores): per-launch HIP events on the stream each kernel is launched on
    (hipExtLaunchKernelGGL start/stop = the dispatch's own begin/end), eager launches of the same kernels and shapes.  Above 4 streams the two
    front branches are issued one after the other while the events are taken (they share the CUs there, and a co-scheduled short kernel's event
    duration would be the long kernel's)."""
    eng.set_profile(True)
    tot_ms = tot_fl = k_ms = k_by = 0.0
    n_l = k_n = 0
    for i in range(reps):
        step(i)
        nl, ms, fl = eng.profile_last()
        tot_ms += ms; tot_fl += fl; n_l += nl
        kn, kms, kby = eng.profile_last_knn()
        k_n += kn; k_ms += kms; k_by += kby
    eng.set_profile(False)
    achieved = tot_fl / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
    # HBM traffic per launch: the committed `rocprofv3 --pmc` pass of this build (committed_traffic: refused when the build hashes differ)
    t_launch, t_knn, t_src = committed_traffic(S, with_index, version, preset)
    roof = {"bound": "mfma", "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 5), "traffic": t_launch, "traffic_unit": "HBM read bytes per launch (class average)",
            "kernel": KERNEL_CLASS, "launches_per_step": n_l // reps,
            "avg_launch_us": round(tot_ms * 1e3 / max(n_l, 1), 3), "flops_per_step": tot_fl / reps,
            "sum_kernel_ms": round(tot_ms / reps, 4),
            "events": "in-run HIP events of every launch" + ("" if S <= 4 else ", front branches issued serially"),
            "note": ("achieved = flops / SUM of per-launch durations; the ContentVec and f0 branches overlap on disjoint CU sets, so the sum "
                     "exceeds the step's wall time -- frac_by_wall uses the step's wall clock") if S <= 4 else
                    ("achieved = flops / SUM of per-launch durations with the two front branches issued one after the other; the timed steps run "
                     "them concurrently -- frac_by_wall uses their wall clock")}
    roof.update(t_src)
    if S > 4:
        roof.update(committed_serial_pass(S, roof["sum_kernel_ms"]))
    if k_n:
        ach = k_by / (k_ms * 1e-3) / 1e9
        roof["retrieval_scan"] = {"bound": "hbm", "kernel": "rvc::knn_scan_select_kernel (one launch: scan + select + exact re-rank + blend)", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(ach / HBM_PEAK_GBS, 4), "bytes_per_launch": k_by / k_n, "avg_launch_us": round(k_ms * 1e3 / k_n, 2),
                                  "traffic": t_knn}
    return roof


def finish_roofline(roof, ms_per_step, peaks, probe):
    """fractions against what THIS box measured: the calibration's bare-MFMA figure, and the nominal peak scaled to the shader clock the leg itself ran at"""
    roof["frac_by_wall"] = round(roof["flops_per_step"] / (ms_per_step * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 5)
    pm = (peaks or {}).get("mfma_f32_tflops")
    if pm:
        roof["peak_measured"] = pm
        roof["frac_vs_measured_peak"] = round(roof["achieved"] / pm, 5)
    clk = (probe or {}).get("sclk_mhz")
    if clk:
        roof["leg_sclk_mhz"] = clk
        roof["frac_at_leg_clock"] = round(roof["achieved"] / (FP32_MFMA_PEAK_TFLOPS * clk / 2400.0), 5)
    hb = (peaks or {}).get("hbm_read_tbs")
    if hb and roof.get("retrieval_scan"):
        roof["retrieval_scan"]["frac_vs_measured_peak"] = round(roof["retrieval_scan"]["achieved"] / (hb * 1e3), 4)
    return roof


def run_config(job, z, g, S, with_index, steps, warmup, graph, index_vecs, want_roofline=True, version=2, soak=0, preset="full", gemm_precision=0):
    """One configuration on every rank: S streams per GPU, retrieval on/off.  soak > 0: that many more synchronised chunks after the
    timed region for the latency distribution (p99.9 needs >= 1000 samples).  -> record (rank 0) / None"""
    from obs_rvc_amd import dist as rdist
    from obs_rvc_amd.rvc import RvcInfer
    torch = job.torch
    eng = RvcInfer(z["data"], device=job.local_rank)
    eng.load_contentvec(version); eng.load_f0(1); eng.load_model(z["model"])
    eng.set_streams(S)
    if not CTX["autotune"]:
        eng.set_plan_autotune(False)
    if gemm_precision:
        eng.set_gemm_precision(gemm_precision)
    # stream s of this rank is stream (s * world + rank) of the job: round-robin sharding
    stream_ids = [s * job.world + job.rank for s in range(S)]
    eng.set_noise_seed(1234, job.rank * S)
    bcast_ms = None
    bcast_info = None
    if with_index:
        b0 = time.perf_counter()
        err = None
        try:
            rdist.load_shared_index(eng, index_vecs if job.rank == 0 else None, 100000, 768, job.rank, job.world)
        except Exception as ex:                   # e.g. librccl missing / communicator bootstrap refused on this node
            err = str(ex)
        if not job.all_ok(err is None):
            raise RuntimeError("index broadcast failed on at least one rank" + (": " + err if err else ""))
        bcast_ms = round((time.perf_counter() - b0) * 1e3, 2)
        bcast_info = eng.index_broadcast_info() if job.world > 1 else None
        eng.set_index_rate(0.75)
    eng.set_use_graph(graph)
    n_rings = 8 if S <= 8 else 4
    rings = make_rings(S, stream_ids, n_rings, g)
    d_rings = torch.from_numpy(rings).cuda()
    d_out = torch.empty((S, g.model_return_size), dtype=torch.float32, device="cuda")
    elapsed, lat, step = timed_leg(job, eng, d_rings, d_out, g, steps, warmup)
    gpu_ms_last = eng.last_gpu_ms()            # device time of the last timed chunk (events around the call's launches): next to the wall clock, it
    soak_lat = []                              # tells a slow GPU from a slow submitting thread
    for i in range(soak):                        # latency distribution: more synchronised chunks behind the timed region
        t1 = time.perf_counter(); step(i); soak_lat.append(time.perf_counter() - t1)
    gc.enable()
    lat_all = job.gather(lat)
    soak_all = job.gather(soak_lat) if soak else []
    per_rank = [FRAMES_PER_CHUNK * steps * S / float(np.sum(l)) for l in lat_all]
    rec = None
    # what the box did WHILE this leg's load ran (outside the timed region and the soak): effective shader clock, socket power, temperature
    probe = clock_probe(job, step, elapsed / steps * 1e3, CTX["box"]) if (CTX["calib"] and CTX["box"] is not None) else None
    roof = roofline_of(eng, step, S, with_index=with_index, version=version, preset=preset) if (job.rank == 0 and want_roofline) else None
    if job.rank == 0:
        allat = np.concatenate(lat_all + soak_all)
        value = FRAMES_PER_CHUNK * steps * S * job.world / elapsed
        ms = elapsed / steps * 1e3
        rec = {"frames_per_s": round(value, 2), "ms_per_step": round(ms, 4), "streams_per_gpu": S, "streams_total": S * job.world, "n_gpus": job.world,
               "retrieval": "100k x768 flat-L2 k=4 rate 0.75" if with_index else "off",
               "latency_ms": {"p50": pct(allat, 50), "p99": pct(allat, 99), "p99.9": pct(allat, 99.9), "max": round(float(allat.max()) * 1e3, 4),
                              "samples": int(allat.size)},
               "gpu_ms_last_chunk": round(float(gpu_ms_last), 4),
               "rtf": round(float(np.percentile(allat, 99)) / 0.160, 5),
               "per_rank_frames_per_s": [round(v, 1) for v in per_rank]}
        if with_index:
            if bcast_info:      # what the communicator itself reports (ncclCommCount) and where the load step's time went
                rec["index_broadcast"] = {"via": "rvc_index_broadcast (ncclBroadcast, librccl over xGMI)", "rccl_ranks": bcast_info["ranks"], "bytes": 100000 * 768 * 4,
                                          "ms_comm_init": bcast_info["ms_comm_init"], "ms_broadcast": bcast_info["ms_broadcast"],
                                          "ms_device_repack": bcast_info["ms_repack"], "ms_total_host": bcast_ms}
            else:
                rec["index_broadcast"] = {"via": "rvc_load_index (one rank: plain upload, no communicator)", "rccl_ranks": 0, "bytes": 100000 * 768 * 4,
                                          "ms_total_host": bcast_ms}
        if probe:
            rec["box_under_load"] = probe
        if roof:
            rec["roofline"] = finish_roofline(roof, ms, CTX["peaks"], probe)
    return rec, eng, rings, d_rings


CTX = {"peaks": None, "box": None, "calib": True, "autotune": True}


def compact_roofline(r):
    """the roofline record as it goes into the one-line JSON: numbers and short strings only"""
    if not r:
        return None
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "launches_per_step", "avg_launch_us", "sum_kernel_ms", "flops_per_step", "frac_by_wall",
            "peak_measured", "frac_vs_measured_peak", "leg_sclk_mhz", "frac_at_leg_clock", "traffic_bytes_per_step", "algorithmic_weight_bytes_per_step",
            "algorithmic_bytes_per_step_estimate", "this_run_events_over_committed_rocprof", "serial_pass_same_build")
    out = {k: r[k] for k in keep if r.get(k) is not None or k in ("frac", "traffic")}
    out["kernel"] = "implicit-GEMM class (igemm2/2w/32/32l, conv_tile, conv32s), all launches"
    if r.get("traffic_source"):
        out["traffic_source"] = r["traffic_source"].split(" ")[0]
    if r.get("retrieval_scan"):
        q = r["retrieval_scan"]
        out["retrieval_scan"] = {k: q[k] for k in ("bound", "achieved", "peak", "unit", "frac", "frac_vs_measured_peak", "avg_launch_us", "bytes_per_launch", "traffic") if q.get(k) is not None}
    return out


def compact_sub(rec):
    """one sub-configuration as a short record: ms per step, frames/s, latency, GPU ms, the box while it ran, roofline fractions"""
    if not rec or "error" in rec:
        return rec
    o = {"ms": rec["ms_per_step"], "fps": round(rec["frames_per_s"]), "p50": rec["latency_ms"]["p50"], "p99": rec["latency_ms"]["p99"], "gpu_ms": rec["gpu_ms_last_chunk"]}
    b = rec.get("box_under_load") or {}
    if b.get("sclk_mhz"):
        o["sclk"] = round(b["sclk_mhz"])
    sysfs = b.get("sensors") or {}
    if sysfs.get("power_w") is not None:
        o["W"] = round(sysfs["power_w"])
    if sysfs.get("temp_c_max") is not None:
        o["C"] = round(sysfs["temp_c_max"])
    if sysfs.get("power_cap_w") is not None and sysfs.get("power_w_max") is not None:
        o["W_max"] = round(sysfs["power_w_max"])
    r = rec.get("roofline") or {}
    for k_in, k_out in (("frac", "frac"), ("frac_by_wall", "frac_wall"), ("frac_vs_measured_peak", "frac_meas"), ("frac_at_leg_clock", "frac_clk")):
        if r.get(k_in) is not None:
            o[k_out] = round(r[k_in], 4)
    if (r.get("retrieval_scan") or {}).get("frac") is not None:
        o["scan_frac_hbm"] = r["retrieval_scan"]["frac"]; o["scan_us"] = r["retrieval_scan"]["avg_launch_us"]
    if rec.get("dtype"):
        o["dtype"] = "bf16x3 (exploratory)"
    return o


def host_buffer_leg(eng, rings, g, steps):
    """The reference's boundary hands over host buffers (`RvcInfer::infer(ArrayView1<f32>) -> Array1<f32>`, rvc.rs:133-220): the same
    chunk through the host-pointer C ABI -- H2D 143 KB + kernels + D2H 40 KB + one synchronisation inside the call (SURVEY.md 8d:
    latency from request bytes available to reply bytes complete)."""
    ts = []
    n = min(max(steps, 30), 200)
    for i in range(n + 5):
        h0 = time.perf_counter()
        eng.infer(rings[i % rings.shape[0], 0], g.sample_frame_16k, 12, g.skip_head, g.model_return_length)
        ts.append(time.perf_counter() - h0)
    ts = np.array(ts[5:])
    return {"p50": pct(ts, 50), "p99": pct(ts, 99), "max": round(float(ts.max()) * 1e3, 4), "samples": int(n),
            "frames_per_s": round(FRAMES_PER_CHUNK * n / float(ts.sum()), 2), "api": "rvc_infer (host pointers: H2D + D2H + sync inside the call)"}


def pipelined_leg(job, eng, d_rings, g, S, steps):
    """Offline throughput mode (rvc_set_pipeline): K unsynchronised chunks, consecutive chunks overlap on the GPU."""
    torch = job.torch
    L, chunk, N = g.input_buffer_16k_size, g.sample_frame_16k, g.model_return_size
    n_rings = d_rings.shape[0]
    outs = torch.empty((n_rings, S, N), dtype=torch.float32, device="cuda")
    eng.set_pipeline(True)
    try:
        for i in range(6):
            eng.infer_device(d_rings[i % n_rings].data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, outs[i % n_rings].data_ptr(), N, sync=False)
        eng.synchronize()
        p0 = time.perf_counter()
        for i in range(steps):
            eng.infer_device(d_rings[i % n_rings].data_ptr(), L, chunk, 12, g.skip_head, g.model_return_length, outs[i % n_rings].data_ptr(), N, sync=False)
        eng.synchronize()
        return round(FRAMES_PER_CHUNK * steps * S / (time.perf_counter() - p0), 2)
    finally:
        eng.set_pipeline(False)


def chain_leg(eng, S, g):
    """The whole plugin-side chain as one native call (rvc_session_process: 48 kHz chunk in -> resample -> infer -> resample ->
    envelope -> SOLA -> 48 kHz frame out, rings resident in HBM).  A chunk on which the reference would panic (RVC_PANIC: the
    synthetic RMVPE weights can hit rmvpe.rs:124 while the 2.24 s ring is still mostly zeros) is COUNTED, not hidden."""
    from common import voice_signal
    from obs_rvc_amd.rvc_common import RvcInferError
    from obs_rvc_amd.streaming import NativeStreamingSession
    chunk = g.sample_frame_16k
    ses = NativeStreamingSession(eng, 48000, 0.16, 0.07, 2.0, 48000, 12, 0.75)
    F, n_ch = ses.sample_frame_size, 34 if S == 1 else 12
    x48 = np.stack([np.interp(np.arange(F * n_ch) / 48000.0, np.arange(chunk * n_ch) / 16000.0, voice_signal(chunk * n_ch, seed=99 + s)).astype(np.float32)
                    for s in range(S)])
    ts, panics, ok = [], 0, []
    for i in range(n_ch):
        xin = x48[:, i * F:(i + 1) * F] if S > 1 else x48[0, i * F:(i + 1) * F]
        h0 = time.perf_counter()
        good = True
        try:
            ses.process_one_frame(np.ascontiguousarray(xin))
        except RvcInferError as ex:
            if "Panic" not in str(ex):
                raise
            panics += 1; good = False          # counted, and kept out of the median
        ts.append(time.perf_counter() - h0); ok.append(good)
    del ses
    kept = [t for t, gd in list(zip(ts, ok))[n_ch // 5:] if gd]
    return {"ms_per_chunk": round(float(np.median(kept)) * 1e3, 4) if kept else None, "chunks": n_ch, "chunks_in_median": len(kept), "panic_chunks": panics}


def cpu_baseline_leg(z, rings, g, with_index, index_vecs):
    """CPU baseline: the C oracle (a port of the reference's path; the reference itself -- Rust + ONNX Runtime -- cannot run here)
    on this node's host cores, bounded sample of the same workload."""
    from oracle import oracle as O
    # 16 OpenMP threads is this restatement's optimum on the 256-CPU host (1: 1158, 8: 358, 16: 297, 32: 550, 64: 1100 ms/chunk --
    # its parallel loops are per layer and short, so more threads only add barrier cost); the single-thread figure is reported too
    threads = min(os.cpu_count() or 1, 16)
    chunk = g.sample_frame_16k
    ora = O.OracleRvcInfer(z["data"])
    ora.load_contentvec(2); ora.load_f0(1); ora.load_model(z["model"]); ora.set_noise_seed(1234, 0)
    if with_index:
        ora.load_index(index_vecs); ora.set_index_rate(0.75)
    O.set_threads(1)
    ora.infer(rings[0, 0], chunk, 12, g.skip_head, g.model_return_length)
    c0 = time.perf_counter()
    for i in range(3):
        ora.infer(rings[i % rings.shape[0], 0], chunk, 12, g.skip_head, g.model_return_length)
    one_thread_ms = (time.perf_counter() - c0) / 3 * 1e3
    O.set_threads(threads)
    n_cpu = 40
    ora.infer(rings[0, 0], chunk, 12, g.skip_head, g.model_return_length)
    c0 = time.perf_counter()
    for i in range(n_cpu):
        ora.infer(rings[i % rings.shape[0], 0], chunk, 12, g.skip_head, g.model_return_length)
    ct = time.perf_counter() - c0
    return {"value": round(FRAMES_PER_CHUNK * n_cpu / ct, 2), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "%d chunks of stream 0 (same rings, same weights), C oracle with OpenMP, %.1f s" % (n_cpu, ct),
            "ms_per_chunk": round(ct / n_cpu * 1e3, 2), "one_thread_ms_per_chunk": round(one_thread_ms, 1), "host_cpus": os.cpu_count()}


# ------------------------------------------------------------------------------------------------------------ dry launch
def dry_launch(args):
    """No GPU: the launcher, the gloo rendezvous, the stream sharding and the reductions, with a stand-in step."""
    from obs_rvc_amd import dist as rdist
    job = Job(dry=True)
    S = args.streams
    total = S * job.world
    mine = rdist.local_streams(total, job.rank, job.world)
    assert mine == [s * job.world + job.rank for s in range(S)]
    lat = []
    job.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t1 = time.perf_counter()
        time.sleep(0.001)
        lat.append(time.perf_counter() - t1)
    job.barrier()
    elapsed = job.max_over_ranks(time.perf_counter() - t0)
    lat_all = job.gather(lat)
    owned = job.gather([float(len(mine))] + [float(s) for s in mine])
    if job.rank == 0:
        streams = sorted(int(v) for o in owned for v in o[1:])
        print(json.dumps({"metric": "dry launch (no GPU work)", "dry_launch": True, "n_gpus": job.world, "steps": args.steps, "warmup": args.warmup,
                          "streams_total": total, "streams_covered": streams == list(range(total)),
                          "per_rank_streams": [[int(v) for v in o[1:]] for o in owned],
                          "ms_per_step": round(elapsed / args.steps * 1e3, 4), "latency_samples": int(sum(len(l) for l in lat_all)),
                          # the real run's shape with more than one rank: BASELINE configs[4] as a top-level key
                          "config4": ({"dry_launch": True, "streams_per_gpu": S, "streams_total": total, "n_gpus": job.world,
                                       "per_rank_streams": [len(o) - 1 for o in owned]} if job.world > 1 else None)}))
    job.close()
    if os.environ.get("RVC_BENCH_FAIL_RANK") == str(job.rank):        # test aid: a rank that dies after the collectives
        print("bench: simulated failure of rank %d (RVC_BENCH_FAIL_RANK)" % job.rank, file=sys.stderr)
        sys.exit(3)


# ------------------------------------------------------------------------------------------------------------ main
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args, argv)
    if args.dry_launch:
        return dry_launch(args)
    job = Job(dry=False)
    if job.world != args.gpus and job.rank == 0:
        print("bench: --gpus %d but WORLD_SIZE=%d: running the %d ranks that exist" % (args.gpus, job.world, job.world), file=sys.stderr)
    from common import BASELINE_160MS as g, zoo
    from obs_rvc_amd import weights as W
    if job.rank == 0:                      # one rank writes the model zoo, the others read it
        z = zoo(args.preset)
        if not args.only_headline and args.preset == "full" and args.streams == 1 and not args.index:
            zoo(args.preset, 1)
    job.barrier()
    z = zoo(args.preset)
    graph = bool(args.graph and not args.no_graph)
    full = args.preset == "full"
    if args.serial_branches:
        from common import set_opt
        set_opt("RVC_SERIAL_BRANCHES", "1")
    for hk in args.hook:
        from common import set_opt
        set_opt(*hk.split("=", 1))
    CTX["autotune"] = not args.no_autotune
    index_vecs = W.make_index() if (job.rank == 0 and full) else None      # only rank 0 ever holds the host copy

    # ---- the box: sensor probe + in-run calibration (what this GPU's matrix cores and HBM sustain right now; every rank calibrates its own GPU)
    # (under rocprofv3 the in-kernel clock monitor is left out: counter collection serialises dispatches, and a monitor that waits for a host flag would
    #  hold every later kernel back until its own time-out)
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ)
    CTX["calib"] = not args.no_calibration and not under_profiler
    bdf = None
    try:
        pr = job.torch.cuda.get_device_properties(job.local_rank)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        pass
    CTX["box"] = BoxProbe(job.local_rank, bdf if CTX["calib"] else "none")      # (--no-calibration: no sensor library is loaded at all)
    box_idle = CTX["box"].sample()
    calib = []
    if CTX["calib"]:
        calib.append(calibrate(job, "start"))
        CTX["peaks"] = calib[0] if "error" not in calib[0] else None
    job.barrier()

    # ---- headline: BASELINE configs[1] (or what --streams / --index ask for) on every rank
    S = args.streams
    soak_n = int(os.environ.get("RVC_BENCH_SOAK", "1000")) if (S == 1 and full and not args.only_headline) else 0
    head, eng, rings, d_rings = run_config(job, z, g, S, args.index and full, args.steps, args.warmup, graph, index_vecs, soak=soak_n, preset=args.preset)
    extra = {}
    if job.rank == 0 and not args.only_headline:
        skip = os.environ.get("RVC_BENCH_SKIP", "").split(",")     # debugging aid: leg names to leave out

        def informational(name, fn):                       # informational legs never cost the headline line
            if name in skip:
                return
            try:
                extra[name] = fn()
            except Exception as ex:
                print("bench: %s leg failed: %s" % (name, ex), file=sys.stderr)
                extra[name] = None
        if S == 1:
            informational("latency_ms_host_buffer", lambda: host_buffer_leg(eng, rings, g, args.steps))
        if S <= 4 and full and not graph:
            informational("offline_pipelined_frames_per_s", lambda: pipelined_leg(job, eng, d_rings, g, S, args.steps))
        if full and not args.index:
            informational("plugin_chain", lambda: chain_leg(eng, S, g))
    del eng, d_rings
    job.torch.cuda.empty_cache()

    # ---- sub-configurations: every rank takes part (the timed regions are bracketed by barriers)
    sub = {}
    if not args.only_headline and full and S == 1 and not args.index:
        # Each leg raises only where every rank raises together (index broadcast: agreed through Job.all_ok; anything else is deterministic
        # per configuration), so skipping a failed leg cannot strand a rank in a collective.
        # latency distributions of the two sub-configurations VERDICT r3 asked for come from >= 200 synchronised chunks (p99 of 20 samples
        # is their maximum); the 64-stream plans get 8 warm-up steps (3 left a first-use outlier inside the driver's 20 timed steps)
        sub_soak = int(os.environ.get("RVC_BENCH_SUB_SOAK", "200"))

        def leg(name, fn):
            if args.legs and name not in args.legs.split(","):
                return
            try:
                sub[name] = fn()
            except Exception as ex:
                print("bench: sub-configuration %s failed: %s" % (name, ex), file=sys.stderr)
                sub[name] = {"error": str(ex)}
            job.torch.cuda.empty_cache()

        def index100k():
            rec, e2, _, d2 = run_config(job, z, g, 1, True, args.steps, args.warmup, graph, index_vecs, soak=sub_soak)
            del e2, d2
            return rec

        def streams64():
            k64 = max(10, min(args.steps, 20))
            want_index = job.world > 1 and "error" not in (sub.get("index100k") or {})      # configs[4] shares the index over RCCL
            rec, e3, _, d3 = run_config(job, z, g, 64, want_index, k64, 8, graph, index_vecs, soak=sub_soak)
            del e3, d3
            if rec:
                rec["steps"] = k64
                rec["config"] = "BASELINE configs[%d]" % (4 if job.world > 1 else 3)
            return rec
        def sweep(S2):
            def fn():
                k = max(10, min(args.steps, 30))
                rec, e4, _, d4 = run_config(job, z, g, S2, False, k, 3, graph, index_vecs, soak=sub_soak)
                del e4, d4
                if rec:
                    rec["steps"] = k
                return rec
            return fn

        def streams64_index100k():
            k64 = max(10, min(args.steps, 20))
            rec, e5, _, d5 = run_config(job, z, g, 64, True, k64, 3, graph, index_vecs, want_roofline=False, soak=sub_soak)
            del e5, d5
            if rec:
                rec["steps"] = k64
                rec["config"] = "one rank of BASELINE configs[4] on one GPU: 64 streams + the 100k x 768 index"
            return rec

        def v1_256():
            z1 = zoo(args.preset, 1)
            rec, e6, _, d6 = run_config(job, z1, g, 1, False, args.steps, args.warmup, graph, index_vecs, version=1, soak=sub_soak)
            del e6, d6
            if rec:
                rec["config"] = "BASELINE configs[1] read literally: ContentVec-256 (v1: layer 9 + final_proj, enums.rs:10-23) + RMVPE + v1 NSF-HiFiGAN 48k"
            return rec
        def streams64_bf16x3():
            # EXPLORATORY sub-configuration, never `value`: the 1-D layers with >= 128 output rows as three bf16 matrix-core products per fp32 product
            # (rvc_set_gemm_precision(e, 1)); everything else, and every other record of this line, computes in f32 like the reference
            k64 = max(10, min(args.steps, 20))
            rec, e7, _, d7 = run_config(job, z, g, 64, False, k64, 8, graph, index_vecs, want_roofline=False, soak=sub_soak, gemm_precision=1)
            del e7, d7
            if rec:
                rec["steps"] = k64
                rec["dtype"] = "bf16x3 in the 1-D layers with >= 128 output rows (a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_bf16, f32 accumulate: ~2^-16 relative per product); f32 everywhere else"
                rec["exploratory"] = "not the reference's arithmetic: reported next to streams64 (f32), never as value; parity of the mode: tests/test_gpu_round5.py::test_split_bf16_gemms_exploratory_mode"
            return rec
        leg("index100k", index100k)
        leg("streams64", streams64)
        if job.world == 1:
            leg("streams64_index100k", streams64_index100k)
        for S2 in (2, 4, 8, 16, 32):
            leg("streams%d" % S2, sweep(S2))
        leg("v1_256", v1_256)
        if job.world == 1:
            leg("streams64_bf16x3", streams64_bf16x3)

    # the CPU baseline runs LAST (round 6): behind it the process keeps the oracle's OpenMP team, and in four driver-style runs the latency-bound GPU legs issued
    # after it were up to 9 % slower (the v1 leg 2.12-2.19 ms; 2.05 with the baseline behind it in the same kind of run, 1.94-1.96 in runs without it) -- the baseline
    # is a separate measurement and must not sit in front of anything it can disturb
    if CTX["calib"]:
        calib.append(calibrate(job, "end"))
    cpu = None
    if job.rank == 0 and job.world == 1 and not args.no_cpu:
        cpu = cpu_baseline_leg(z, rings, g, args.index and full, index_vecs)
    if job.rank == 0:
        if args.serial_branches:
            head["serial_branches"] = True
        workload = ("BASELINE configs[%d]: %d stream(s)/GPU, ContentVec v2-768 + RMVPE + NSF-HiFiGAN v2-48k, retrieval %s, preset %s"
                    % (2 if args.index else (1 if S == 1 else 3), S, "100k x768 flat-L2 k=4" if args.index else "off", args.preset))
        rccl_ranks = (((sub.get("streams64") or {}).get("index_broadcast") or (sub.get("index100k") or {}).get("index_broadcast") or head.get("index_broadcast") or {}).get("rccl_ranks", 0))
        gpu_name = None
        try:
            gpu_name = job.torch.cuda.get_device_name(job.local_rank)
        except Exception:
            pass
        box = {"gpu": gpu_name, "pci": bdf, "uuid": CTX["box"].uuid, "sensor_source": CTX["box"].source, "host_cpus": os.cpu_count(), "idle": box_idle, "headline_under_load": head.get("box_under_load")}
        # ---- the verbose record: everything, with notes and sources (file + optionally stderr)
        full_rec = {
            "metric": "audio frames/sec (10 ms hops of new input, 160 ms chunks @16 kHz)", "value": head["frames_per_s"], "unit": "frames/s",
            "n_gpus": job.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "gpu_ms_last_chunk": head.get("gpu_ms_last_chunk"),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "streams_per_gpu": S, "streams_total": S * job.world, "chunk_ms": 160, "input_samples_16k": g.input_buffer_16k_size,
                       "output_samples": g.model_return_size, "hip_graph": graph,
                       "timed_api": "rvc_infer_device: inputs resident in HBM when the timed region starts, one synchronisation per chunk; the host-buffer boundary (H2D + D2H inside the call) is latency_ms_host_buffer"},
            "latency_ms": head["latency_ms"], "rtf": head["rtf"], "rccl_ranks": rccl_ranks, "per_rank_frames_per_s": head["per_rank_frames_per_s"],
            "value_contract": "value / ms_per_step / latency_ms: rvc_infer_device, inputs resident in HBM when the timed region starts (the bench contract); the reference's host-buffer boundary (SURVEY 8d: request bytes available -> reply bytes complete, H2D + D2H inside the call) is latency_ms_host_buffer, measured in the same run",
            "latency_ms_host_buffer": extra.get("latency_ms_host_buffer"), "plugin_chain": extra.get("plugin_chain"),
            "offline_pipelined_frames_per_s": extra.get("offline_pipelined_frames_per_s"),
            "roofline": head.get("roofline"), "cpu_baseline": cpu, "peak_measured": calib, "box": box, "sub_configs": sub or None,
            "serial_branches": bool(args.serial_branches),
            "notes": {
                "frac": "roofline.frac = algorithmic flops of the implicit-GEMM class / SUM of its launches' own HIP-event durations of THIS run (above 4 streams with the two front branches issued serially), against the nominal 157.3 TF/s fp32 matrix-core peak",
                "frac_vs_measured_peak": "the same achieved figure against peak_measured[0].mfma_f32_tflops: a bare v_mfma_f32_32x32x2_f32 stream timed on this GPU at the start of this run (rvc_calibrate)",
                "frac_at_leg_clock": "against 157.3 TF/s x (effective shader clock while the leg ran / 2400 MHz); the clock is counted by sleeping waves inside the GPU (s_memtime cycles per s_memrealtime tick, rvc_clock_monitor_*) during >= 0.6 s of the leg's own load, outside the timed region",
                "box": "sclk / W / C per leg: shader clock from that monitor; socket power (W, against power_cap_w) and hotspot temperature from the SMU (amdsmi, device matched by PCI bus id) sampled every 100 ms during the same probe steps",
                "serial_pass": "profiles/<round>_serial_<S>streams.json: rocprofv3 kernel-trace sum of the implicit-GEMM class per step of one serial-branch process on the builder's box (whole steps behind the plan build's autotune trials: tests/tools/trace_stats.py); this_run_events_over_committed_rocprof = this unprofiled run's HIP-event sum over it (1.00 on the same build and box)",
            },
        }
        if args.index and head.get("index_broadcast"):
            full_rec["index_broadcast"] = head["index_broadcast"]
        if job.world > 1 and sub.get("streams64"):
            full_rec["config4"] = sub["streams64"]        # BASELINE configs[4]: 64 streams per GPU x N GPUs, index broadcast over RCCL
        full_path = None
        try:
            logdir = os.path.join(ROOT, "gpurun_out")
            os.makedirs(logdir, exist_ok=True)
            full_path = os.path.join(logdir, "bench_full.json")
            with open(full_path, "w") as fh:
                json.dump(full_rec, fh, indent=1)
            full_path = os.path.relpath(full_path, ROOT)
        except Exception as ex:
            print("bench: could not write the verbose record: %s" % ex, file=sys.stderr)
        if args.verbose_line:
            print(json.dumps(full_rec), file=sys.stderr)
        # ---- the line: compact, `summary_ms` last
        hb = extra.get("latency_ms_host_buffer") or {}
        out = {k: full_rec[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "gpu_ms_last_chunk", "higher_is_better", "scaling",
                                        "vs_baseline", "dtype", "data")}
        out["config"] = {"workload": workload, "streams_per_gpu": S, "streams_total": S * job.world, "chunk_ms": 160, "hip_graph": graph,
                         "timed_api": "rvc_infer_device (inputs in HBM, one sync per chunk)"}
        out["latency_ms"] = head["latency_ms"]; out["rtf"] = head["rtf"]; out["rccl_ranks"] = rccl_ranks
        out["per_rank_frames_per_s"] = head["per_rank_frames_per_s"]
        out["latency_ms_host_buffer"] = {k: hb[k] for k in ("p50", "p99", "samples") if k in hb} or None
        out["plugin_chain_ms_per_chunk"] = (extra.get("plugin_chain") or {}).get("ms_per_chunk")
        out["plugin_chain_panic_chunks"] = (extra.get("plugin_chain") or {}).get("panic_chunks")
        out["offline_pipelined_frames_per_s"] = extra.get("offline_pipelined_frames_per_s")
        out["roofline"] = compact_roofline(head.get("roofline"))
        out["cpu_baseline"] = cpu
        out["peak_measured"] = [{k: c[k] for k in ("when", "mfma_f32_tflops", "mfma_sclk_mhz", "hbm_read_tbs", "error") if k in c} for c in calib] or None
        hl = head.get("box_under_load") or {}
        out["box"] = {"gpu": gpu_name, "pci": bdf, "uuid": CTX["box"].uuid, "sensors": CTX["box"].source, "idle": box_idle, "headline": {"sclk": hl.get("sclk_mhz"), "sclk_min_xcd": hl.get("sclk_mhz_min_xcd"), **{k: v for k, v in (hl.get("sensors") or {}).items() if k != "samples"}}}
        out["sub_configs"] = {k: compact_sub(v) for k, v in sub.items()} or None
        if job.world > 1 and sub.get("streams64"):
            c4 = sub["streams64"]
            out["config4"] = dict(compact_sub(c4), streams_total=c4.get("streams_total"), n_gpus=c4.get("n_gpus"), per_rank_frames_per_s=c4.get("per_rank_frames_per_s"),
                                  index_broadcast={k: v for k, v in (c4.get("index_broadcast") or {}).items() if k != "via"})
        out["serial_branches"] = bool(args.serial_branches)
        out["plan_autotune"] = bool(CTX["autotune"])
        if args.hook:
            out["hooks"] = args.hook
        out["keys"] = "sub_configs: ms per step, frames/s, p50 / p99 ms, GPU ms of the last chunk, sclk MHz / W / C while the leg ran, frac (events, nominal peak), frac_wall, frac_meas (vs peak_measured[0]), frac_clk (nominal peak at the leg's clock)"
        out["full_record"] = full_path
        summ = {"headline": head["ms_per_step"]}
        for k in ("index100k", "streams2", "streams4", "streams8", "streams16", "streams32", "streams64", "streams64_index100k", "v1_256", "streams64_bf16x3"):
            if k in sub:
                summ[k] = (sub[k] or {}).get("ms_per_step")
        out["summary_ms"] = summ           # LAST key: survives a truncated tail
        # librccl prints a version banner through C stdio when its first communicator is created: push it out now so that the
        # JSON line below is the LAST line on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    job.close()


if __name__ == "__main__":
    main()
