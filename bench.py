#!/usr/bin/env python3
"""bench.py -- headline benchmark of the cvsteer hot path on MI355X.

Metric (BASELINE.json): Mpix/s for the G2+H2 7-basis filter + scalar steer at 4096x4096 f32,
and the fraction of the HBM roofline the dominant kernel reaches.

A "step" = one pass of the hot path over one 4096x4096 synthetic image that is already
resident in HBM: SteerableFiltersG2::setup's 7 separable filters (reference
SteerableFiltersG2.cpp:62-68) + steer(theta=0.3) (G2.cpp:137-145), as ONE fused kernel launch
(cvs_setup_steer).  Basis planes are persisted (7 planes) and g2/h2 written: 4 B read + 36 B
written = 40 algorithmic bytes per pixel (SURVEY.md 8(d) "M2").

The ONE JSON line (round 5: <= 6 KB, so that it survives the driver's 8 KB tail):
  * contract keys at the top level; `roofline` carries, as FLAT scalars (nested objects do not survive the driver's parse), the
    headline kernel's figures and the fractions of the legs that matter: m1_* (north_star's own target: the basis pass alone),
    fresh_frac (a new image every call), one_object_frac / first_call_frac (the reference's one object per image), after_idle_frac,
    m4 / m5 / g4 / c3 / c4, and what RCCL saw (rccl_ranks, transport, scatter / compute / gather ms);
  * `legs`: name -> [frac_hbm, ms, ms_min, ms_max, config index, frac_valu] with the launch configurations listed once in
    `launch_configs`; frac_valu = the leg's VALU roof (vector instructions per pixel and the cycles per instruction of the kernel's
    instruction mix from profiles/valu_insts.json, 1024 SIMDs, the sustained shader clock of this run) so that `bound` can be min(hbm, valu);
  * probe-only legs (8-bit inputs, untuned twin, host planes, two streams, tuner-off twins, separate outputs, pyramid parts)
    run with --all-legs.

Multi-GPU: the image/batch axis shards with no data-path collective -- every rank filters its own images (weak scaling); RCCL
is used for the barrier and the max-over-ranks reduction, for a self-test of the transport (one 1080p frame per rank through
ncclSend/ncclRecv, bytes compared; after the headline and M1, which use no transport, and before every leg that does) and in the `C4_e2e` leg for the scatter of frames from rank 0 and the gather of results.

    python bench.py [--gpus N] [--steps K] [--warmup W]
        N > 1 without WORLD_SIZE in the environment: this process starts N rank processes
        (one per GPU, RCCL rendezvous on 127.0.0.1) BEFORE touching the GPU, waits for them and
        exits with their status; rank 0 prints the JSON line.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
        the launcher provides RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; --gpus must equal WORLD_SIZE.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
N_SIMD = 1024              # 256 CUs x 4 SIMDs
VALU_CYCLES_PER_INST = 4.4 # default when profiles/valu_insts.json has no figure for a kernel: a wave64 vector instruction with a scalar-register operand
                           # (or a packed one) takes a SIMD for ~4.4 cycles, two tap-free simple ones share that time (tools/valu_rate.hip, profiles/r05_valu_rate.txt)
NOMINAL_SCLK_MHZ = 2000.0  # used for the VALU roof only when the card's clock cannot be read
ROWS = COLS = 4096
THETA = 0.3
BYTES_PER_PIX = {"M1": 32, "M2": 40, "M3": 64, "M4": 52, "M5": 84, "M6": 48, "M6s": 56, "M2_u8": 37, "C4_feat3": 16, "C4_u8_feat3": 13}
MAX_SETTLE_CALLS = 2100    # the online tuner compares its candidates on the caller's own calls, in sustained turns of 20-100 calls (at most
                           # a burn-in round + 4 rounds x 4 candidates); legs call until it has decided
EXIT_WATCHDOG = 3     # secondary legs ran into --extra-timeout: headline printed, status non-zero
EXIT_LEGS_FAILED = 5  # a secondary leg raised: headline + the legs finished so far are printed first
EXIT_TERMINATED = 4   # SIGTERM (another rank failed / the launcher gave up): whatever was measured is printed first
LEAD_IN_MS = 20.0


def _dist_env():
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    return ws, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def _spawn_ranks(n):
    """--gpus N without a launcher: start N rank processes of this script, one per GPU.  This parent makes no
    GPU call (it does not even import torch); it waits, forwards rank 0's stdout (inherited) and returns the
    first non-zero exit status, ending the other ranks (the exact PIDs it started) if one fails."""
    import signal
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"WORLD_SIZE": str(n), "RANK": str(r), "LOCAL_RANK": str(r), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "CVS_BENCH_SPAWNED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
        # rank 0 owns stdout (the JSON line); the other ranks' stdout goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        live = list(procs)
        while live:
            time.sleep(0.1)
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print("bench.py: rank process %d exited with status %d; stopping the others" % (p.pid, code), file=sys.stderr)
                    for q in live:   # every rank prints what it has (rank 0: the JSON line) from its SIGTERM watcher thread
                        q.send_signal(signal.SIGTERM)
    except KeyboardInterrupt:
        for q in procs:
            if q.poll() is None:
                q.send_signal(signal.SIGTERM)
        rc = 130
    for q in procs:
        try:
            q.wait(timeout=30)
        except Exception:
            q.kill()
    return rc


def _time_steps(torch, fn, steps, warmup, barrier, repeats=1, idle_s=0.0):
    """W untimed warm-ups, then R regions of exactly K steps, each between barrier + synchronize on both sides;
    returns ([wall seconds per region], [HIP-event milliseconds per region, on the launch stream]).

    Every region is led into by untimed steps of the same call (at least W, enough for about LEAD_IN_MS of GPU time) with
    nothing but the synchronize + barrier between them and the first timed step: a card that has sat idle for some tens of
    milliseconds (a generation-2 collection of the interpreter is enough) starts the next launches at a lower shader clock, and
    a region of 2-5 ms is over before the clock is back.  `idle_s` > 0 puts exactly such a pause in front of the region: the
    `*_after_idle` legs."""
    import gc
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(max(1, warmup)):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    per_ms = max(ev0.elapsed_time(ev1) / max(1, warmup), 1e-3)
    lead = max(warmup, min(400, int(LEAD_IN_MS / per_ms)))
    walls, evs = [], []
    for _ in range(repeats):
        # a collection inside a region of 0.1-1 ms steps starves the GPU and shows up as a 10x outlier of that repeat:
        # collect before the lead-in, not in there
        gc.collect()
        gc.disable()
        for _ in range(lead):
            fn()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        if idle_s > 0:
            time.sleep(idle_s)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(steps):
            fn()
        ev1.record()
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        gc.enable()
        walls.append(t1 - t0)
        evs.append(ev0.elapsed_time(ev1))
    return walls, evs


def _telemetry_files(torch, dev_index):
    """sysfs files of THIS rank's card (a box shows all cards of its host): shader clock, package power, power cap"""
    import glob
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        want = "%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        cards = [c for c in glob.glob("/sys/class/drm/card*/device") if want in os.path.realpath(c)]
        if len(cards) != 1:
            return None
        pick = lambda pat: (sorted(glob.glob(os.path.join(cards[0], pat))) or [None])[0]
        fs = {"sclk": pick("hwmon/hwmon*/freq1_input"), "power": pick("hwmon/hwmon*/power1_input") or pick("hwmon/hwmon*/power1_average"),
              "cap": pick("hwmon/hwmon*/power1_cap")}
        return fs if fs["sclk"] and fs["power"] else None
    except Exception:
        return None


def _sustained(torch, fs, fn, seconds=0.5):
    """`fn` back to back for `seconds` (one host synchronisation per 50 calls) while a thread reads the card's shader clock
    and package power every 20 ms: what the launch runs at when the power management has settled -- the timed regions are
    bursts of a few milliseconds.  Feeds the VALU roof and explains box-to-box differences; not used for `value`."""
    def rd(p):
        try:
            return int(open(p).read().split()[0])
        except Exception:
            return None
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            samples.append((rd(fs["sclk"]), rd(fs["power"])))
            time.sleep(0.02)
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler)
    th.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    dt = time.perf_counter() - t0
    stop[0] = True
    th.join()
    half = samples[len(samples) // 2:] or samples    # the second half: after the power management has reacted
    clk = [a for a, _ in half if a]
    pw = [b for _, b in half if b]
    cap = rd(fs["cap"]) if fs.get("cap") else None
    return {"ms": round(1e3 * dt / n, 5), "sclk_mhz": round(_median(clk) / 1e6) if clk else None,
            "power_w": round(_median(pw) / 1e6) if pw else None, "cap_w": round(cap / 1e6) if cap else None}


def _median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _valu_table():
    """vector instructions per OUTPUT pixel of every timed kernel (halo rows included), measured with rocprofv3 --pmc
    SQ_INSTS_VALU on this code (tools/collect_valu.py -> profiles/valu_insts.json); {} when the file is missing"""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "valu_insts.json")))
        cpi = d.get("cycles_per_inst", {})
        return {k: (v, cpi.get(k, VALU_CYCLES_PER_INST)) for k, v in d.get("per_pixel", {}).items()}
    except Exception:
        return {}


def _opencv_baseline(theta, threads_all):
    """BASELINE.md 3.3a / SURVEY 8(d): when OpenCV exists on the box, the literal reference sequence -- 7 x cv::sepFilter2D
    (SteerableFiltersG2.cpp:62-68) + the scalar steer (G2.cpp:137-145) -- on the same synthetic image, one thread and all."""
    try:
        import cv2
    except Exception:
        return None
    import numpy as np
    import cvsteer_amd as cv
    taps = [cv.make_taps(cv.KIND_G2, i, 4, 0.67) for i in range(7)]
    pairs = [cv.basis_taps(cv.KIND_G2, p) for p in range(7)]
    w = cv.steer_weights(cv.KIND_G2, theta)
    img = np.random.default_rng(1234).random((ROWS, COLS), dtype=np.float32)

    def once():
        b = [cv2.sepFilter2D(img, cv2.CV_32F, taps[kx].reshape(1, -1), taps[ky].reshape(-1, 1)) for kx, ky in pairs]
        return w[0] * b[0] + w[1] * b[1] + w[2] * b[2], w[3] * b[3] + w[4] * b[4] + w[5] * b[5] + w[6] * b[6]

    out = {"opencv": cv2.__version__}
    for label, nthr in (("opencv_1_thread", 1), ("opencv_all_threads", threads_all)):
        cv2.setNumThreads(nthr)
        once()
        reps, total = 0, 0.0
        while total < 3.0 and reps < 6:
            t0 = time.perf_counter()
            once()
            total += time.perf_counter() - t0
            reps += 1
        out[label] = round(reps * ROWS * COLS / total / 1e6, 3)
    return out


def _cpu_baseline(theta, full):
    """The CPU restatement of the reference call sequence (oracle/, kind 'port'), one thread, on a bounded sample of the same
    workload (full 4096x4096 images, about 10 s of CPU work); then, on ALL host cores the process may use, one
    image split by rows and -- the example's own model of parallelism, example/steer.cpp:169 -- one image per thread.  Flat
    scalars: nested objects do not survive the driver's parse."""
    import numpy as np
    import oracle  # test infrastructure used as the timed CPU baseline leg only
    img = np.random.default_rng(1234).random((ROWS, COLS), dtype=np.float32)
    oracle.time_g2_filter_steer(img[:256], theta, 1)  # warm caches / page in
    reps, total = 0, 0.0
    while total < 10.0 and reps < 16:
        total += oracle.time_g2_filter_steer(img, theta, 1)
        reps += 1
    from concurrent.futures import ThreadPoolExecutor
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    frame = np.random.default_rng(99).random((1080, 1920), dtype=np.float32)
    per = 12 if full else 3
    t0 = time.perf_counter()
    with ThreadPoolExecutor(avail) as pool:  # ctypes releases the GIL inside the C call
        list(pool.map(lambda _: oracle.time_g2_filter_steer(frame, theta, per), range(avail)))
    per_thread = avail * per * 1080 * 1920 / (time.perf_counter() - t0) / 1e6
    t_mt = min(oracle.time_g2_filter_steer_mt(img, theta, 1, avail) for _ in range(3))
    res = {"value": round(reps * ROWS * COLS / total / 1e6, 3), "unit": "Mpix/s", "cores": 1, "kind": "port",
           "sample": "%d x (4096x4096 f32, 7 sepFilter2D + scalar steer), 1 thread, oracle/ C restatement (-O3 -march=native)" % reps,
           "row_parallel_value": round(ROWS * COLS / t_mt / 1e6, 3), "row_parallel_cores": avail,
           "per_thread_value": round(per_thread, 3), "per_thread_cores": avail,
           "per_thread_sample": "%d threads x %d x 1080x1920 frames, one frame per thread (example/steer.cpp:169)" % (avail, per),
           "host_cpus": os.cpu_count(), "cpu_model": _cpu_model(), "opencv": "absent"}
    ocv = _opencv_baseline(theta, avail)
    if ocv:
        res.update(ocv)
    return res


def _traffic_child():
    """`bench.py --traffic-child`: nothing but the headline launch on the library's default configuration, a few times -- the
    program the parent runs under `rocprofv3 --pmc` (counters only, no tracing) to read the launch's HBM traffic"""
    import torch
    import cvsteer_amd as cv
    from cvsteer_amd import _lib as L
    gen = torch.Generator(device="cuda").manual_seed(1234)
    img = torch.rand((ROWS, COLS), generator=gen, device="cuda", dtype=torch.float32)
    f = cv.SteerableFiltersG2(None, 4, 0.67, device=0)
    f.set_option(L.OPT_AUTOTUNE, 0)    # the engine's default configuration (what the tuner keeps unless a challenger clearly wins)
    g, h = cv.alloc_planes(2, ROWS, COLS, device="cuda")
    for _ in range(10):
        f.setup_steer(img, THETA, flags=cv.SETUP_BASIS, out=(g, h))
    torch.cuda.synchronize()


def _live_traffic():
    """HBM bytes per headline launch from the PMC counters, measured by THIS run: two child processes (FETCH_SIZE, then
    WRITE_SIZE -- separate passes, never combined with tracing, as MI355X_MICROARCH.md prescribes) of `--traffic-child`
    under rocprofv3 (children, never an exec of this process).  gfx950 corrections as in tools/collect_profiles.py: counters are
    KiB per dispatch, FETCH_SIZE reports half of the streamed read bytes, WRITE_SIZE is exact.  None (with the reason) when
    rocprofv3 is missing, this process is itself being profiled, or a pass fails; the committed value is then replayed."""
    import csv, glob, shutil, subprocess, tempfile
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process runs under a profiler"
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "rocprofv3 not found"
    got = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="cvs_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            # its own session: on a timeout the whole group goes (rocprofv3 AND the program under it), by its exact group id
            po = subprocess.Popen([exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--traffic-child"],
                                  cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
            try:
                _o, err_txt = po.communicate(timeout=40)   # a pass takes ~3 s; a hung one must not cost the run its line
            except subprocess.TimeoutExpired:
                import signal as _sig
                try:
                    os.killpg(po.pid, _sig.SIGKILL)
                except OSError:
                    pass
                po.communicate()
                return None, "%s pass timed out" % ctr
            vals = []
            for fn in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(fn)):
                    if r.get("Counter_Name") == ctr and "k_basis" in r.get("Kernel_Name", ""):
                        vals.append(float(r["Counter_Value"]) * 1024.0)
            if po.returncode != 0 or len(vals) < 4:
                return None, "%s pass failed (rc %d, %d samples): %s" % (ctr, po.returncode, len(vals), (err_txt or b"").decode(errors="replace")[-200:])
            vals = vals[2:]     # the first launches touch fresh pages
            got[ctr] = sum(vals) / len(vals)
        except Exception as ex:
            return None, "%s pass: %s: %s" % (ctr, type(ex).__name__, ex)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"read_bytes": round(2.0 * got["FETCH_SIZE"]), "write_bytes": round(got["WRITE_SIZE"]),
            "hbm_bytes_per_launch": round(2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"])}, None


def main():
    global LEAD_IN_MS
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary legs (M1 stays: it is north_star's own target)")
    ap.add_argument("--all-legs", action="store_true", help="also run the probe-only legs (8-bit inputs, untuned twin, host planes, two streams, tuner-off twins, pyramid parts)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--strip-rows", type=int, default=0)
    ap.add_argument("--extra-timeout", type=int, default=300, help="seconds the secondary legs may take before every rank gives up on them (0 = no watchdog)")
    ap.add_argument("--repeats", type=int, default=15, help="the --steps region is timed this many times; `value` is the median (spread reported beside it)")
    ap.add_argument("--lead-ms", type=float, default=LEAD_IN_MS, help="GPU time of the untimed lead-in in front of every timed region (at least --warmup steps)")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-live-traffic", action="store_true", help="do not start the two rocprofv3 --pmc child runs; roofline.traffic is then replayed from profiles/traffic.json")
    ap.add_argument("--leg-repeats", type=int, default=7, help="repeats of every secondary leg's timed region (median reported)")
    args = ap.parse_args()
    LEAD_IN_MS = max(0.0, args.lead_ms)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.traffic_child:
        return _traffic_child()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_spawn_ranks(args.gpus))   # before torch is imported: the parent never touches the GPU
    t_start = time.perf_counter()

    # SIGTERM (a sibling rank failed and the launcher -- ours or torchrun -- stops everybody): blocked in every thread
    # of this process (set before any library starts threads; threads inherit the mask) and received by ONE watcher
    # thread through sigwait, which prints what has been measured so far and leaves with a distinct status.  A Python
    # signal handler would never run here: the main thread sits inside a collective or a device synchronisation (C code).
    import signal
    signal.pthread_sigmask(signal.SIG_BLOCK, {signal.SIGTERM})

    # stdout carries exactly one thing, the JSON line: libraries that print banners there (RCCL does, at communicator
    # creation) are sent to stderr at the file-descriptor level, and the line goes out through the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    ws, rank, local_rank = _dist_env()
    if ws != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d -- start it as `python bench.py --gpus N` or as "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (args.gpus, ws))

    import torch
    import cvsteer_amd as cv
    from cvsteer_amd import _lib as L
    from cvsteer_amd import batch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    # CVS_BENCH_TEST_BACKEND=gloo: rehearse the N > 1 code path on a box with ONE GPU (every rank on device 0,
    # collectives through host memory).  Testing only -- real runs use RCCL, one rank per GPU.
    test_backend = os.environ.get("CVS_BENCH_TEST_BACKEND", "")
    if test_backend:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d needs GPU %d but only %d visible (one rank per GPU; "
                         "CVS_BENCH_TEST_BACKEND=gloo rehearses N ranks on one GPU)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    backend = "none"
    if ws > 1:
        import torch.distributed as dist
        backend = test_backend or "nccl"
        if test_backend:
            dist.init_process_group(test_backend)
        else:
            dist.init_process_group("nccl", device_id=dev)   # "nccl" IS RCCL on ROCm
        assert dist.get_world_size() == ws == args.gpus
    cdev = "cpu" if test_backend else dev    # where collective payloads live

    def barrier():
        if dist is not None:
            dist.barrier()

    def max_over_ranks(*vals):
        if dist is None:
            return vals
        t = torch.tensor(vals, device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return tuple(float(v) for v in t)

    # which physical device every rank sits on: a run with N ranks on fewer than N distinct GPUs is a rehearsal and must
    # never be scored as a scaling result
    try:
        my_uuid = str(torch.cuda.get_device_properties(local_rank).uuid)
    except Exception:
        my_uuid = "device-%d" % local_rank
    uuids = [my_uuid]
    if dist is not None:
        uuids = [None] * ws
        dist.all_gather_object(uuids, my_uuid)
    distinct = len(set(uuids)) == len(uuids)
    if ws > 1 and not test_backend and not distinct:
        raise SystemExit("bench.py: %d ranks on %d distinct GPUs -- one rank per GPU is required for a real run "
                         "(CVS_BENCH_TEST_BACKEND=gloo is the rehearsal mode)" % (ws, len(set(uuids))))

    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    img = torch.rand((ROWS, COLS), generator=gen, device=dev, dtype=torch.float32)  # i.i.d. uniform [0,1)
    f = cv.SteerableFiltersG2(None, 4, 0.67, device=local_rank)   # a handle as the library hands it out: no option is touched
    if args.strip_rows:
        f.set_strip_rows(args.strip_rows)
    # the two output planes as the Python API itself allocates them when the caller passes none: rows of one block
    # ([row][plane][column], cv.alloc_planes), strided views like any cv::Mat ROI
    g, h = cv.alloc_planes(2, ROWS, COLS, device=dev)
    npix = ROWS * COLS

    def step():
        f.setup_steer(img, THETA, flags=cv.SETUP_BASIS, out=(g, h))

    # the one-line result and its emitter exist before anything is timed: the SIGTERM watcher and the watchdog print
    # whatever has been measured when they fire
    out = {"metric": "Mpix/s for G2+H2 7-basis filter+steer at 4096x4096 f32; % HBM roofline", "value": None, "unit": "Mpix/s",
           "n_gpus": ws, "steps": args.steps, "warmup": args.warmup}
    legs, configs = {}, []
    done_flag = {"printed": False}
    lock = threading.Lock()

    def emit(final, why=None):
        with lock:
            if done_flag["printed"]:
                return
            done_flag["printed"] = True
            if rank == 0:
                line = None
                for _try in range(5):   # the main thread may be adding a leg at this very moment (watchdog / SIGTERM path)
                    try:
                        snap = dict(out)
                        # the same figures once more in the nested form the round-4 verdict named (roofline.m1.frac, roofline.fresh.frac ...);
                        # the flat keys stay: whoever keeps only scalars of `roofline` still has them
                        rfn = dict(snap.get("roofline") or {})
                        if "m1_frac" in rfn:
                            rfn["m1"] = {"frac": rfn["m1_frac"], "avg_launch_ms": rfn.get("m1_ms"), "ms_min": rfn.get("m1_ms_min"), "ms_max": rfn.get("m1_ms_max")}
                        for nm in ("fresh", "one_object", "first_call", "after_idle"):
                            if nm + "_frac" in rfn:
                                rfn[nm] = {"frac": rfn[nm + "_frac"]}
                        if rfn:
                            snap["roofline"] = rfn
                        if legs:
                            snap["legs_fmt"] = "[frac_hbm, ms, ms_min, ms_max, launch_configs index, frac_valu]"
                            snap["legs"] = dict(legs)
                            snap["launch_configs_fmt"] = "[block_order, strip_rows, nt_stores, state_layout, warm, wg_per_cu, tuned]"
                            snap["launch_configs"] = list(configs)
                        if not final:
                            snap["extra_error"] = why or "incomplete"
                            snap.setdefault("cpu_baseline", None)
                        snap["bench_wall_s"] = round(time.perf_counter() - t_start, 1)
                        line = json.dumps(snap, separators=(",", ":"))
                        break
                    except RuntimeError:
                        time.sleep(0.01)
                sys.stdout.flush()
                os.write(json_fd, ((line or json.dumps({"metric": out["metric"], "value": out.get("value"), "extra_error": why or "incomplete"})) + "\n").encode())

    def sigterm_watcher():
        signal.sigwait({signal.SIGTERM})
        emit(False, "terminated by SIGTERM before the run finished (a sibling rank failed, or the launcher gave up); "
                    "the line holds what had been measured")
        os._exit(EXIT_TERMINATED)

    threading.Thread(target=sigterm_watcher, daemon=True).start()

    def cfg_index(handle):
        li = handle.launch_info()
        c = [li["block_order"], li["strip_rows"], li["nt_stores"], li["state_layout"], li["warm"], li["wg_per_cu"], li["tuned"]]
        if c not in configs:
            configs.append(c)
        return configs.index(c)

    def settle(fn, handle=None, n=None):
        """calls before a timed region on a new (handle, entry point, shape): the online tuner compares its candidates on the
        caller's own calls and says when it has decided (cvs_launch_info.tune_state)"""
        if n is None:
            n = MAX_SETTLE_CALLS if handle is not None else 8
        for i in range(n):
            fn()
            if handle is not None and i % 25 == 24:
                torch.cuda.synchronize()     # the tuner reads its samples back when their launches have finished, never by waiting
                hs_ = handle if isinstance(handle, (list, tuple)) else [handle]
                if all(h_.launch_info()["tune_state"] != 1 for h_ in hs_):
                    break
        torch.cuda.synchronize()

    # (the transport self-test runs right after the headline and M1 -- which touch no transport: images per rank, no collective on the data
    # path -- and BEFORE every leg that uses the batch layer, under the watchdog that guards those legs: a transport that hangs must not
    # take the headline with it)
    mg = {"rccl_ranks": ws if (ws > 1 and not test_backend) else 0, "transport": "none", "selftest": "n/a"}
    settle(step, f)
    R = max(1, args.repeats)
    walls, evs = _time_steps(torch, step, args.steps, args.warmup, barrier, repeats=R)
    # per repeat: the slowest rank; then the median over the repeats
    both = max_over_ranks(*(walls + evs))
    walls, evs = list(both[:R]), list(both[R:])
    wall, ev_ms = _median(walls), _median(evs)
    value = ws * args.steps * npix / wall / 1e6
    k_ms = ev_ms / args.steps  # average launch-to-launch duration of the single kernel, HIP events on the launch stream
    achieved = BYTES_PER_PIX["M2"] * npix / (k_ms * 1e-3) / 1e9
    achieved_wall = BYTES_PER_PIX["M2"] * npix / (wall / args.steps) / 1e9

    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("k_basis_g2_steer_4096", {}).get("hbm_bytes_per_launch")
            traffic_source = "replayed from profiles/traffic.json (committed rocprofv3 --pmc passes)"
        except Exception:
            traffic = None

    valu_tab = _valu_table()
    clock = {"mhz": None}

    def frac_valu(key, pix, ms):
        """fraction of the VALU roof: vector instructions of the launch (counted: SQ_INSTS_VALU) x the cycles per instruction of the kernel's
        instruction mix (tools/valu_model.py) / (1024 SIMDs x shader clock) / measured time"""
        per = valu_tab.get(key)
        if not per:
            return None
        mhz = clock["mhz"] or NOMINAL_SCLK_MHZ
        return round(per[0] * pix * per[1] / N_SIMD / (mhz * 1e6) / (ms * 1e-3), 4)

    out.update({
        "value": round(value, 1), "ms_per_step": round(wall / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "G2+H2 7-basis separable pass + scalar steer, one 4096x4096 f32 image per GPU per step, resident in HBM (BASELINE configs[1])",
                   "rows": ROWS, "cols": COLS, "width": 4, "spacing": 0.67, "theta": THETA, "sharding": "images per rank, no collective",
                   "backend": backend, "ranks_started_by": "bench.py" if os.environ.get("CVS_BENCH_SPAWNED") else ("launcher" if ws > 1 else "single process"),
                   "library_defaults": not args.strip_rows, "repeats": R, "lead_in_ms": LEAD_IN_MS,
                   "Mpix_s_min": round(ws * args.steps * npix / max(walls) / 1e6, 1), "Mpix_s_max": round(ws * args.steps * npix / min(walls) / 1e6, 1),
                   "distinct_devices": distinct, "launch_config": cfg_index(f)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_wall": round(achieved_wall / HBM_PEAK_GBS, 4),
                     "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": "cvs::k_basis<BankG2, F_STEER>", "algorithmic_bytes_per_launch": BYTES_PER_PIX["M2"] * npix,
                     "avg_launch_ms": round(k_ms, 5), "ms_min": round(min(evs) / args.steps, 5), "ms_max": round(max(evs) / args.steps, 5),
                     "rccl_ranks": mg["rccl_ranks"], "transport": mg["transport"], "rccl_selftest": mg["selftest"]},
    })
    rf = out["roofline"]

    # ---- north_star's own target: the G2+H2 7-basis separable pass ALONE (M1, 32 B/pix), same handle, same image, library
    # defaults, timed like the headline (median of the repeats, slowest rank) ----
    def step_m1():
        f.setup(img, flags=cv.SETUP_BASIS)

    settle(step_m1, f)
    _w1, e1 = _time_steps(torch, step_m1, args.steps, args.warmup, barrier, repeats=R)
    e1 = sorted(v / args.steps for v in max_over_ranks(*e1))
    m1_ms = _median(e1)
    rf.update({"m1_frac": round(BYTES_PER_PIX["M1"] * npix / (m1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "m1_ms": round(m1_ms, 5),
               "m1_ms_min": round(e1[0], 5), "m1_ms_max": round(e1[-1], 5), "m1_Mpix_s": round(npix / (m1_ms * 1e-3) / 1e6, 1),
               "m1_target": ">= 0.70 (north_star)"})
    legs["M1_basis"] = [rf["m1_frac"], rf["m1_ms"], rf["m1_ms_min"], rf["m1_ms_max"], cfg_index(f), None]
    settle(step, None, 4)   # back to the headline entry point

    # ---- secondary legs (reported, not the headline) ----
    # Insurance for runs with several ranks: the secondary legs contain collectives (barriers, the RCCL scatter / gather
    # of `C4_e2e`).  If a rank fails inside a leg, the others would wait in a collective for ever and the headline measured
    # above would be lost with them.  A watchdog armed for the secondary legs makes every rank leave after `--extra-timeout`
    # seconds: rank 0 prints the JSON line with the headline, the legs finished so far and an `extra_error` note, and all
    # ranks exit with status EXIT_WATCHDOG (non-zero).  A rank that CRASHES is covered by the SIGTERM watcher above.
    def watchdog():
        emit(False, "secondary legs did not finish within %d s (watchdog); headline unaffected" % args.extra_timeout)
        os._exit(EXIT_WATCHDOG)   # non-zero: a hung or failed set of secondary legs must not look like a clean run

    timer = None
    if not args.no_extra and args.extra_timeout > 0:
        timer = threading.Timer(args.extra_timeout, watchdog)
        timer.daemon = True
        timer.start()

    def run_extras():
        if os.environ.get("CVS_BENCH_TEST_CRASH_RANK") == str(rank):   # tests only: a rank that dies inside the secondary legs
            os._exit(17)
        # ---- self-test of the transport before any leg uses it (runs of several ranks on distinct GPUs): one 1080p frame per rank
        # through the native batch layer's ncclSend / ncclRecv, results compared bit for bit with rank 0's own single-GPU run.  Every rank
        # reaches the agreement collective below exactly once, whatever happened to it before.
        if ws > 1 and not test_backend:
            ok, err = 1.0, None
            try:
                nb0 = batch.NativeBatch.from_torch_distributed(local_rank)
                mg["transport"] = nb0.transport
                if distinct and nb0.transport != "rccl":
                    raise RuntimeError("ranks on distinct GPUs but transport %r" % nb0.transport)
                nb0.set_persist(False)
                frames = torch.rand((ws, 1080, 1920), generator=torch.Generator(device=dev).manual_seed(77), device=dev) if rank == 0 else None
                res = torch.empty((ws, 3, 1080, 1920), device=dev) if rank == 0 else None
                nb0.run(frames, ws, (1080, 1920), outputs=(5, 6, 7), out=res)
                if rank == 0:
                    fr_ = cv.SteerableFiltersG2(None, 4, 0.67, device=local_rank)
                    fr_.set_persist(False)
                    want = fr_.pipeline_batch(frames, outputs=(5, 6, 7))
                    torch.cuda.synchronize()
                    ok = 1.0 if torch.equal(want, res) else 0.0
                    del fr_, want
                nb0.close()
                del frames, res
            except Exception as ex:
                err = "error: %s: %s" % (type(ex).__name__, str(ex)[:80])
            (bad, failed) = max_over_ranks(1.0 - ok, 1.0 if err else 0.0)
            mg["selftest"] = err or ("error on another rank" if failed else ("ok" if bad == 0.0 else "MISMATCH"))
            rf.update({"transport": mg["transport"], "rccl_selftest": mg["selftest"]})
            if bad != 0.0:
                raise RuntimeError("frames sent through ncclSend/ncclRecv came back different from the single-GPU run")
        ksteps, kwarm = args.steps, max(5, args.warmup)
        LR = max(1, args.leg_repeats)

        def timed(fn, steps, warm, idle_s=0.0):
            """median / min / max over LR repeats of the HIP-event time per step (slowest rank per repeat)"""
            _w, e_ = _time_steps(torch, fn, steps, warm, barrier, repeats=LR, idle_s=idle_s)
            per = sorted(v / steps for v in max_over_ranks(*e_))
            return _median(per), per[0], per[-1]

        def record(name, ms, lo, hi, bpp, pix, handle=None, valu_key=None):
            fr = round(bpp * pix / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            legs[name] = [fr, round(ms, 5), round(lo, 5), round(hi, 5), cfg_index(handle) if handle is not None else None,
                          frac_valu(valu_key, pix, ms) if valu_key else None]
            return fr

        def leg(name, fn, bpp, pix=npix, steps=None, warm=None, handle=None, settle_calls=None, valu_key=None, idle_s=0.0):
            settle(fn, handle if settle_calls is None else None, settle_calls)
            ms, lo, hi = timed(fn, steps or ksteps, kwarm if warm is None else warm, idle_s)
            return record(name, ms, lo, hi, bpp, pix, handle, valu_key)

        # clock and power of this card while the headline runs back to back (the VALU roofs below use this clock)
        fs = _telemetry_files(torch, local_rank) if rank == 0 else None
        if fs:
            sus = _sustained(torch, fs, step)
            clock["mhz"] = sus["sclk_mhz"]
            rf.update({"sclk_mhz": sus["sclk_mhz"], "power_w": sus["power_w"], "power_cap_w": sus["cap_w"], "sustained_ms": sus["ms"]})
        rf["valu_frac"] = frac_valu("M2", npix, k_ms)
        rf["m1_valu_frac"] = legs["M1_basis"][5] = frac_valu("M1", npix, m1_ms)

        # the headline loop re-filters ONE 64 MiB image, which can stay resident in the 256 MiB Infinity Cache
        # between steps; this leg rotates 8 distinct images (512 MiB) so every input read comes from HBM
        imgs8 = [img] + [torch.rand((ROWS, COLS), generator=gen, device=dev, dtype=torch.float32) for _ in range(7)]
        rot = {"i": 0}

        def step_rot():
            rot["i"] = (rot["i"] + 1) & 7
            f.setup_steer(imgs8[rot["i"]], THETA, flags=cv.SETUP_BASIS, out=(g, h))

        rf["fresh_frac"] = leg("M2_fresh_8_rotating", step_rot, BYTES_PER_PIX["M2"], handle=f, valu_key="M2")

        # the reference's usage pattern: ONE object per image (example/steer.cpp:86, test/test.cpp:85).
        # (a) a loop of 64 objects -- create, one fused call, destroy -- on a stream of different images WITHOUT any host
        #     synchronisation between them (cvs_destroy parks the state block with an event): HIP events around the loop / 64.
        # (b) the same object by object with a synchronisation after each (latency view): events around the single call.
        def object_loop(nobj):
            e0, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(nobj):
                fo = cv.SteerableFiltersG2(None, 4, 0.67, device=local_rank)
                fo.setup_steer(imgs8[i & 7], THETA, flags=cv.SETUP_BASIS, out=(g, h))
                del fo
            e1_.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1_) / nobj

        object_loop(MAX_SETTLE_CALLS)
        per_obj = sorted(max_over_ranks(*[object_loop(64) for _ in range(LR)]))
        rf["one_object_frac"] = record("M2_one_object_per_image", _median(per_obj), per_obj[0], per_obj[-1], BYTES_PER_PIX["M2"], npix, valu_key="M2")

        def one_object(image):
            e0, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fo = cv.SteerableFiltersG2(None, 4, 0.67, device=local_rank)
            e0.record()
            fo.setup_steer(image, THETA, flags=cv.SETUP_BASIS, out=(g, h))
            e1_.record()
            torch.cuda.synchronize()
            del fo
            return e0.elapsed_time(e1_)

        runs = sorted(one_object(imgs8[(2 + i) & 7]) for i in range(10))
        rf["first_call_frac"] = record("M2_first_call_synchronised", _median(runs), runs[0], runs[-1], BYTES_PER_PIX["M2"], npix)
        # what a synchronised first call cannot avoid: the device is idle when the call starts, so the launch's dispatch latency lies INSIDE the
        # event pair (in every other leg it hides behind the previous launch) -- measured as the event-to-event time of a one-element fill
        tiny = torch.empty(64, device=dev)

        def idle_floor():
            e0, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record(); tiny.zero_(); e1_.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1_)
        floor_ms = _median([idle_floor() for _ in range(15)])
        rf["first_call_idle_launch_floor_us"] = round(floor_ms * 1e3, 1)
        rf["first_call_frac_net_of_floor"] = round(BYTES_PER_PIX["M2"] * npix / ((_median(runs) - floor_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        # a caller that does host work between images (example/steer.cpp:73-122) meets an idle card: 30 ms of pause before every region
        rf["after_idle_frac"] = leg("M2_after_idle", step, BYTES_PER_PIX["M2"], handle=f, settle_calls=4, valu_key="M2", idle_s=0.03)

        if args.all_legs:
            # 8-bit images, what the reference's callers hold (test/test.cpp:73,85): read as bytes by the kernel itself
            imgs8_u8 = [(im * 255.0).to(torch.uint8) for im in imgs8]

            def step_rot_u8():
                rot["i"] = (rot["i"] + 1) & 7
                f.setup_steer(imgs8_u8[rot["i"]], THETA, flags=cv.SETUP_BASIS, out=(g, h))

            leg("M2_u8_fresh_8_rotating", step_rot_u8, BYTES_PER_PIX["M2_u8"], handle=f)
            del imgs8_u8
            fu = cv.SteerableFiltersG2(None, 4, 0.67, device=local_rank)   # the engine's defaults without the online tuner
            fu.set_option(L.OPT_AUTOTUNE, 0)
            leg("M2_untuned_handle", lambda: fu.setup_steer(img, THETA, flags=cv.SETUP_BASIS, out=(g, h)), BYTES_PER_PIX["M2"], handle=fu, settle_calls=4)
            del fu
        del imgs8

        if ws == 1:
            # The facade's own regime (what a cv::Mat caller of test/test.cpp:85-90 gets): HOST planes in and out, PCIe-inclusive, never
            # `value`.  Beside it the link's roof: the bytes that must cross in each direction / the rate of a pinned 256 MiB copy in that
            # direction measured here (the two directions overlap: full duplex), so that the figure reads as a fraction of what the link allows.
            import numpy as np
            pin = torch.empty(64 << 20, dtype=torch.float32).pin_memory()
            dbuf = torch.empty(64 << 20, dtype=torch.float32, device=dev)

            def link_rate(dst, src):
                best = 0.0
                for _ in range(4):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    dst.copy_(src, non_blocking=True)
                    torch.cuda.synchronize()
                    best = max(best, src.numel() * 4 / (time.perf_counter() - t0) / 1e9)
                return best
            time.sleep(1.0)   # host-link copies run at half rate for a moment after large device frees
            h2d, d2h = link_rate(dbuf, pin), link_rate(pin, dbuf)
            del pin, dbuf
            rf.update({"pcie_h2d_GBs": round(h2d, 1), "pcie_d2h_GBs": round(d2h, 1)})
            himgs = [np.random.default_rng(500 + i).random((ROWS, COLS), dtype=np.float32) for i in range(4)]
            hg, hh = np.empty_like(himgs[0]), np.empty_like(himgs[0])
            fh_ = cv.SteerableFiltersG2(None, 4, 0.67, device=local_rank)
            fh_.setup_steer(himgs[0], THETA, flags=cv.SETUP_BASIS, out=(hg, hh))
            torch.cuda.synchronize()

            def host_stream():
                t0 = time.perf_counter()
                for im in himgs:
                    fh_.setup_steer(im, THETA, flags=cv.SETUP_BASIS, out=(hg, hh))
                return (time.perf_counter() - t0) / len(himgs)

            dt = min(host_stream(), host_stream(), host_stream())
            legs["M2_host_planes_pcie_inclusive"] = [None, round(dt * 1e3, 3), None, None, None, None]
            out["pcie_inclusive_Mpix_s"] = round(npix / dt / 1e6, 1)
            roof_s = max(4.0 * npix / (h2d * 1e9), 8.0 * npix / (d2h * 1e9))    # 4 B/pix up, two f32 planes down
            rf.update({"host_e2e_Gpix_s": round(npix / dt / 1e9, 3), "host_e2e_pcie_roof_Gpix_s": round(npix / roof_s / 1e9, 3),
                       "host_e2e_pcie_frac": round(roof_s / dt, 4)})
            del himgs, fh_, hg, hh
            # the batch driver's shape (cvsteer-run, example/steer.cpp:69-122): 8-bit host frames in, the three 8-bit feature maps out --
            # 1 B/pix up, 3 B/pix down, one native call (cvs_batch_run on a one-GPU world: upload, launch, normalise, download in chunks)
            try:
                hb_ = batch.NativeBatch.local((local_rank,))
                hb_.set_persist(False)
                nfr_h = 32
                host_u8 = np.random.default_rng(77).integers(0, 256, (nfr_h, 1080, 1920), dtype=np.uint8)
                q8 = np.zeros((nfr_h, 3, 1080, 1920), np.uint8)
                bbest = None
                for rep in range(4):
                    t0 = time.perf_counter()
                    hb_.run_to_u8(host_u8, out=q8)
                    dtb = time.perf_counter() - t0
                    if rep and (bbest is None or dtb < bbest):
                        bbest = dtb
                hp_ = nfr_h * 1080 * 1920
                roof_b = max(1.0 * hp_ / (h2d * 1e9), 3.0 * hp_ / (d2h * 1e9))
                legs["C4_u8_host_in_u8_host_out_pcie_inclusive"] = [None, round(bbest * 1e3, 3), None, None, None, None]
                rf.update({"host_u8_e2e_Gpix_s": round(hp_ / bbest / 1e9, 3), "host_u8_pcie_roof_Gpix_s": round(hp_ / roof_b / 1e9, 3),
                           "host_u8_pcie_frac": round(roof_b / bbest, 4)})
                hb_.close()
                del host_u8, q8
            except Exception as ex:
                rf["host_u8_error"] = ("%s: %s" % (type(ex).__name__, ex))[:120]

        if ws == 1 and args.all_legs:
            def same_handle_default(name, fn, bpp):
                """the leg just timed, on the SAME handle (same state block) with the tuner switched off: what the tuner's pick is worth"""
                f.set_option(L.OPT_AUTOTUNE, 0)
                leg(name, fn, bpp, handle=f, settle_calls=4)
                f.set_option(L.OPT_AUTOTUNE, 1)
                fn()
        else:
            same_handle_default = None

        if ws == 1:
            rf["m4_frac"] = leg("M4_full_setup", lambda: f.setup(img, flags=cv.SETUP_FULL), BYTES_PER_PIX["M4"], handle=f, valu_key="M4")
            if same_handle_default:
                same_handle_default("M4_tuner_off_same_handle", lambda: f.setup(img, flags=cv.SETUP_FULL), BYTES_PER_PIX["M4"])
            # the eight outputs of the pipeline as rows of ONE block ([row][plane][column]; cv.alloc_planes), the layout the
            # engine gives its own state planes: strided views like any cv::Mat ROI
            outs8 = cv.alloc_planes(8, ROWS, COLS, device=dev)
            rf["m5_frac"] = leg("M5_pipeline", lambda: f.pipeline(img, out=outs8), BYTES_PER_PIX["M5"], handle=f, valu_key="M5")
            rf["m5_after_idle_frac"] = leg("M5_pipeline_after_idle", lambda: f.pipeline(img, out=outs8), BYTES_PER_PIX["M5"], handle=f, settle_calls=4, valu_key="M5", idle_s=0.03)
            if same_handle_default:
                same_handle_default("M5_tuner_off_same_handle", lambda: f.pipeline(img, out=outs8), BYTES_PER_PIX["M5"])
                outs8s = [torch.empty_like(img) for _ in range(8)]
                leg("M5_pipeline_separate_outputs", lambda: f.pipeline(img, out=outs8s), BYTES_PER_PIX["M5"], handle=f, settle_calls=8)
                del outs8s
            f.setup(img, flags=cv.SETUP_FULL)
            leg("M3_steer_map_full", lambda: f.steer(None, full=True, out=outs8[:5]), BYTES_PER_PIX["M3"], settle_calls=4)
            if args.all_legs:
                leg("M3_steer_scalar", lambda: f.steer(THETA, out=(g, h)), 36, settle_calls=4)
            f4 = cv.SteerableFiltersG4(None, 6, 0.5, device=local_rank)
            rf["g4_frac"] = leg("M6_g4_basis", lambda: f4.setup(img), BYTES_PER_PIX["M6"], handle=f4, valu_key="M6")
            rf["g4_steer_frac"] = leg("M6_g4_filter_steer", lambda: f4.setup_steer(img, THETA, out=(g, h)), BYTES_PER_PIX["M6s"], handle=f4, valu_key="M6s")
            rf["g4_after_idle_frac"] = leg("M6_g4_basis_after_idle", lambda: f4.setup(img), BYTES_PER_PIX["M6"], handle=f4, settle_calls=4, valu_key="M6", idle_s=0.03)
            rf["g4_valu_frac"] = legs["M6_g4_basis"][5]
            rf["m5_valu_frac"] = legs["M5_pipeline"][5]
            if fs:   # what these two run at when sustained (they follow the shader clock): their VALU roofs are taken at their OWN clock
                for nm, fn_, lg in (("m5", lambda: f.pipeline(img, out=outs8), "M5_pipeline"), ("g4", lambda: f4.setup(img), "M6_g4_basis")):
                    s_ = _sustained(torch, fs, fn_, 0.4)
                    rf[nm + "_sclk_mhz"] = s_["sclk_mhz"]
                    if s_["sclk_mhz"] and legs[lg][5]:
                        legs[lg][5] = rf[nm + "_valu_frac"] = round(legs[lg][5] * (clock["mhz"] or NOMINAL_SCLK_MHZ) / s_["sclk_mhz"], 4)
                step()
            # both roofs sit at ~0.7 for the G4 pair launch at the clock the power cap leaves it (profiles/r05_g4_bound.txt): neither alone bounds it
            gv = rf["g4_valu_frac"] or 0
            rf["g4_bound"] = "valu and hbm (both roofs within 0.05)" if abs(gv - rf["g4_frac"]) <= 0.05 else ("valu" if gv > rf["g4_frac"] else "hbm")
            del outs8, f4
            # size dependence: the same kernels on one 8192x8192 image (4x the pixels per launch) -- the fixed start-up cost of
            # a launch (every wave primes its 8-row window before its first store) amortises -- and with two images taking turns
            big2 = torch.rand((8192, 8192), generator=gen, device=dev, dtype=torch.float32)
            fb = cv.SteerableFiltersG2(None, 4, 0.67, device=local_rank)
            gb, hb = torch.empty_like(big2), torch.empty_like(big2)
            bsteps = max(5, args.steps // 4)
            leg("M1_basis_8192", lambda: fb.setup(big2, flags=cv.SETUP_BASIS), 32, pix=4 * npix, steps=bsteps, warm=2, handle=fb, valu_key="M1")
            leg("M2_filter_steer_8192", lambda: fb.setup_steer(big2, THETA, flags=cv.SETUP_BASIS, out=(gb, hb)), 40, pix=4 * npix, steps=bsteps, warm=2, handle=fb, valu_key="M2")
            big3 = torch.rand((8192, 8192), generator=gen, device=dev, dtype=torch.float32)
            flipb = {"i": 0}

            def step_big_rot():
                flipb["i"] ^= 1
                fb.setup_steer(big3 if flipb["i"] else big2, THETA, flags=cv.SETUP_BASIS, out=(gb, hb))

            rf["fresh_8192_frac"] = leg("M2_filter_steer_8192_fresh_2_rotating", step_big_rot, 40, pix=4 * npix, steps=bsteps, warm=2, handle=fb, valu_key="M2")
            del big2, big3, fb, gb, hb

        # ---- BASELINE config 4: 1080 x 1920 frames, the callers' whole pipeline per frame, 32 frames per GPU ----
        # Two frame sets alternate so that every launch reads frames the previous launch did not touch; >= 10 timed steps.
        nfr = 32
        fsets = [torch.rand((nfr, 1080, 1920), generator=gen, device=dev, dtype=torch.float32) for _ in range(2)]
        fout = torch.empty((nfr, 8, 1080, 1920), device=dev)
        ff = cv.SteerableFiltersG2(None, 4, 0.67, device=local_rank)
        csteps = max(10, args.steps // 10)
        alt = {"i": 0}
        fp = nfr * 1080 * 1920

        def step_c4():
            alt["i"] ^= 1
            ff.pipeline_batch(fsets[alt["i"]], out=fout)

        rf["c4_frac"] = leg("C4_32x1080p_pipeline_state_kept", step_c4, 84, pix=fp, steps=csteps, warm=2, handle=ff, valu_key="M5")
        out["c4_Mpix_s"] = round(ws * fp / (legs["C4_32x1080p_pipeline_state_kept"][1] * 1e-3) / 1e6, 1)
        ff.set_persist(False)
        fo3 = torch.empty((nfr, 3, 1080, 1920), device=dev)

        def step_c4f():
            alt["i"] ^= 1
            ff.pipeline_batch(fsets[alt["i"]], out=fo3, outputs=(5, 6, 7))

        leg("C4_32x1080p_three_maps_only", step_c4f, BYTES_PER_PIX["C4_feat3"], pix=fp, steps=csteps, warm=2, handle=ff, valu_key="C4_feat3")
        rf["c4_maps_Gpix_s"] = round(ws * fp / (legs["C4_32x1080p_three_maps_only"][1] * 1e-3) / 1e9, 1)
        rf["c4_maps_valu_frac"] = legs["C4_32x1080p_three_maps_only"][5]
        if args.all_legs:
            fsets_u8 = [(fs_ * 255.0).to(torch.uint8) for fs_ in fsets]

            def step_c4f_u8():
                alt["i"] ^= 1
                ff.pipeline_batch(fsets_u8[alt["i"]], out=fo3, outputs=(5, 6, 7))

            leg("C4_32x1080p_u8_three_maps", step_c4f_u8, BYTES_PER_PIX["C4_u8_feat3"], pix=fp, steps=csteps, warm=2, handle=ff)
            del fsets_u8
        del fout, fo3

        # ---- config 4 end to end through the NATIVE batch entry (cvs_batch_run, cvs_batch.cpp): frames on rank 0 ->
        # scatter (grouped ncclSend/ncclRecv) -> one fused launch per rank -> gather of the three feature maps on rank 0.
        # Phase times are HIP events on the ranks' own streams (max over ranks).
        n_all = nfr * ws
        shape = (1080, 1920)
        all_frames = None
        if rank == 0:
            all_frames = fsets[0] if ws == 1 else torch.cat([fsets[0]] + [torch.rand((nfr,) + shape, generator=gen, device=dev) for _ in range(ws - 1)])
        if not test_backend:   # the rehearsal backend has no RCCL communicator to build on
            try:
                nbat = batch.NativeBatch.local((local_rank,)) if ws == 1 else batch.NativeBatch.from_torch_distributed(local_rank)
                if ws > 1 and distinct and nbat.transport != "rccl":
                    raise RuntimeError("%d ranks on distinct GPUs but the batch layer chose transport %r -- a rehearsal transport must "
                                       "never carry a real multi-GPU run" % (ws, nbat.transport))
                rf["transport"] = nbat.transport if ws > 1 else nbat.transport + " (one-rank world: plumbing only)"
                nbat.set_persist(False)
                e2e_out = torch.empty((n_all, 3) + shape, device=dev) if rank == 0 else None
                reps, acc, wall_e2e = 5, {"scatter": 0.0, "compute": 0.0, "gather": 0.0}, 0.0
                for rep in range(reps + 1):
                    torch.cuda.synchronize(); barrier(); t0 = time.perf_counter()
                    _, tm = nbat.run(all_frames, n_all, shape, outputs=(5, 6, 7), out=e2e_out)
                    barrier(); dt = time.perf_counter() - t0
                    if rep:   # the first repetition warms staging allocations and RCCL channels
                        wall_e2e += dt / reps
                        for k in acc:
                            acc[k] += tm[k] / reps
                acc = dict(zip(acc.keys(), max_over_ranks(*acc.values())))
                (wall_e2e,) = max_over_ranks(wall_e2e)
                rf.update({"scatter_ms": round(acc["scatter"], 3), "compute_ms": round(acc["compute"], 3), "gather_ms": round(acc["gather"], 3),
                           "e2e_ms_wall": round(wall_e2e * 1e3, 3), "e2e_frames": n_all,
                           "e2e_compute_Mpix_s": round(n_all * 1080 * 1920 / (acc["compute"] * 1e-3) / 1e6, 1),
                           "e2e_Mpix_s": round(n_all * 1080 * 1920 / wall_e2e / 1e6, 1)})
                nbat.close()
                del e2e_out
            except Exception as ex:   # a failing end-to-end leg must not take the headline down with it
                rf["e2e_error"] = ("%s: %s" % (type(ex).__name__, ex))[:120]
        del fsets, ff, all_frames

        if ws == 1:
            # BASELINE config 3: G2+H2 over a 5-level Gaussian pyramid of one 8192x8192 image (pyrDown is this build's own
            # component -- the reference has no pyramid code): build the pyramid AND filter every level in one native call
            # (cvs_pyramid_setup: the filter launch of level k writes level k+1, every level image is read once); two 8192^2
            # images alternate, so that no input is a leftover of the previous step in the Infinity Cache.
            bigs = [torch.rand((8192, 8192), generator=gen, device=dev, dtype=torch.float32) for _ in range(2)]
            fp3 = cv.SteerableFiltersG2(None, 4, 0.67, device=local_rank)
            lv = fp3.pyramid(bigs[0], 5)
            ppix = sum(l.shape[0] * l.shape[1] for l in lv)
            hp = [cv.SteerableFiltersG2(None, 4, 0.67, device=local_rank) for _ in lv]
            flip3 = {"i": 0}

            def pyr_one_call():
                flip3["i"] ^= 1
                cv.pyramid_setup(hp, bigs[flip3["i"]], level_images=lv[1:], flags=cv.SETUP_BASIS)

            c3 = max(10, args.steps // 10)
            # algorithmic bytes: 4 B read + 28 B written per pixel of every level, plus the 4 B written per pixel of every level made here
            whole_bytes = 32 * ppix + 4 * (ppix - lv[0].shape[0] * lv[0].shape[1])
            settle(pyr_one_call, hp)   # (the small levels are compared by the tuner on these calls too: every level's handle must have decided)
            ms, lo, hi = timed(pyr_one_call, c3, 2)
            rf["c3_frac"] = record("C3_pyramid_8192_5_levels_whole", ms, lo, hi, whole_bytes / ppix, ppix, hp[0], None)
            out["c3_Mpix_s"] = round(ppix / (ms * 1e-3) / 1e6, 1)
            if args.all_legs:
                def pyr_filter():
                    for hnd, l in zip(hp, lv):
                        hnd.setup(l, flags=cv.SETUP_BASIS)
                leg("C3_filter_only_prebuilt_pyramid", pyr_filter, 32, pix=ppix, steps=c3, warm=2, settle_calls=8)
                leg("C3_pyramid_build_only", lambda: fp3.pyramid(bigs[0], 5), 5.0 * (ppix - lv[0].shape[0] * lv[0].shape[1]) / ppix, pix=ppix, steps=c3, warm=2, settle_calls=4)
            del bigs, lv, hp, fp3
        if ws > 1 and not test_backend:
            # BASELINE config 3 over the ranks (one large image, every level split into row bands): the native
            # entry cvs_batch_pyramid_setup -- ncclBroadcast of the 8192^2 image from rank 0, every rank builds the (cheap)
            # pyramid and filters its band of every level, the bands are gathered into rank 0's state planes.
            try:
                pb = batch.NativeBatch.from_torch_distributed(local_rank)
                big = torch.rand((8192, 8192), generator=gen, device=dev, dtype=torch.float32) if rank == 0 else None
                reps, acc, wall3 = 3, {"broadcast": 0.0, "compute": 0.0, "gather": 0.0}, 0.0
                for rep in range(reps + 1):
                    torch.cuda.synchronize(); barrier(); t0 = time.perf_counter()
                    tm = pb.pyramid_setup(big, 8192, 8192, 5, flags=cv.SETUP_BASIS, root=0)
                    barrier(); dt = time.perf_counter() - t0
                    if rep:
                        wall3 += dt / reps
                        for k in acc:
                            acc[k] += tm[k] / reps
                acc = dict(zip(acc.keys(), max_over_ranks(*acc.values())))
                (wall3,) = max_over_ranks(wall3)
                ppix3 = sum((8192 >> l) ** 2 for l in range(5))
                rf.update({"c3_band_split_ranks": ws, "c3_broadcast_ms": round(acc["broadcast"], 3), "c3_band_compute_ms": round(acc["compute"], 3),
                           "c3_band_gather_ms": round(acc["gather"], 3), "c3_band_e2e_Mpix_s": round(ppix3 / wall3 / 1e6, 1)})
                pb.close()
                del big
            except Exception as ex:
                rf["c3_band_split_error"] = ("%s: %s" % (type(ex).__name__, ex))[:120]

    if not args.no_extra:
        try:
            run_extras()
        except Exception as ex:   # a failing leg must not take the headline with it: print what exists, leave non-zero
            import traceback
            traceback.print_exc()
            emit(False, "secondary legs failed on rank %d: %s: %s; headline unaffected" % (rank, type(ex).__name__, ex))
            os._exit(EXIT_LEGS_FAILED)
    if timer is not None:
        timer.cancel()

    # HBM traffic of the headline launch from the PMC counters, measured by this run -- AFTER everything that is timed (the
    # children's allocations change where this process's first state block lands), in a thread beside the CPU baseline (host
    # work only).  One rank only: the counter passes use device 0.
    live_box = {}
    th_live = None
    if rank == 0 and out.get("roofline") and not args.no_live_traffic and ws == 1:
        torch.cuda.synchronize()
        th_live = threading.Thread(target=lambda: live_box.update(zip(("live", "why"), _live_traffic())))
        th_live.start()

    # the CPU baseline runs on rank 0's host cores (the other ranks wait in the final barrier)
    if rank == 0 and not args.no_cpu:
        out["cpu_baseline"] = _cpu_baseline(THETA, args.all_legs)
    elif rank == 0:
        out["cpu_baseline"] = None
    if th_live is not None:
        th_live.join()
        live = live_box.get("live")
        if live:
            rf["traffic"] = live["hbm_bytes_per_launch"]
            rf["traffic_read_bytes"], rf["traffic_write_bytes"] = live["read_bytes"], live["write_bytes"]
            rf["traffic_source"] = "measured by this run: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, two counter-only child runs"
        elif rf.get("traffic_source"):
            rf["traffic_source"] = (rf["traffic_source"] + "; live passes: " + str(live_box.get("why")))[:128]

    emit(True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
