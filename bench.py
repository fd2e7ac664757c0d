#!/usr/bin/env python3
"""bench.py -- headline benchmark of the cvsteer hot path on MI355X.

Metric (BASELINE.json): Mpix/s for the G2+H2 7-basis filter + scalar steer at 4096x4096 f32,
and the fraction of the HBM roofline the dominant kernel reaches.

A "step" = one pass of the hot path over one 4096x4096 synthetic image that is already
resident in HBM: SteerableFiltersG2::setup's 7 separable filters (reference
SteerableFiltersG2.cpp:62-68) + steer(theta=0.3) (G2.cpp:137-145), as ONE fused kernel launch
(cvs_setup_steer).  Basis planes are persisted (7 planes) and g2/h2 written: 4 B read + 36 B
written = 40 algorithmic bytes per pixel (SURVEY.md 8(d) "M2").

Multi-GPU: the image/batch axis shards with no data-path collective -- every rank filters its
own images (weak scaling); RCCL is used for the barrier and the max-over-ranks reduction, and in
the `C4_e2e` leg for the scatter of frames from rank 0 and the gather of results to it.

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
ROWS = COLS = 4096
THETA = 0.3
BYTES_PER_PIX = {"M1": 32, "M2": 40, "M4": 52, "M5": 84, "M6": 48, "M6s": 56, "M2_u8": 37, "C4_u8_feat3": 13}
# Everything here runs on the library's DEFAULTS (round 4): plain hipMalloc state block, row-interleaved state planes, the
# online launch tuner.  The tuner compares a handful of launch configurations on the caller's own calls (no extra launches):
# a new (handle kind, entry point, shape) needs 40-55 calls to settle, which every leg makes before its timed region
# (SETTLE_CALLS).  The allocation-time placement search stays an opt-in knob of the library; `extra.M2_placement_window`
# shows what it is worth on the box the run landed on.  `M2_untuned` = CVS_OPT_AUTOTUNE 0.
SETTLE_CALLS = 60
INIT_CALLS = SETTLE_CALLS
EXIT_WATCHDOG = 3     # secondary legs ran into --extra-timeout: headline printed, status non-zero
EXIT_LEGS_FAILED = 5  # a secondary leg raised: headline + the legs finished so far are printed first
EXIT_TERMINATED = 4   # SIGTERM (another rank failed / the launcher gave up): whatever was measured is printed first


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


LEAD_IN_MS = 20.0


def _time_steps(torch, fn, steps, warmup, barrier, repeats=1, idle_s=0.0):
    """W untimed warm-ups, then R regions of exactly K steps, each between barrier + synchronize on both sides;
    returns ([wall seconds per region], [HIP-event milliseconds per region, on the launch stream]).

    Every region is led into by untimed steps of the same call (at least W, enough for about LEAD_IN_MS of GPU time) with
    nothing but the synchronize + barrier between them and the first timed step: a card that has sat idle for some tens of
    milliseconds (a generation-2 collection of the interpreter is enough) starts the next launches at a lower shader clock, and
    a region of 2-5 ms is over before the clock is back -- the launches within reach of the VALU then read 10-20 % slow
    (`tools/burst_probe.py`, profiles/r04_burst_probe.txt).  `idle_s` > 0 puts exactly such a pause in front of the region:
    the `*_after_idle` legs."""
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


def _sustained(torch, fs, fn, seconds=1.0):
    """`fn` back to back for `seconds` (one host synchronisation per 50 calls) while a thread reads the card's shader clock
    and package power every 20 ms: what the launch runs at when the power management has settled -- the timed regions above
    are bursts of a few milliseconds.  Explains box-to-box differences; not used for `value` or any roofline figure."""
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
    return {"ms_per_call": round(1e3 * dt / n, 5), "calls": n, "sclk_mhz": round(_median(clk) / 1e6) if clk else None,
            "power_w": round(_median(pw) / 1e6) if pw else None, "power_cap_w": round(cap / 1e6) if cap else None}


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


def _opencv_baseline(theta, threads_all):
    """BASELINE.md 3.3a / SURVEY 8(d): when OpenCV exists on the box, the literal reference sequence -- 7 x cv::sepFilter2D
    (SteerableFiltersG2.cpp:62-68) + the scalar steer (G2.cpp:137-145) -- on the same synthetic image, one thread and all."""
    try:
        import cv2
    except Exception:
        return "absent"
    import numpy as np
    import cvsteer_amd as cv
    taps = [cv.make_taps(cv.KIND_G2, i, 4, 0.67) for i in range(7)]
    pairs = [cv.basis_taps(cv.KIND_G2, p) for p in range(7)]
    w = cv.steer_weights(cv.KIND_G2, theta)
    img = np.random.default_rng(1234).random((ROWS, COLS), dtype=np.float32)

    def once():
        b = [cv2.sepFilter2D(img, cv2.CV_32F, taps[kx].reshape(1, -1), taps[ky].reshape(-1, 1)) for kx, ky in pairs]
        g = w[0] * b[0] + w[1] * b[1] + w[2] * b[2]
        hq = w[3] * b[3] + w[4] * b[4] + w[5] * b[5] + w[6] * b[6]
        return g, hq

    out = {"version": cv2.__version__}
    for label, nthr in (("1_thread", 1), ("all_threads", threads_all)):
        cv2.setNumThreads(nthr)
        once()
        reps, total = 0, 0.0
        while total < 4.0 and reps < 8:
            t0 = time.perf_counter()
            once()
            total += time.perf_counter() - t0
            reps += 1
        out[label] = {"value": round(reps * ROWS * COLS / total / 1e6, 3), "unit": "Mpix/s", "cores": nthr, "sample": "%d x 4096x4096" % reps}
    return out


def _cpu_baseline(theta):
    """The CPU restatement of the reference call sequence (oracle/, kind 'port'), one thread, on a bounded sample of the same
    workload (full 4096x4096 images, ~10-20 s of CPU work); then the example's own model of parallelism (one image per thread,
    example/steer.cpp:169) and one image row-parallel -- on ALL host cores the process may use, with the
    64-thread figures of earlier rounds beside them."""
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

    def per_thread(threads, per=12):
        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as pool:  # ctypes releases the GIL inside the C call
            list(pool.map(lambda _: oracle.time_g2_filter_steer(frame, theta, per), range(threads)))
        wall = time.perf_counter() - t0
        return {"value": round(threads * per * 1080 * 1920 / wall / 1e6, 3), "unit": "Mpix/s", "cores": threads,
                "sample": "%d threads x %d x (1080x1920 f32, 7 sepFilter2D + scalar steer), one frame per thread" % (threads, per)}

    def banded(threads):
        t_mt = min(oracle.time_g2_filter_steer_mt(img, theta, 1, threads) for _ in range(3))
        return {"value": round(ROWS * COLS / t_mt / 1e6, 3), "unit": "Mpix/s", "cores": threads,
                "sample": "best of 3 x (4096x4096 f32, 7 sepFilter2D + scalar steer), rows split over %d threads" % threads}

    t64 = max(1, min(avail, 64))
    res = {
        "value": round(reps * ROWS * COLS / total / 1e6, 3), "unit": "Mpix/s", "cores": 1, "kind": "port",
        "sample": "%d x (4096x4096 f32, 7 sepFilter2D + scalar steer), single thread, oracle/ C restatement "
                  "(-O3 -march=native)" % reps,
        "one_image_per_thread": per_thread(avail), "one_image_row_parallel": banded(avail),
        "host_cpus": os.cpu_count(), "usable_cpus": avail, "cpu_model": _cpu_model(),
    }
    if t64 != avail:
        res["one_image_per_thread_64"] = per_thread(t64)
        res["one_image_row_parallel_64"] = banded(t64)
    res["opencv"] = _opencv_baseline(theta, avail)
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
    f.set_option(L.OPT_AUTOTUNE, 0)    # the engine's default configuration (what the tuner keeps unless a challenger wins by 2 %)
    g, h = cv.alloc_planes(2, ROWS, COLS, device="cuda")
    for _ in range(10):
        f.setup_steer(img, THETA, flags=cv.SETUP_BASIS, out=(g, h))
    torch.cuda.synchronize()


def _live_traffic():
    """HBM bytes per headline launch from the PMC counters, measured by THIS run: two child processes (FETCH_SIZE, then
    WRITE_SIZE -- separate passes, never combined with tracing, as MI355X_MICROARCH.md prescribes) of `--traffic-child`
    under rocprofv3 (children, never an exec of this process).  gfx950 corrections as in tools/collect_profiles.py
    (calibrated on kernels of known traffic, profiles/r04_pmc_traffic.json): counters are KiB per dispatch, FETCH_SIZE
    reports half of the streamed read bytes, WRITE_SIZE is exact.  None (with the reason) when rocprofv3 is missing, this
    process is itself being profiled, or a pass fails; the committed value is then replayed and labelled so."""
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
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary legs (M1/M4/M5/G4/C3/C4)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--strip-rows", type=int, default=0)
    ap.add_argument("--extra-timeout", type=int, default=300, help="seconds the secondary legs may take before every rank gives up on them (0 = no watchdog)")
    ap.add_argument("--placement", type=int, default=0, choices=(0, 1),
                    help="CVS_OPT_PLACEMENT_SEARCH of the headline handle: 0 (default) = the library default, a plain hipMalloc block; "
                         "1 = the library's opt-in allocation-time placement search (A/B aid: extra.M2_placement_window reports it in any case)")
    ap.add_argument("--repeats", type=int, default=15, help="the --steps region is timed this many times; `value` is the median (spread reported beside it)")
    ap.add_argument("--lead-ms", type=float, default=LEAD_IN_MS, help="GPU time of the untimed lead-in in front of every timed region (at least --warmup steps)")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-live-traffic", action="store_true", help="do not start the two rocprofv3 --pmc child runs; roofline.traffic is then replayed from profiles/traffic.json")
    ap.add_argument("--leg-repeats", type=int, default=9, help="repeats of every secondary leg's timed region (median reported)")
    args = ap.parse_args()
    LEAD_IN_MS = max(0.0, args.lead_ms)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.traffic_child:
        return _traffic_child()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_spawn_ranks(args.gpus))   # before torch is imported: the parent never touches the GPU
    live, live_why = None, "the counter passes had not run yet when this line was printed"

    # SIGTERM (a sibling rank failed and the launcher -- ours or torchrun -- stops everybody): blocked in every thread
    # of this process (set before any library starts threads; threads inherit the mask) and received by ONE watcher
    # thread through sigwait, which prints what has been measured so far and leaves with a distinct status.  A Python
    # signal handler would never run here: the main thread sits inside a collective or a device synchronisation (C code).
    import signal
    import threading
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
    f = cv.SteerableFiltersG2(None, 4, 0.67, device=local_rank)
    # the headline handle is a handle as the library hands it out: no option is touched (--placement 1 is an A/B aid)
    if args.placement:
        f.set_option(L.OPT_PLACEMENT_SEARCH, args.placement)
    if args.strip_rows:
        f.set_strip_rows(args.strip_rows)
    # the two output planes as the Python API itself allocates them when the caller passes none: rows of one block
    # ([row][plane][column], cv.alloc_planes), strided views like any cv::Mat ROI
    g, h = cv.alloc_planes(2, ROWS, COLS, device=dev)
    npix = ROWS * COLS

    # ---- headline: filter + steer (M2), one fused launch per step ----
    def step():
        f.setup_steer(img, THETA, flags=cv.SETUP_BASIS, out=(g, h))

    # the one-line result and its emitter exist before anything is timed: the SIGTERM watcher and the watchdog print
    # whatever has been measured when they fire
    out = {"metric": "Mpix/s for G2+H2 7-basis filter+steer at 4096x4096 f32; % HBM roofline", "value": None, "unit": "Mpix/s",
           "n_gpus": ws, "steps": args.steps, "warmup": args.warmup}
    extra = {}
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
                        if not args.no_extra:
                            snap["extra"] = dict(extra)
                        if not final:
                            snap["extra_error"] = why or "incomplete"
                            snap.setdefault("cpu_baseline", None)
                        line = json.dumps(snap)
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

    def launch_of(handle):
        li = handle.launch_info()
        return {k: li[k] for k in ("block_order", "xcd_weights", "strip_rows", "nt_stores", "state_layout", "read_ahead", "wg_per_cu", "tuning_launches")}

    def settle(fn, n=SETTLE_CALLS):
        """calls before a timed region on a new (handle, entry point, shape): the online tuner compares its candidates on them"""
        for _ in range(n):
            fn()
        torch.cuda.synchronize()

    settle(step, INIT_CALLS)
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
            traffic_source = "profiles/traffic.json (replayed from the committed rocprofv3 --pmc passes of this kernel; not measured in this run: %s)" % live_why
        except Exception:
            traffic = None

    info = f.launch_info()
    out.update({
        "value": round(value, 1), "ms_per_step": round(wall / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "repeats": {"repeats": R, "of": "the %d-step timed region (barrier + synchronize on both sides, slowest rank per repeat)" % args.steps,
                    "ms_per_step": {"min": round(min(walls) / args.steps * 1e3, 5), "median": round(wall / args.steps * 1e3, 5),
                                    "max": round(max(walls) / args.steps * 1e3, 5)},
                    "Mpix/s": {"min": round(ws * args.steps * npix / max(walls) / 1e6, 1), "median": round(value, 1),
                               "max": round(ws * args.steps * npix / min(walls) / 1e6, 1)},
                    "event_ms_per_launch": {"min": round(min(evs) / args.steps, 5), "median": round(k_ms, 5), "max": round(max(evs) / args.steps, 5)}},
        "config": {"workload": "G2+H2 7-basis separable pass + scalar steer (theta=0.3), one 4096x4096 f32 image "
                               "per GPU per step, image resident in HBM, bases persisted (BASELINE configs[1])",
                   "rows": ROWS, "cols": COLS, "width": 4, "spacing": 0.67, "sharding": "images per rank, no collective",
                   "init_calls": INIT_CALLS, "lead_in": "every timed region follows >= W untimed steps of the same call (about %g ms of GPU time) "
                                                        "with only the synchronize + barrier in between; extra.*_after_idle = with a 30 ms pause instead" % LEAD_IN_MS,
                   "backend": backend, "ranks_started_by": "bench.py" if os.environ.get("CVS_BENCH_SPAWNED") else ("launcher" if ws > 1 else "single process"),
                   "library_defaults": not args.placement and not args.strip_rows,
                   "output_planes": "g, h = cv.alloc_planes(2, rows, cols): rows of one block, what setup_steer() allocates by itself",
                   "placement": {"mode": info["placement_mode"], "window_found": bool(info["window_found"]), "probe_ms": round(info["probe_ms"], 3),
                                 "note": "CVS_OPT_PLACEMENT_SEARCH of the headline handle; 0 = plain hipMalloc block, the library default "
                                         "(the opt-in search is reported as extra.M2_placement_window)"},
                   "launch": dict(launch_of(f), note="configuration of the timed launches: the engine's default or what its online tuner kept; "
                                                     "tuning_launches = launches issued beyond the caller's own calls")},
        "clocks": {"value": "host wall clock around the timed region, median of the repeats",
                   "roofline": "HIP events on the launch stream around the same region, median of the repeats; roofline.frac_wall = the same from the wall clock"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_wall": round(achieved_wall / HBM_PEAK_GBS, 4),
                     "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": "cvs::k_basis<BankG2, F_STEER>", "algorithmic_bytes_per_launch": BYTES_PER_PIX["M2"] * npix,
                     "avg_launch_ms": round(k_ms, 5)},
        "multi_gpu": {"rccl_ranks": ws if (ws > 1 and not test_backend) else 0, "ranks": ws, "device_uuids": uuids, "distinct_devices": distinct,
                      "transport": None, "note": "filled in by the C4_e2e / C3 band-split legs when ranks > 1"},
    })

    # ---- north_star's own target, at the top level: the G2+H2 7-basis separable pass ALONE (M1, 32 B/pix), same handle,
    # same image, library defaults, timed like the headline (median of the repeats, slowest rank) ----
    def step_m1():
        f.setup(img, flags=cv.SETUP_BASIS)

    settle(step_m1)
    R1 = R   # as many repeats as the headline: a region of 20 steps is 2 ms long, and at the power cap the card's clock control makes
             # single regions scatter by +-3 % (min / max of the repeats are in the line); the median of 15 is good to ~1 %
    _w1, e1 = _time_steps(torch, step_m1, args.steps, args.warmup, barrier, repeats=R1)
    e1 = sorted(v / args.steps for v in max_over_ranks(*e1))
    m1_ms = _median(e1)
    out["roofline_m1"] = {"bound": "hbm", "kernel": "cvs::k_basis<BankG2, 0>", "frac": round(BYTES_PER_PIX["M1"] * npix / (m1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "achieved": round(BYTES_PER_PIX["M1"] * npix / (m1_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "avg_launch_ms": round(m1_ms, 5), "ms_min": round(e1[0], 5), "ms_max": round(e1[-1], 5), "repeats": R1,
                          "Mpix/s": round(npix / (m1_ms * 1e-3) / 1e6, 1), "algorithmic_bytes_per_launch": BYTES_PER_PIX["M1"] * npix,
                          "target": "north_star: >= 0.70 of the HBM roofline on this pass", "launch": launch_of(f)}
    settle(step, 4)   # back to the headline entry point

    # ---- secondary legs (reported, not the headline) ----
    # Insurance for runs with several ranks: the secondary legs contain collectives (barriers, the RCCL scatter / gather
    # of `C4_e2e`) that have only ever run on one GPU here.  If a rank fails inside a leg, the others would wait in a
    # collective for ever and the headline measured above would be lost with them.  A watchdog armed for the secondary
    # legs makes every rank leave after `--extra-timeout` seconds: rank 0 prints the JSON line with the headline, the
    # legs finished so far and an `extra_error` note, and all ranks exit with status EXIT_WATCHDOG (non-zero).  A rank
    # that CRASHES is covered by the SIGTERM watcher above (the launcher stops the siblings; rank 0 prints first).
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
        ksteps, kwarm = args.steps, max(5, args.warmup)

        def rate(ms, bpp, pix):
            return {"Mpix/s": round(pix / (ms * 1e-3) / 1e6, 1), "ms": round(ms, 5), "GB/s": round(bpp * pix / (ms * 1e-3) / 1e9, 1),
                    "frac_hbm": round(bpp * pix / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "B/pix": bpp}

        LR = max(1, args.leg_repeats)

        def timed(fn, steps, warm):
            """median / min / max over LR repeats of the HIP-event time per step (slowest rank per repeat)"""
            _w, e_ = _time_steps(torch, fn, steps, warm, barrier, repeats=LR)
            per = sorted(v / steps for v in max_over_ranks(*e_))
            return _median(per), per[0], per[-1]

        def leg(name, fn, bpp, pix=npix, steps=None, warm=None, handle=None, settle_calls=SETTLE_CALLS):
            if settle_calls:
                settle(fn, settle_calls)
            ms, lo, hi = timed(fn, steps or ksteps, kwarm if warm is None else warm)
            extra[name] = dict(rate(ms, bpp, pix), ms_min=round(lo, 5), ms_max=round(hi, 5), repeats=LR)
            if handle is not None:
                extra[name]["launch"] = launch_of(handle)

        # the headline loop re-filters ONE 64 MiB image, which can stay resident in the 256 MiB Infinity Cache
        # between steps; this leg rotates 8 distinct images (512 MiB) so every input read comes from HBM
        imgs8 = [img] + [torch.rand((ROWS, COLS), generator=gen, device=dev, dtype=torch.float32) for _ in range(7)]
        rot = {"i": 0}

        def step_rot():
            rot["i"] = (rot["i"] + 1) & 7
            f.setup_steer(imgs8[rot["i"]], THETA, flags=cv.SETUP_BASIS, out=(g, h))

        leg("M2_rotating_8_inputs", step_rot, BYTES_PER_PIX["M2"], handle=f)

        # 8-bit images, what the reference's callers hold (test/test.cpp:73,85; example/steer.cpp:73-86): read as bytes by the
        # kernel itself, 1 B/pix of input -> 1 + 36 = 37 algorithmic bytes per pixel; 8 images take turns
        imgs8_u8 = [(im * 255.0).to(torch.uint8) for im in imgs8]

        def step_rot_u8():
            rot["i"] = (rot["i"] + 1) & 7
            f.setup_steer(imgs8_u8[rot["i"]], THETA, flags=cv.SETUP_BASIS, out=(g, h))

        leg("M2_u8_input_rotating", step_rot_u8, BYTES_PER_PIX["M2_u8"], handle=f)
        extra["M2_u8_input_rotating"]["note"] = "8 rotating 8-bit images, bytes read inside the filter kernel (no widening pass): 1 B in + 36 B out per pixel"
        del imgs8_u8

        # the engine's defaults without the online tuner, same image, same outputs
        fu = cv.SteerableFiltersG2(None, 4, 0.67, device=local_rank)
        fu.set_option(L.OPT_AUTOTUNE, 0)
        leg("M2_untuned", lambda: fu.setup_steer(img, THETA, flags=cv.SETUP_BASIS, out=(g, h)), BYTES_PER_PIX["M2"], handle=fu, settle_calls=4)
        extra["M2_untuned"]["note"] = "fresh handle, CVS_OPT_AUTOTUNE=0 (the engine's default configuration from the first call)"
        del fu

        # the reference's usage pattern: ONE object per image (example/steer.cpp:86, test/test.cpp:85).
        # (a) `M2_one_object_per_image`: a loop of 64 objects -- create, one fused call, destroy -- on a stream of different
        #     images WITHOUT any host synchronisation between them (cvs_destroy parks the state block with an event, the next
        #     object's launch is queued behind it): HIP events around the whole loop / 64.
        # (b) `M2_first_call`: the same object by object with a synchronisation after each (latency view): `ms` = events
        #     around the single call, `ms_object` = wall time of create + call + sync + destroy; cold = the process-wide
        #     state-block cache emptied first (hipMalloc of 0.8 GB inside the call).
        def object_loop(nobj):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(nobj):
                fo = cv.SteerableFiltersG2(None, 4, 0.67, device=local_rank)
                fo.setup_steer(imgs8[i & 7], THETA, flags=cv.SETUP_BASIS, out=(g, h))
                del fo
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / nobj

        object_loop(SETTLE_CALLS)
        per_obj = sorted(max_over_ranks(*[object_loop(64) for _ in range(LR)]))
        extra["M2_one_object_per_image"] = dict(rate(_median(per_obj), BYTES_PER_PIX["M2"], npix), ms_min=round(per_obj[0], 5), ms_max=round(per_obj[-1], 5),
                                                repeats=LR, objects_per_repeat=64,
                                                note="64 x (create, one fused filter+steer call on a different image, destroy), no host synchronisation "
                                                     "inside the loop; events around the loop / 64")

        def one_object(image):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            fo = cv.SteerableFiltersG2(None, 4, 0.67, device=local_rank)
            e0.record()
            fo.setup_steer(image, THETA, flags=cv.SETUP_BASIS, out=(g, h))
            e1.record()
            torch.cuda.synchronize()
            del fo
            return (time.perf_counter() - t0) * 1e3, e0.elapsed_time(e1)

        torch.cuda.synchronize()
        cv.lib().cvs_release_cached_memory()
        cold = one_object(imgs8[1])
        runs = [one_object(imgs8[(2 + i) & 7]) for i in range(10)]
        ms_call = sorted(r[1] for r in runs)[len(runs) // 2]
        ms_obj = sorted(r[0] for r in runs)[len(runs) // 2]
        extra["M2_first_call"] = dict(rate(ms_call, BYTES_PER_PIX["M2"], npix), ms_object=round(ms_obj, 4),
                                      ms_call_cold=round(cold[1], 4), ms_object_cold=round(cold[0], 4),
                                      note="one new handle per image with a host synchronisation after each; median of 10; "
                                           "ms = events around the single call (includes the host's launch latency on an idle GPU), "
                                           "ms_object = create+call+sync+destroy wall; cold = block cache emptied first")
        del imgs8

        # the library's OPT-IN placement search (CVS_OPT_PLACEMENT_SEARCH = 1, planar per-plane windows): the headline loop on a
        # handle with the knob on -- what it finds and costs on THIS box
        try:
            fp_ = cv.SteerableFiltersG2(None, 4, 0.67, device=local_rank)
            fp_.set_option(L.OPT_PLACEMENT_SEARCH, 1)
            leg("M2_placement_window", lambda: fp_.setup_steer(img, THETA, flags=cv.SETUP_BASIS, out=(g, h)), BYTES_PER_PIX["M2"], handle=fp_)
            pi_ = fp_.launch_info()
            extra["M2_placement_window"].update({"window_found": bool(pi_["window_found"]), "probe_ms": round(pi_["probe_ms"], 3),
                                                 "note": "CVS_OPT_PLACEMENT_SEARCH = 1 (opt-in tuning knob, off by default), otherwise the headline loop"})
            del fp_
        except Exception as ex:
            extra["M2_placement_window"] = {"error": "%s: %s" % (type(ex).__name__, ex)}

        if ws == 1:
            # PCIe-inclusive figure (never the headline `value`): the same unit of work with HOST planes in and out -- a
            # stream of 8 different host images, 64 MiB up and 2 x 64 MiB down per image, pageable host memory.  The call
            # overlaps upload, filtering and download band by band (cvs_host.cpp); `sequential` is the same
            # with CVS_OPT_HOST_OVERLAP = 0.  Floor of the link: 128 MiB down at ~56 GB/s = 2.4 ms per image.
            # (for a second or two after gigabytes of device memory have been released host-link copies of the process run
            # at about half rate, tools/d2h_probe.hip; the handles below are created and warmed first, then the leg waits)
            import numpy as np
            himgs = [np.random.default_rng(500 + i).random((ROWS, COLS), dtype=np.float32) for i in range(8)]
            hg, hh = np.empty_like(himgs[0]), np.empty_like(himgs[0])
            fhs = {}
            for overlap in (0, 1):
                fhs[overlap] = cv.SteerableFiltersG2(None, 4, 0.67, device=local_rank)
                fhs[overlap].set_option(L.OPT_HOST_OVERLAP, overlap)
                fhs[overlap].setup_steer(himgs[0], THETA, flags=cv.SETUP_BASIS, out=(hg, hh))
            torch.cuda.synchronize()
            time.sleep(2.5)

            def host_stream(overlap):
                t0 = time.perf_counter()
                for im in himgs:
                    fhs[overlap].setup_steer(im, THETA, flags=cv.SETUP_BASIS, out=(hg, hh))
                return (time.perf_counter() - t0) / len(himgs)

            # two interleaved passes, the better one of each mode (the first pass also pages the host arrays in)
            dt_seq, dt_ovl = min(host_stream(0), host_stream(0)), min(host_stream(1), host_stream(1))
            dt_seq, dt_ovl = min(dt_seq, host_stream(0)), min(dt_ovl, host_stream(1))
            del fhs
            extra["M2_host_planes_pcie_inclusive"] = {"Mpix/s": round(npix / dt_ovl / 1e6, 1), "ms": round(dt_ovl * 1e3, 3),
                                                      "sequential_Mpix/s": round(npix / dt_seq / 1e6, 1), "sequential_ms": round(dt_seq * 1e3, 3),
                                                      "note": "stream of 8 host f32 images in, g2/h2 out to host, bases stay on device; "
                                                              "link floor = 128 MiB down per image"}
            del himgs

        if ws == 1:
            # throughput mode: consecutive images go to two handles on two HIP streams, so the tail of one launch
            # overlaps the start-up of the next (tools/two_streams.py)
            f2 = cv.SteerableFiltersG2(None, 4, 0.67, device=local_rank)
            img2 = torch.rand((ROWS, COLS), generator=gen, device=dev, dtype=torch.float32)
            g2_, h2_ = torch.empty_like(img), torch.empty_like(img)
            side = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
            pair = [(f, img, (g, h)), (f2, img2, (g2_, h2_))]
            flip = {"i": 0}

            def step_two():
                flip["i"] ^= 1
                fi, im, oo = pair[flip["i"]]
                with torch.cuda.stream(side[flip["i"]]):
                    fi.setup_steer(im, THETA, flags=cv.SETUP_BASIS, out=oo)

            def timed_two(k):
                main = torch.cuda.current_stream()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(main)
                for st in side:
                    st.wait_event(e0)
                for _ in range(k):
                    step_two()
                for st in side:
                    main.wait_stream(st)
                e1.record(main)
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / k

            timed_two(2 * SETTLE_CALLS)
            ms2 = timed_two(ksteps)
            extra["M2_two_streams_two_images"] = dict(rate(ms2, 40, npix), note="alternating images on two handles / two streams; not the headline configuration")
            torch.cuda.current_stream().synchronize()
            f.setup_steer(img, THETA, flags=cv.SETUP_BASIS, out=(g, h))  # back on the main stream
            del f2, img2, g2_, h2_
            def same_handle_default(name, fn, bpp):
                """the leg just timed, on the SAME handle (same state block) with the tuner switched off: what the tuner's pick is
                worth.  (`M2_untuned` is another handle: another state block, worth up to +-5 % by itself on the 12- / 20-plane launches)"""
                f.set_option(L.OPT_AUTOTUNE, 0)
                leg(name, fn, bpp, handle=f, settle_calls=4)
                f.set_option(L.OPT_AUTOTUNE, 1)
                fn()

            same_handle_default("M2_default_same_handle", step, BYTES_PER_PIX["M2"])
            same_handle_default("M1_default_same_handle", step_m1, BYTES_PER_PIX["M1"])
            leg("M4_full_setup", lambda: f.setup(img, flags=cv.SETUP_FULL), BYTES_PER_PIX["M4"], handle=f)
            same_handle_default("M4_default_same_handle", lambda: f.setup(img, flags=cv.SETUP_FULL), BYTES_PER_PIX["M4"])
            # the eight outputs of the pipeline as rows of ONE block ([row][plane][column]; cv.alloc_planes), the layout the
            # engine gives its own state planes: strided views like any cv::Mat ROI.  `M5_pipeline_separate_outputs` = eight
            # separate allocations (rounds 1-3).
            outs8 = cv.alloc_planes(8, ROWS, COLS, device=dev)
            leg("M5_pipeline", lambda: f.pipeline(img, out=outs8), BYTES_PER_PIX["M5"], handle=f)
            same_handle_default("M5_default_same_handle", lambda: f.pipeline(img, out=outs8), BYTES_PER_PIX["M5"])
            outs8s = [torch.empty_like(img) for _ in range(8)]
            leg("M5_pipeline_separate_outputs", lambda: f.pipeline(img, out=outs8s), BYTES_PER_PIX["M5"], handle=f, settle_calls=8)
            del outs8s
            f.setup(img, flags=cv.SETUP_FULL)
            leg("M3_steer_scalar", lambda: f.steer(THETA, out=(g, h)), 36, settle_calls=4)
            leg("M3_steer_map_full", lambda: f.steer(None, full=True, out=outs8[:5]), 64, settle_calls=4)
            f4 = cv.SteerableFiltersG4(None, 6, 0.5, device=local_rank)
            leg("M6_g4_basis", lambda: f4.setup(img), BYTES_PER_PIX["M6"], handle=f4)
            leg("M6_g4_filter_steer", lambda: f4.setup_steer(img, THETA, out=(g, h)), BYTES_PER_PIX["M6s"], handle=f4)
            # what a lead-in without a pause is worth: the same two launches with 30 ms of idleness in front of every region
            for nm, fn_, bpp_ in (("M5_pipeline_after_idle", lambda: f.pipeline(img, out=outs8), BYTES_PER_PIX["M5"]),
                                  ("M6_g4_basis_after_idle", lambda: f4.setup(img), BYTES_PER_PIX["M6"])):
                _w, e_ = _time_steps(torch, fn_, ksteps, kwarm, barrier, repeats=LR, idle_s=0.03)
                per = sorted(v / ksteps for v in max_over_ranks(*e_))
                extra[nm] = dict(rate(_median(per), bpp_, npix), ms_min=round(per[0], 5), ms_max=round(per[-1], 5), repeats=LR,
                                 note="the card idle for 30 ms before every timed region of %d steps: the first launches run at the shader "
                                      "clock the power management had dropped to (every other leg is led into without a pause)" % ksteps)
            # clock and power of this card while three of the launches above run back to back for a second each
            fs = _telemetry_files(torch, local_rank) if rank == 0 else None
            if fs:
                sus = {"M2_filter_steer": (step, BYTES_PER_PIX["M2"]), "M5_pipeline": (lambda: f.pipeline(img, out=outs8), BYTES_PER_PIX["M5"]),
                       "M6_g4_basis": (lambda: f4.setup(img), BYTES_PER_PIX["M6"])}
                tel = {}
                for name, (fn, bpp) in sus.items():
                    tel[name] = _sustained(torch, fs, fn)
                    tel[name]["frac_hbm"] = round(bpp * npix / (tel[name]["ms_per_call"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                tel["note"] = ("each launch back to back for 1 s, host synchronisation every 50 calls; median shader clock and package power of "
                               "the second half, from the card's hwmon files.  The card sits at its power cap under these launches and the "
                               "shader clock is what the cap leaves: a lower clock on another box shows first in the VALU-heavier launches")
                out["device_telemetry"] = tel
                step()
            del outs8, f4
            # size dependence: the same kernels on one 8192x8192 image (4x the pixels per launch) -- the fixed
            # start-up cost of a launch (every wave primes its 8-row window before its first store) amortises
            big2 = torch.rand((8192, 8192), generator=gen, device=dev, dtype=torch.float32)
            fb = cv.SteerableFiltersG2(None, 4, 0.67, device=local_rank)
            gb, hb = torch.empty_like(big2), torch.empty_like(big2)
            bsteps = max(5, args.steps // 4)
            leg("M1_basis_only_8192", lambda: fb.setup(big2, flags=cv.SETUP_BASIS), 32, pix=4 * npix, steps=bsteps, warm=2, handle=fb)
            leg("M2_filter_steer_8192", lambda: fb.setup_steer(big2, THETA, flags=cv.SETUP_BASIS, out=(gb, hb)), 40, pix=4 * npix, steps=bsteps, warm=2, handle=fb)
            # ... and with two images taking turns: the two legs above re-filter ONE 256 MiB image, most of which is still in
            # the 256 MiB Infinity Cache when the next step starts (the streaming stores do not displace it); any launch in
            # between that touches 64 MiB ends that (round 2, profiles/r02_issue_and_c3_probes.txt), and so does a second image
            big3 = torch.rand((8192, 8192), generator=gen, device=dev, dtype=torch.float32)
            flipb = {"i": 0}

            def step_big_rot():
                flipb["i"] ^= 1
                fb.setup_steer(big3 if flipb["i"] else big2, THETA, flags=cv.SETUP_BASIS, out=(gb, hb))

            leg("M2_filter_steer_8192_rotating_2_inputs", step_big_rot, 40, pix=4 * npix, steps=bsteps, warm=2, handle=fb)
            del big2, big3, fb, gb, hb

        # ---- BASELINE config 4: 1080 x 1920 frames, the callers' whole pipeline per frame, 32 frames per GPU ----
        # Two frame sets alternate so that every launch reads frames the previous launch did not touch (2 x 265 MB
        # of inputs + 2.1 GB of outputs per launch pass through the 256 MiB Infinity Cache in between); >= 10 timed steps.
        nfr = 32
        fsets = [torch.rand((nfr, 1080, 1920), generator=gen, device=dev, dtype=torch.float32) for _ in range(2)]
        fout = torch.empty((nfr, 8, 1080, 1920), device=dev)
        ff = cv.SteerableFiltersG2(None, 4, 0.67, device=local_rank)
        csteps = max(10, args.steps // 10)
        alt = {"i": 0}

        def step_c4():
            alt["i"] ^= 1
            ff.pipeline_batch(fsets[alt["i"]], out=fout)

        settle(step_c4)
        ms, ms_lo, ms_hi = timed(step_c4, csteps, 2)
        fp = nfr * 1080 * 1920
        extra["C4_32x1080p_pipeline_batch"] = dict(rate(ms, 84, ws * fp), ms_per_frame=round(ms / nfr, 5), launches_per_batch=1,
                                                   frames_per_gpu=nfr, timed_steps=csteps, frame_sets=2, ms_min=round(ms_lo, 5), ms_max=round(ms_hi, 5), repeats=LR,
                                                   launch=launch_of(ff))
        ff.set_persist(False)
        fo3 = torch.empty((nfr, 3, 1080, 1920), device=dev)

        def step_c4f():
            alt["i"] ^= 1
            ff.pipeline_batch(fsets[alt["i"]], out=fo3, outputs=(5, 6, 7))

        settle(step_c4f)
        ms, ms_lo, ms_hi = timed(step_c4f, csteps, 2)
        extra["C4_32x1080p_feature_maps_only"] = dict(rate(ms, 16, ws * fp), ms_per_frame=round(ms / nfr, 5), timed_steps=csteps,
                                                      ms_min=round(ms_lo, 5), ms_max=round(ms_hi, 5), repeats=LR, launch=launch_of(ff),
                                                      note="edges + dark + bright only, no state persisted (what example/steer.cpp keeps)")
        # ... and from 8-bit frames, the example's sources (steer.cpp:73-80): bytes read inside the kernel, 1 + 12 = 13 B/pix
        fsets_u8 = [(fs * 255.0).to(torch.uint8) for fs in fsets]

        def step_c4f_u8():
            alt["i"] ^= 1
            ff.pipeline_batch(fsets_u8[alt["i"]], out=fo3, outputs=(5, 6, 7))

        settle(step_c4f_u8)
        ms, ms_lo, ms_hi = timed(step_c4f_u8, csteps, 2)
        extra["C4_32x1080p_u8_feature_maps"] = dict(rate(ms, BYTES_PER_PIX["C4_u8_feat3"], ws * fp), ms_per_frame=round(ms / nfr, 5), timed_steps=csteps,
                                                    ms_min=round(ms_lo, 5), ms_max=round(ms_hi, 5), repeats=LR, launch=launch_of(ff),
                                                    note="8-bit frames in (read as bytes by the kernel), edges + dark + bright out, no state persisted")
        del fout, fo3, fsets_u8

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
                out["multi_gpu"]["transport"] = nbat.transport
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
                extra["C4_e2e"] = {"frames": n_all, "ms": {k: round(v, 3) for k, v in acc.items()}, "ms_wall": round(wall_e2e * 1e3, 3),
                                   "compute_only_Mpix/s": round(n_all * 1080 * 1920 / (acc["compute"] * 1e-3) / 1e6, 1),
                                   "end_to_end_Mpix/s": round(n_all * 1080 * 1920 / wall_e2e / 1e6, 1),
                                   "gathered": "3 feature maps per frame on rank 0", "entry": "cvs_batch_run", "transport": nbat.transport,
                                   "world": "one process" if ws == 1 else "one process per GPU (ncclCommInitRank, id carried by torch.distributed)"}
                out["multi_gpu"]["C4_e2e"] = {"frames": n_all, "phase_ms": extra["C4_e2e"]["ms"], "ms_wall": extra["C4_e2e"]["ms_wall"],
                                              "end_to_end_Mpix/s": extra["C4_e2e"]["end_to_end_Mpix/s"], "transport": nbat.transport}
                nbat.close()
                del e2e_out
            except Exception as ex:   # a failing end-to-end leg must not take the headline down with it
                extra["C4_e2e"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
                out["multi_gpu"]["C4_e2e"] = extra["C4_e2e"]
            # ---- the same from HOST planes (what example/steer.cpp holds): each rank uploads its own frames from host memory,
            # launches, downloads its three maps -- chunked and overlapped inside the rank (cvs_batch_run, host planes).
            # PCIe-inclusive; never `value`.
            try:
                import numpy as _np
                hb = batch.NativeBatch.local((local_rank,))
                hb.set_persist(False)
                host_in = fsets[0].cpu().numpy()
                host_out = _np.empty((nfr, 3) + shape, _np.float32)
                time.sleep(2.0)   # host-link copies run at half rate for a moment after large device frees
                best, tm_best = None, None
                for rep in range(4):
                    t0 = time.perf_counter()
                    _, tm = hb.run(host_in, nfr, shape, outputs=(5, 6, 7), out=host_out)
                    dt = time.perf_counter() - t0
                    if rep and (best is None or dt < best):
                        best, tm_best = dt, tm
                (best,) = max_over_ranks(best)
                extra["C4_e2e_host_planes"] = {"frames_per_gpu": nfr, "ms_wall": round(best * 1e3, 2), "ms": {"upload": round(tm_best["scatter"], 2), "download": round(tm_best["gather"], 2), "span": round(tm_best["compute"], 2)},
                                               "end_to_end_Mpix/s": round(ws * nfr * 1080 * 1920 / best / 1e6, 1),
                                               "link_floor_ms": round(nfr * 1080 * 1920 * 4 * 3 / 56e9 * 1e3, 2),
                                               "note": "host f32 frames in, 3 host f32 maps out per frame; every rank moves its own shard over its own link; floor = the download at 56 GB/s"}
                # ... and the example's real flow: 8-bit images in, three 8-bit maps per image out (steer.cpp:73-122).  Bytes up,
                # maps kept on the GPU, normalize(0, 255, MINMAX) there (one launch pair per block), bytes down.
                host_u8 = (fsets[0] * 255.0).to(torch.uint8).cpu().numpy()
                bbest = None
                q8 = _np.zeros((nfr, 3) + shape, _np.uint8)
                for rep in range(4):
                    t0 = time.perf_counter()
                    hb.run_to_u8(host_u8, out=q8)
                    dt = time.perf_counter() - t0
                    if rep and (bbest is None or dt < bbest):
                        bbest = dt
                (bbest,) = max_over_ranks(bbest)
                extra["C4_e2e_bytes"] = {"frames_per_gpu": nfr, "ms_wall": round(bbest * 1e3, 2), "end_to_end_Mpix/s": round(ws * nfr * 1080 * 1920 / bbest / 1e6, 1),
                                         "link_floor_ms": round(nfr * 1080 * 1920 * 3 / 56e9 * 1e3, 2),
                                         "note": "8-bit frames in, 3 normalised 8-bit maps per frame out (example/steer.cpp flow); floor = the download at 56 GB/s"}
                hb.close()
                del host_in, host_out, host_u8, q8
            except Exception as ex:
                extra["C4_e2e_host_planes"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        del fsets, ff, all_frames

        if ws == 1:
            # BASELINE config 3: G2+H2 over a 5-level Gaussian pyramid of one 8192x8192 image (pyrDown is this
            # build's own component -- the reference has no pyramid code).  `whole_*` = the configuration as a user
            # runs it: build the pyramid AND filter every level, with the filter launch of level k writing level k+1
            # (cvs_setup_pyr: the image is read once per level); two 8192^2 images alternate, so that no input is a
            # leftover of the previous step in the Infinity Cache.  `filter_*` = the five filter launches alone on a
            # pyramid built beforehand (round 1's figure), `pyramid_build_ms` = the four stand-alone pyrDown launches.
            bigs = [torch.rand((8192, 8192), generator=gen, device=dev, dtype=torch.float32) for _ in range(2)]
            fp3 = cv.SteerableFiltersG2(None, 4, 0.67, device=local_rank)
            lv = fp3.pyramid(bigs[0], 5)
            ppix = sum(l.shape[0] * l.shape[1] for l in lv)
            hp = [cv.SteerableFiltersG2(None, 4, 0.67, device=local_rank) for _ in lv]
            flip3 = {"i": 0}

            def pyr_filter():
                for hnd, l in zip(hp, lv):
                    hnd.setup(l, flags=cv.SETUP_BASIS)

            def pyr_whole():
                flip3["i"] ^= 1
                cur = bigs[flip3["i"]]
                for k, hnd in enumerate(hp):
                    if k + 1 < len(hp):
                        hnd.setup_pyr(cur, flags=cv.SETUP_BASIS, out=lv[k + 1])
                        cur = lv[k + 1]
                    else:
                        hnd.setup(cur, flags=cv.SETUP_BASIS)

            def pyr_one_call():
                flip3["i"] ^= 1
                cv.pyramid_setup(hp, bigs[flip3["i"]], level_images=lv[1:], flags=cv.SETUP_BASIS)

            c3 = max(10, args.steps // 10)
            settle(pyr_filter)
            e_, _lo, _hi = timed(pyr_filter, c3, 2)
            e2_, _lo, _hi = timed(lambda: fp3.pyramid(bigs[0], 5), c3, 2)
            settle(pyr_whole)
            e4_, e4_lo, e4_hi = timed(pyr_whole, c3, 2)
            settle(pyr_one_call)
            e3_, e3_lo, e3_hi = timed(pyr_one_call, c3, 2)
            e_, e2_, e3_ = e_ * c3, e2_ * c3, e3_ * c3
            # algorithmic bytes of the whole configuration: 4 B read + 28 B written per pixel of every level, plus the
            # 4 B written per pixel of every level that is made here (levels 1..4)
            whole_bytes = 32 * ppix + 4 * (ppix - lv[0].shape[0] * lv[0].shape[1])
            extra["C3_pyramid_8192_5_levels"] = {"whole_ms": round(e3_ / c3, 4), "whole_Mpix/s": round(ppix / (e3_ / c3 * 1e-3) / 1e6, 1),
                                                "whole_frac_hbm": round(whole_bytes / (e3_ / c3 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                "whole_algorithmic_bytes": whole_bytes, "launches": len(hp),
                                                "filter_Mpix/s": round(ppix / (e_ / c3 * 1e-3) / 1e6, 1), "filter_ms": round(e_ / c3, 4),
                                                "filter_frac_hbm": round(32 * ppix / (e_ / c3 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                "pyramid_build_ms": round(e2_ / c3, 4), "total_pixels": ppix, "timed_steps": c3,
                                                "whole_ms_min": round(e3_lo, 4), "whole_ms_max": round(e3_hi, 4), "repeats": LR,
                                                "whole_entry": "cvs_pyramid_setup (the chain in one native call)",
                                                "chain_of_five_calls_ms": round(e4_, 4),
                                                "chain_of_five_calls_frac_hbm": round(whole_bytes / (e4_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                "launch_per_level": [launch_of(hnd) for hnd in hp],
                                                "note": "whole = build + filter in one cvs_pyramid_setup call (level k+1 written by the filter launch of level k), two alternating images; chain_of_five_calls = the same as five cvs_setup_pyr / cvs_setup calls from Python; "
                                                        "filter = five filter launches on a prebuilt pyramid; separate build + filter = filter_ms + pyramid_build_ms"}
            del bigs, lv, hp, fp3
        if ws > 1 and not test_backend:
            # BASELINE config 3 over the ranks (one large image, every level split into row bands): the native
            # entry cvs_batch_pyramid_setup -- ncclBroadcast of the 8192^2 image from rank 0, every rank builds the (cheap)
            # pyramid and filters its band of every level, the bands are gathered into rank 0's state planes.  Phase times
            # are HIP events on the ranks' streams (slowest rank).  Never run on more than one GPU before the driver's node.
            try:
                pb = batch.NativeBatch.from_torch_distributed(local_rank)
                if distinct and pb.transport != "rccl":
                    raise RuntimeError("%d ranks on distinct GPUs but transport %r" % (ws, pb.transport))
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
                extra["C3_pyramid_8192_5_levels_band_split"] = {
                    "ranks": ws, "ms": {k: round(v, 3) for k, v in acc.items()}, "ms_wall": round(wall3 * 1e3, 3),
                    "compute_only_Mpix/s": round(ppix3 / (acc["compute"] * 1e-3) / 1e6, 1), "end_to_end_Mpix/s": round(ppix3 / wall3 / 1e6, 1),
                    "entry": "cvs_batch_pyramid_setup", "transport": pb.transport,
                    "note": "image on rank 0 -> ncclBroadcast -> every rank: pyramid + its row band of every level -> bands gathered into rank 0's state"}
                out["multi_gpu"]["C3_band_split"] = {"phase_ms": extra["C3_pyramid_8192_5_levels_band_split"]["ms"], "ms_wall": round(wall3 * 1e3, 3),
                                                     "end_to_end_Mpix/s": round(ppix3 / wall3 / 1e6, 1), "transport": pb.transport}
                pb.close()
                del big
            except Exception as ex:
                extra["C3_pyramid_8192_5_levels_band_split"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
                out["multi_gpu"]["C3_band_split"] = extra["C3_pyramid_8192_5_levels_band_split"]

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

    # HBM traffic of the headline launch from the PMC counters, measured by this run -- AFTER everything that is timed: in
    # front of it, two of four runs had the headline and M1 3-6 % slow (tools/ab_live_traffic.sh; the children's allocations
    # change where this process's first state block lands).  One rank only: the counter passes use device 0.
    if rank == 0 and out.get("roofline"):
        if args.no_live_traffic:
            live_why = "switched off (--no-live-traffic)"
        elif ws != 1:
            live_why = "runs of several ranks replay the committed value"
        else:
            torch.cuda.synchronize()
            live, live_why = _live_traffic()
        if live:
            out["roofline"]["traffic"] = live["hbm_bytes_per_launch"]
            out["roofline"]["traffic_source"] = (
                "measured by this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, two counter-only child runs of the headline launch "
                "(`bench.py --traffic-child`, library defaults) after the timed part; read %d B (2 x FETCH_SIZE, the gfx950 correction) + "
                "written %d B" % (live["read_bytes"], live["write_bytes"]))
        elif out["roofline"].get("traffic_source"):
            out["roofline"]["traffic_source"] = out["roofline"]["traffic_source"].replace(
                "the counter passes had not run yet when this line was printed", live_why or "counter passes failed")

    # the CPU baseline runs on rank 0's host cores (the other ranks wait in the final barrier)
    if rank == 0 and not args.no_cpu:
        out["cpu_baseline"] = _cpu_baseline(THETA)
    elif rank == 0:
        out["cpu_baseline"] = None

    emit(True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
