#!/usr/bin/env python3
"""tools/sustained_probe.py -- each entry point back to back for ~0.6 s: milliseconds per call, the shader clock and the package power
the card settles at (sysfs, 20 ms samples, second half of the run).  One line per leg; run once per library build (CVSTEER_HIP_LIB)
to compare two builds at the operating point the power management chooses for each.  Tuner off."""
import os, sys, time, threading, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CVS_OPTS", "autotune=0")
import torch
import cvsteer_amd as cv

def files():
    pr = torch.cuda.get_device_properties(0)
    want = "%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    card = [c for c in glob.glob("/sys/class/drm/card*/device") if want in os.path.realpath(c)][0]
    pick = lambda pat: (sorted(glob.glob(os.path.join(card, pat))) or [None])[0]
    return pick("hwmon/hwmon*/freq1_input"), pick("hwmon/hwmon*/power1_input") or pick("hwmon/hwmon*/power1_average")

def rd(p):
    try: return int(open(p).read().split()[0])
    except Exception: return None

def run(name, fn, bytes_per_call, seconds=0.6):
    fclk, fpow = files()
    for _ in range(20): fn()
    torch.cuda.synchronize()
    samples, stop = [], [False]
    def sampler():
        while not stop[0]:
            samples.append((rd(fclk), rd(fpow))); time.sleep(0.02)
    th = threading.Thread(target=sampler); th.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20): fn()
        torch.cuda.synchronize(); n += 20
    dt = time.perf_counter() - t0
    stop[0] = True; th.join()
    half = samples[len(samples) // 2:]
    clk = sorted(a for a, _ in half if a); pw = sorted(b for _, b in half if b)
    ms = 1e3 * dt / n
    print("%-34s %8.4f ms  %5.3f of 8 TB/s  sclk %4d MHz  power %4d W" % (name, ms, bytes_per_call / (ms * 1e-3) / 8e12, clk[len(clk) // 2] / 1e6, pw[len(pw) // 2] / 1e6), flush=True)

n = 4096
img = torch.rand((n, n), device="cuda")
g, h = cv.alloc_planes(2, n, n, device="cuda")
outs8 = cv.alloc_planes(8, n, n, device="cuda")
f = cv.SteerableFiltersG2(None, 4, 0.67)
print("library:", cv.lib_path() if hasattr(cv, "lib_path") else os.environ.get("CVSTEER_HIP_LIB", "default"))
run("M1 basis", lambda: f.setup(img, flags=cv.SETUP_BASIS), 32 * n * n)
run("M2 filter + steer", lambda: f.setup_steer(img, 0.3, flags=cv.SETUP_BASIS, out=(g, h)), 40 * n * n)
run("M4 full setup", lambda: f.setup(img, flags=cv.SETUP_FULL), 52 * n * n)
run("M5 pipeline", lambda: f.pipeline(img, out=outs8), 84 * n * n)
f4 = cv.SteerableFiltersG4(None, 6, 0.5)
run("M6 G4 basis", lambda: f4.setup(img), 48 * n * n)
del f4, outs8
frames = torch.rand((32, 1080, 1920), device="cuda")
ff = cv.SteerableFiltersG2(None, 4, 0.67)
fo3 = torch.empty((32, 3, 1080, 1920), device="cuda")
fo8 = torch.empty((32, 8, 1080, 1920), device="cuda")
fp = 32 * 1080 * 1920
run("C4 32x1080p state kept", lambda: ff.pipeline_batch(frames, out=fo8), 84 * fp)
ff.set_persist(False)
run("C4 32x1080p three maps", lambda: ff.pipeline_batch(frames, out=fo3, outputs=(5, 6, 7)), 16 * fp)
del ff, fo3, fo8, frames
bigs = [torch.rand((8192, 8192), device="cuda") for _ in range(2)]
fp3 = cv.SteerableFiltersG2(None, 4, 0.67)
lv = fp3.pyramid(bigs[0], 5)
ppix = sum(l.shape[0] * l.shape[1] for l in lv)
hp = [cv.SteerableFiltersG2(None, 4, 0.67) for _ in lv]
flip = {"i": 0}
def pyr():
    flip["i"] ^= 1
    cv.pyramid_setup(hp, bigs[flip["i"]], level_images=lv[1:], flags=cv.SETUP_BASIS)
run("C3 pyramid 8192 5 levels", pyr, 32 * ppix + 4 * (ppix - 8192 * 8192))
fb = cv.SteerableFiltersG2(None, 4, 0.67)
def m1big():
    flip["i"] ^= 1
    fb.setup(bigs[flip["i"]], flags=cv.SETUP_BASIS)
run("M1 8192 two images in turn", m1big, 32 * 8192 * 8192)
run("M1 8192 resident", lambda: fb.setup(bigs[0], flags=cv.SETUP_BASIS), 32 * 8192 * 8192)
