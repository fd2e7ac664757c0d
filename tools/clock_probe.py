#!/usr/bin/env python3
"""tools/clock_probe.py -- what the box tells about its clocks and power while the filter kernels run.

Some boxes run the launches that are within reach of the VALU (pipeline, G4, three-maps batch) 10-20 % slower than others while
the HBM-bound ones barely move.  This samples whatever sysfs exposes (hwmon freq*_input / power*_average / power*_cap,
pp_dpm_sclk / pp_dpm_mclk / pp_dpm_fclk) once idle and every 50 ms during a 3 s loop of the caller pipeline and of the basis pass."""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import cvsteer_amd as cv


def my_cards():
    """the sysfs card of torch's device 0, by PCI address (a box shows all eight cards of its host)"""
    pr = torch.cuda.get_device_properties(0)
    want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", -1), getattr(pr, "pci_device_id", 0))
    hit = [c for c in sorted(glob.glob("/sys/class/drm/card*/device")) if want in os.path.realpath(c)]
    print("device 0 = %s at PCI %s -> %s" % (pr.name, want, hit))
    return hit or sorted(glob.glob("/sys/class/drm/card*/device"))


def files():
    out = []
    for card in my_cards():
        for pat in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "hwmon/hwmon*/freq*_input", "hwmon/hwmon*/freq*_label",
                    "hwmon/hwmon*/power*_average", "hwmon/hwmon*/power*_input", "hwmon/hwmon*/power*_cap", "hwmon/hwmon*/temp*_input",
                    "gpu_busy_percent", "current_link_speed"):
            out += sorted(glob.glob(os.path.join(card, pat)))
    return out


def read(p):
    try:
        return open(p).read().strip().replace("\n", " | ")
    except Exception as e:
        return "<%s>" % type(e).__name__


def main():
    fs = files()
    print("sysfs files found: %d" % len(fs))
    for p in fs:
        print("  idle  %-70s %s" % (p, read(p)))
    os.system("rocm-smi --showclocks --showpower --showperflevel 2>&1 | head -40")
    watch = [p for p in fs if ("freq" in p and "input" in p) or "power" in p and "cap" not in p or p.endswith("pp_dpm_sclk")]
    img = torch.rand((4096, 4096), device="cuda")
    f = cv.SteerableFiltersG2(img, 4, 0.67)
    f4 = cv.SteerableFiltersG4(img, 6, 0.5)
    for name, fn in (("pipeline", lambda: f.pipeline(img)), ("setup", lambda: f.setup(img)), ("g4 setup", lambda: f4.setup(img))):
        samples, stop = {p: [] for p in watch}, [False]

        def sampler():
            while not stop[0]:
                for p in watch:
                    samples[p].append(read(p))
                time.sleep(0.05)
        for _ in range(60):
            fn()
        torch.cuda.synchronize()
        th = threading.Thread(target=sampler)
        th.start()
        t0 = time.time()
        n = 0
        while time.time() - t0 < 3.0:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
            n += 50
        dt = time.time() - t0
        stop[0] = True
        th.join()
        print("== %s: %d calls in %.2f s = %.4f ms per call" % (name, n, dt, 1e3 * dt / n))
        for p in watch:
            v = samples[p]
            print("  load  %-70s n=%d first %s | mid %s | last %s" % (p, len(v), v[0] if v else "", v[len(v) // 2] if v else "", v[-1] if v else ""))


if __name__ == "__main__":
    main()
