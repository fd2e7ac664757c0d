#!/usr/bin/env python3
"""tools/isa_summary.py -- compile the kernel translation units to gfx950 assembly (no GPU needed) and
tabulate, per kernel: VGPRs, SGPRs, scratch, LDS, waves/SIMD, and the instruction mix of the body
(VALU / SALU / LDS / buffer loads / buffer stores / v_pk_* / v_readlane+v_writelane SGPR-spill traffic).
Writes profiles/<round>_isa_summary.csv, and gpurun_out/valu_cpi.json (tools/valu_model.py: model cycles per vector instruction per kernel)."""
import csv, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
         "-I" + os.path.join(ROOT, "cvsteer_amd", "csrc"), "-S", "--cuda-device-only"]
FLAGS += os.environ.get("ISA_EXTRA", "").split()   # e.g. -DCVS_G2_SRED_MASK=15 for a what-if table
UNITS = [("cvs_kernels_basis.hip", ["-fno-slp-vectorize"]), ("cvs_kernels_point.hip", [])]

def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\((cvs::PointArgs|cvs::BasisArgs|float const\*|float\*|unsigned char|int\*).*", "", n).replace("void ", "") for n in out]

def waves(v):
    alloc = (int(v) + 7) // 8 * 8
    return min(8, 512 // max(alloc, 1))

rows = []
for src, extra in UNITS:
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + [os.path.join(ROOT, "cvsteer_amd", "csrc", src), "-o", tmp.name],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(tmp.name).read()
        if src == "cvs_kernels_basis.hip":   # cycles per vector instruction of every strip kernel's mix, for tools/collect_valu.py
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "valu_model.py"), tmp.name, "--json", os.path.join(ROOT, "gpurun_out", "valu_cpi.json")],
                           check=True, stdout=subprocess.DEVNULL)
    meta = {}
    for blk in text.split("  - .agpr_count:")[1:]:   # one chunk of the amdhsa.kernels metadata per kernel
        nm = re.search(r"\.name:\s+(\S+)", blk)
        if not nm:
            continue
        get = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1)) if re.search(r"\." + k + r":\s+(\d+)", blk) else 0
        meta[nm.group(1)] = (get("vgpr_count"), get("sgpr_count"), get("private_segment_fixed_size"), get("group_segment_fixed_size"))
    names = list(meta)
    for name, pretty in zip(names, demangle(names)):
        body = re.search(r"^" + re.escape(name) + r":[^\n]*\n(.*?)s_endpgm", text, re.S | re.M)
        b = body.group(1) if body else ""
        cnt = lambda pat: len(re.findall(pat, b, re.M))
        v, s, scr, lds = meta[name]
        rows.append([pretty, v, s, scr, lds, waves(v), cnt(r"^\s+v_"), cnt(r"^\s+s_"), cnt(r"^\s+ds_"), cnt(r"^\s+buffer_load|^\s+global_load"),
                     cnt(r"^\s+buffer_store|^\s+global_store"), cnt(r"^\s+v_pk_"), cnt(r"^\s+v_readlane|^\s+v_writelane"), cnt(r"^\s+s_waitcnt")])
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
path = os.path.join(ROOT, "profiles", "%s_isa_summary.csv" % rnd)
with open(path, "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "vgpr", "sgpr", "scratch_bytes", "lds_bytes", "waves_per_simd", "valu", "salu", "lds_ops", "vmem_loads", "vmem_stores",
                "packed_f32", "sgpr_spill_lane_ops", "waitcnt"])
    w.writerows(sorted(rows))
print(open(path).read())
