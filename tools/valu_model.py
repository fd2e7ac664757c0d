#!/usr/bin/env python3
"""tools/valu_model.py FILE.s [pattern ...] -- vector-ALU cycles of the strip kernels from their ISA, with the issue costs measured by
tools/valu_rate.hip on MI355X (profiles/r05_valu_rate.txt), per wave64 instruction and SIMD:
  2 cycles   v_fma / v_fmac / v_add / v_sub / v_mul (f32), v_mov_b32, v_add_u32 / v_sub_u32, v_and / v_or / v_xor, v_fmaak / v_fmamk -- when no
             operand is a scalar register (literals and inline constants are free)
  8 cycles   v_rcp / v_sqrt / v_rsq / v_exp / v_log / v_sin / v_cos
  4 cycles   everything else: any scalar-register operand (taps!), v_pk_*, v_cmp*, v_cndmask, v_cvt*, v_min / v_max, v_rndne, v_ldexp, v_bfi,
             v_div_*, shifts, v_readlane / v_writelane
Prints per kernel: instructions and model cycles of the whole body (window priming group + one steady group), split by class."""
import re, subprocess, sys, collections
TWO = {"v_fma_f32", "v_fmac_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
       "v_and_b32", "v_or_b32", "v_xor_b32", "v_fmaak_f32", "v_fmamk_f32"}
EIGHT = ("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")
def classify(op, args):
    base = re.sub(r"_e32$|_e64$|_sdwa$|_dpp$", "", op)
    if base.startswith(EIGHT): return "trans", 8
    if base.startswith("v_pk_"): return "packed", 4
    if base.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "spill/lane", 4
    sg = bool(re.search(r"(?<![\w.])s\d+|s\[\d+:\d+\]|vcc|exec|m0", args))
    if base in TWO and not sg: return "simple", 2
    if base in TWO: return "simple+sgpr", 4
    return "other", 4
def main():
    text = open(sys.argv[1]).read()
    pats = sys.argv[2:]
    names = re.findall(r"^(_ZN3cvs\S+):", text, re.M)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    for n, d in zip(names, dem):
        d = re.sub(r"\(cvs::BasisArgs.*", "", d).replace("void cvs::", "")
        if pats and not any(p in d for p in pats): continue
        m = re.search(r"^" + re.escape(n) + r":[^\n]*\n(.*?)s_endpgm", text, re.S | re.M)
        if not m: continue
        cyc = collections.Counter(); cnt = collections.Counter()
        for l in m.group(1).split("\n"):
            mm = re.match(r"\s+(v_\w+)\s*(.*)", l)
            if not mm: continue
            c, k = classify(mm.group(1), mm.group(2))
            cyc[c] += k; cnt[c] += 1
        print("%-70s instr %5d cycles %6d | %s" % (d[:70], sum(cnt.values()), sum(cyc.values()), "  ".join("%s %d/%d" % (c, cnt[c], cyc[c]) for c in sorted(cyc))))
main()
