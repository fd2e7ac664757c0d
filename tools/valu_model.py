#!/usr/bin/env python3
"""tools/valu_model.py FILE.s [pattern ...] -- vector-ALU time of the strip kernels from their ISA, with the issue costs measured by
tools/valu_rate.hip on MI355X (profiles/r05_valu_rate.txt).  A SIMD behaves like two 16-lane pipes that take one wave64 instruction
every ~4.4 cycles each:
  simple     v_fma / v_fmac / v_add / v_sub / v_mul (f32), v_mov_b32, v_add_u32 / v_sub_u32, v_and / v_or / v_xor, v_fmaak / v_fmamk with no
             scalar-register operand (literals and inline constants are free): either pipe -- two of them go through per ~4.6 cycles
  other      any scalar-register operand (the filter taps!), v_cmp*, v_cndmask, v_cvt*, v_min / v_max, v_rndne, v_ldexp, v_bfi, v_div_*,
             shifts, v_readlane / v_writelane: one per ~4.4 cycles, the second pipe stays free for another wave's simple instructions
  packed     v_pk_fma / v_pk_mul / v_pk_add (f32): two results per lane, both pipes for ~4.4 cycles
  trans      v_rcp / v_sqrt / v_rsq / ...: ~8.2 cycles (counted as two `other`)
Lower bound of the vector time of a mix: 4.4 cycles x max(packed + other, (2 packed + other + simple) / 2).
Prints per kernel: instructions by class (whole body = window priming group + one steady group), that bound, and the bound per instruction;
with --json FILE writes {kernel: cycles per instruction} for tools/collect_valu.py."""
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
    argv = sys.argv[1:]
    jpath = None
    if "--json" in argv:
        k = argv.index("--json"); jpath = argv[k + 1]; del argv[k:k + 2]
    text = open(argv[0]).read()
    pats = argv[1:]
    table = {}
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
        P = cnt["packed"]; S = cnt["simple"]; N = sum(cnt.values()) - P - S + cnt["trans"]
        bound = 4.4 * max(P + N, (2 * P + N + S) / 2.0)
        cpi = bound / max(1, sum(cnt.values()))
        table[d] = round(cpi, 3)
        print("%-70s instr %5d  bound %6.0f cycles  %.2f cycles/instr | %s" % (d[:70], sum(cnt.values()), bound, cpi, "  ".join("%s %d" % (c, cnt[c]) for c in sorted(cnt))))
    if jpath:
        import json
        json.dump(table, open(jpath, "w"), indent=0, sort_keys=True)
main()
