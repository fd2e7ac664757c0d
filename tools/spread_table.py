"""tools/spread_table.py [FILE.jsonl] -- bench.py lines (default profiles/r06_bench_lines.jsonl, one run per line) -> the min / median / max table of
DESIGN.md section 6 (last column: the four runs of round 5's final code, profiles/r05_bench_lines.jsonl)."""
import json,os,statistics,sys
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
runs=[json.loads(l) for l in open(sys.argv[1] if len(sys.argv)>1 else os.path.join(ROOT,'profiles','r06_bench_lines.jsonl')) if l.startswith('{')]
earlier=os.path.join(ROOT,'profiles','r06_bench_lines_earlier_in_the_round.jsonl')
allruns=runs+([json.loads(l) for l in open(earlier) if l.startswith('{')] if os.path.exists(earlier) else [])
rows=[("**M2 filter + steer (headline)**",40,"headline","0.810 / 0.827 / 0.833"),
("**M1 basis pass (north_star ≥ 0.70)**",32,"M1_basis","0.786 / 0.805 / 0.816"),
("M4 full setup",52,"M4_full_setup","0.783 / 0.803 / 0.810"),
("M5 caller pipeline, state kept",84,"M5_pipeline","0.747 / 0.772 / 0.787"),
("M2, 8 rotating images",40,"M2_fresh_8_rotating","0.700 / 0.725 / 0.747"),
("M2, one object per image, no host sync",40,"M2_one_object_per_image","0.705 / 0.718 / 0.739"),
("M2, first call of a handle, synchronised",40,"M2_first_call_synchronised","—"),
("M6 G4 basis",48,"M6_g4_basis","0.652 / 0.657 / 0.668"),
("M6 G4 + scalar steer",56,"M6_g4_filter_steer","0.642 / 0.686 / 0.697"),
("M1 at 8192² (resident)",32,"M1_basis_8192","0.826–0.841"),
("M2 at 8192² (resident)",40,"M2_filter_steer_8192","0.855–0.880"),
("M2 at 8192², two images alternating",40,"M2_filter_steer_8192_fresh_2_rotating","0.694 / 0.708 / 0.718"),
("C4 32 × 1080p pipeline, state kept",84,"C4_32x1080p_pipeline_state_kept","0.657 / 0.740 / 0.763"),
("C4 32 × 1080p, three maps only",16,"C4_32x1080p_three_maps_only","0.292–0.304 (146–152 Gpix/s)"),
("C3 pyramid of 8192², 5 levels, whole","—","C3_pyramid_8192_5_levels_whole","0.656 / 0.665 / 0.675"),
("M5 / G4 after 30 ms of idleness",84,"M5_pipeline_after_idle","0.606–0.647 / 0.564–0.648"),]
prev=[json.loads(l) for l in open(os.path.join(ROOT,'profiles','r05_bench_lines.jsonl')) if l.startswith('{')]
def fmt(k, rs=None):
    vals=sorted((r["roofline"]["frac"] if k=="headline" else (r["legs"].get(k) or [None])[0]) for r in (rs or runs))
    vals=[v for v in vals if v]
    return "%.3f / %.3f / %.3f"%(vals[0],statistics.median(vals),vals[-1])
print("| leg | B/pix | round 6, final code, %d runs (min / median / max) | all %d runs of round 6 | round 5, final code (4 runs) |"%(len(runs),len(allruns)))
print("|---|---|---|---|---|")
for name,b,k,r4 in rows:
    if k=="M5_pipeline_after_idle":
        print("| %s | %s | %s and %s | %s and %s | %s and %s |"%(name,"84 / 48",fmt(k),fmt("M6_g4_basis_after_idle"),fmt(k,allruns),fmt("M6_g4_basis_after_idle",allruns),fmt(k,prev),fmt("M6_g4_basis_after_idle",prev)))
    else:
        print("| %s | %s | %s | %s | %s |"%(name,b,("**%s**"%fmt(k)) if name.startswith("**") else fmt(k),fmt(k,allruns),fmt(k,prev)))
