# probe (needs tools/patches/tapered_strips_probe.patch applied): a shorter last round of strips (CVS_TAPER=percent:rows) against the uniform launch, same box, bench legs
for T in off 75:10 85:10 90:10 off 60:10; do
  echo "== taper $T"
  if [ $T = off ]; then unset CVS_TAPER; else export CVS_TAPER=$T; fi
  python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']
print('headline %.4f  plain %.4f rot %.4f untuned %.4f first %.4f  M1 %.4f M4 %.4f M5 %.4f' % (d['roofline']['frac'], e['M2_plain_block']['frac_hbm'], e['M2_rotating_8_inputs']['frac_hbm'], e['M2_untuned']['frac_hbm'], e['M2_first_call']['frac_hbm'], e['M1_basis_only']['frac_hbm'], e['M4_full_setup']['frac_hbm'], e['M5_pipeline']['frac_hbm']))"
done
