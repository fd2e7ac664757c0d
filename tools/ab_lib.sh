#!/bin/bash
# tools/ab_lib.sh OUT.txt N LIB_A LIB_B -- bench.py (no CPU baseline, no counter passes) alternately on two builds of the library
# (CVSTEER_HIP_LIB picks the twin), N rounds; prints the legs' fractions side by side.  One box, one call.
out=$1; n=$2; a=$3; b=$4
: > $out
for i in $(seq 1 $n); do
  for lib in $a $b; do
    echo "== $lib round $i" >> $out
    CVSTEER_HIP_LIB=$lib timeout -k 10 300 python3 bench.py --no-cpu --no-live-traffic --steps 20 --warmup 5 >> $out 2>> ${out%.txt}.err || exit 1
  done
done
python3 - "$out" <<'PY'
import json,sys,collections
rows=collections.OrderedDict(); lib=None
for l in open(sys.argv[1]):
    if l.startswith("== "): lib=l.split()[1]; continue
    if not l.startswith("{"): continue
    d=json.loads(l)
    rows.setdefault(("headline_M2",lib),[]).append(d["roofline"]["frac"])
    for k,v in d.get("legs",{}).items():
        if v[0] is not None: rows.setdefault((k,lib),[]).append(v[0])
names=[]
for (k,lib) in rows:
    if k not in names: names.append(k)
libs=sorted({lib for (_,lib) in rows})
print("leg".ljust(44)+"".join(l.split("/")[-1].ljust(40) for l in libs))
for k in names:
    print(k.ljust(44)+"".join((" ".join("%.3f"%x for x in rows.get((k,l),[]))).ljust(40) for l in libs))
PY
