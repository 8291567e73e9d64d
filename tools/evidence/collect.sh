# here, after the evidence passes: copy what the bench line and DESIGN section 8 cite from gpurun_out/<round>/ into profiles/
# (tracked), named per round; the two PMC summaries bench.py reads keep their fixed names.
R=${1:-r06}; O=gpurun_out/$R
for f in $O/*.json $O/*.txt $O/*.md $O/gpu_tests.log $O/smoke.log; do
  [ -f "$f" ] || continue
  b=$(basename $f)
  case $b in pmc_traffic.json|pmc_util.json) cp $f profiles/$b ;; esac
  cp $f profiles/${R}_$b
done
ls profiles | grep "^${R}_" | wc -l
