# A/B on one box: libraries (product | tools/dev/variants/libseirhip_<v>.so) in turn, three rounds:  bash tools/dev/ab_many.sh product prio3 ...
mkdir -p gpurun_out/r04
out=gpurun_out/r04/ab_many.txt; : > $out
for rep in 0 1 2; do
  for v in "$@"; do
    timeout -k 10 300 python tools/dev/leap_variant.py $v ${AB_CHAINS:-8} ${AB_SWEEPS:-300} ${AB_WORKLOAD:-uk380} 2>/dev/null >> $out
  done
done
cat $out
