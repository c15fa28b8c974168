# A/B on one box: the product library against tools/dev/variants/libseirhip_<v>.so (default: base, built from HEAD), alternating
mkdir -p gpurun_out/r04
out=gpurun_out/r04/ab_base.txt; : > $out
for rep in 0 1 2; do
  for v in ${1:-base} product; do
    timeout -k 10 200 python tools/dev/leap_variant.py $v 8 300 2>/dev/null >> $out
  done
done
cat $out
