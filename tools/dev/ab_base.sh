# A/B on one box: the product library against tools/dev/variants/libseirhip_base.so (built from HEAD), alternating
mkdir -p gpurun_out/r04
out=gpurun_out/r04/ab_base.txt; : > $out
for rep in 0 1 2; do
  for v in base product; do
    if [ $v = base ]; then a="--variant base"; else a=""; fi
    echo "== $v rep$rep" >> $out
    timeout -k 10 200 python tools/quick_sweep_bench.py --groups 1 --sweeps 300 $a 2>/dev/null | grep -E "ms_per_sweep" >> $out
  done
done
cat $out
