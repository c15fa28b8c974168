set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_sampler_gpu.py tests/test_invariance_gpu.py tests/test_recovery_gpu.py -q -m gpu > gpurun_out/r04/gpu_fold2.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r04/gpu_fold2.log | cut -c1-200
for rep in 0 1; do
for cfg in "uk380 16" "uk380 32" "uk380 64" "syn2048 8"; do
  set -- $cfg
  for hmc in chunk-launch chunk-launch-fold; do
    echo "== $1 x$2 $hmc rep$rep"
    timeout -k 10 200 python tools/quick_sweep_bench.py --workload $1 --chains $2 --sweeps 100 --groups 1 --hmc $hmc --moves paired-launch 2>/dev/null | grep -E "ms_per_sweep|samples_per_s"
  done
done
done > gpurun_out/r04/fold_ab.txt 2>&1
cat gpurun_out/r04/fold_ab.txt
