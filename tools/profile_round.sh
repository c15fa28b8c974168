#!/bin/bash
# Collect the round's profile evidence on the GPU box (run from the repo root through gpurun):
#   bash tools/profile_round.sh r01
# 1. bench.py under rocprofv3 --kernel-trace --stats (per-kernel durations of the timed region)
# 2. PMC passes, one counter each and nothing else (FETCH_SIZE, WRITE_SIZE) over a short sampler run
# 2b. MFMA-utilisation counters (SQ_INSTS_VALU_MFMA_MOPS_F64, SQ_VALU_MFMA_BUSY_CYCLES, ...) over the stateless evaluation
# 3. the same FETCH_SIZE pass over the load-only probe with known traffic (calibrates gfx950's unit)
# Raw output goes to gpurun_out/; tools/summarize_profiles.py condenses it into profiles/.
set -e
tag=${1:-r01}
export TMPDIR=/tmp
out=gpurun_out/${tag}
rm -rf "$out"; mkdir -p "$out"
python3 bench.py --steps 200 --warmup 20 > "$out/bench.json" 2> "$out/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o k -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-chains-scaling --no-cli --no-egress \
    > "$out/bench_under_rocprof.json" 2> "$out/stats.err"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$out/pmc_$c" -o k -- python3 tools/quick_sweep_bench.py --groups 1 --sweeps 20 \
      > "$out/pmc_$c.log" 2>&1
done
# 2b. matrix-core counters of the stateless evaluation (k_gemm): SQ counters only, in a pass of their own
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES \
    --output-format csv -d "$out/pmc_mfma" -o k -- python3 tools/quick_eval_bench.py --iters 10 > "$out/pmc_mfma.log" 2>&1 \
  || echo "MFMA counter pass failed (see $out/pmc_mfma.log): counter names on this ROCm are listed in $out/counters.txt"
rocprofv3 -L > "$out/counters.txt" 2>&1 || true
if [ -x tools/probes/se_probe ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_probe" -o k -- tools/probes/se_probe > "$out/pmc_probe.log" 2>&1
fi
# 4. BASELINE config 5 (SYN-2048 x 730, 8 chains): the bench line, kernel stats of a short sampler run, and the matrix-core
#    counters of the fp32 and fp64 contraction kernels (SQ counters only, in passes of their own)
python3 bench.py --workload syn2048 --steps 20 --warmup 3 --adapt-sweeps 10 > "$out/bench_syn2048.json" 2> "$out/bench_syn2048.err" \
  || echo "syn2048 bench failed (see $out/bench_syn2048.err)"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_syn2048" -o k -- python3 tools/quick_sweep_bench.py --workload syn2048 --groups 1 --sweeps 10 \
    > "$out/stats_syn2048.log" 2>&1 || echo "syn2048 kernel stats failed"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES \
    --output-format csv -d "$out/pmc_mfma_syn_f32" -o k -- python3 tools/quick_eval_bench.py --workload syn2048 --iters 3 --form four-launch --gemm-f32 1 \
    > "$out/pmc_mfma_syn_f32.log" 2>&1 || echo "syn2048 fp32 MFMA counter pass failed"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES \
    --output-format csv -d "$out/pmc_mfma_syn_f64" -o k -- python3 tools/quick_eval_bench.py --workload syn2048 --iters 3 --form four-launch \
    > "$out/pmc_mfma_syn_f64.log" 2>&1 || echo "syn2048 fp64 MFMA counter pass failed"
echo "raw output in $out; now run: python3 tools/summarize_profiles.py $tag $out  (here or after gpurun merged gpurun_out/ back)"
