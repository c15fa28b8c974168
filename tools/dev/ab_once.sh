mkdir -p gpurun_out/r04
for v in "$@"; do timeout -k 10 300 python tools/dev/leap_variant.py $v 8 30 syn2048 2>/dev/null | cut -c1-120; done
