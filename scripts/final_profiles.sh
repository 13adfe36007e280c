#!/bin/bash
# The round's profile set in one go (on the GPU box): the bench line, the same command under the kernel trace, the counter
# passes of one 16-spp batch.  usage: scripts/final_profiles.sh <tag>  (files land in gpurun_out/<tag>_*; then, where the
# repository is: python scripts/update_profiles.py <tag> <prefix in profiles/>)
export TMPDIR=/tmp
set -e
T=${1:-r03}
timeout -k 10 500 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
rm -rf gpurun_out/${T}_kstats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_kstats -- python3 bench.py --no-cpu-baseline --no-configs --no-live-traffic > gpurun_out/${T}_bench_under_rocprof.json 2> gpurun_out/${T}_rocprof.err
SPP=16 bash scripts/pmc_passes.sh ${T}
python scripts/pmc_summary.py ${T} > gpurun_out/${T}_pmc_summary.log 2>&1 || true
python scripts/pmc_by_dispatch.py ${T} k_extend > gpurun_out/${T}_extend_by_bounce.txt 2>&1 || true
find gpurun_out/${T}_kstats -name "*kernel_stats.csv" | head -3
