export TMPDIR=/tmp
set -e
timeout -k 10 400 python bench.py > gpurun_out/r02d_bench.json 2> gpurun_out/r02d_bench.err
rm -rf gpurun_out/r02d_kstats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02d_kstats -- python3 bench.py --no-cpu-baseline --no-configs > gpurun_out/r02d_bench_under_rocprof.json 2> gpurun_out/r02d_rocprof.err
SPP=16 bash scripts/pmc_passes.sh r02d
python scripts/pmc_summary.py r02d > gpurun_out/r02d_pmc_summary.log 2>&1 || true
python scripts/pmc_by_dispatch.py r02d k_extend > gpurun_out/r02d_extend_by_bounce.txt 2>&1 || true
find gpurun_out/r02d_kstats -name "*kernel_stats.csv" | head -3
