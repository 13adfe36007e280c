# A/B of all configurations on one box (development aid): bench.py with the library as built and with the variant given
# as the first argument (default variants/librt_amd_slp.so), two rounds
V=${1:-$PWD/variants/librt_amd_slp.so}
for i in 1 2; do for lib in "" "$V"; do
RT_AMD_LIBRARY=$lib timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib'[-12:] or 'current', d['value'], {k[:2]:(v['msamples_per_s'], v['ms_per_call']) for k,v in d['configs'].items()})" || exit 1
done; done
