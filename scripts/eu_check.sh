# GPU-box helper: Euclidean single-query scans with the handle's initial select plan either way
set -e
B="timeout -k 10 300 python bench.py --no-cpu-baseline --no-pmc --metric euclidean"
for c in "--rows 10000 --steps 1000 --warmup 20" "--rows 100000 --steps 500 --warmup 20" "--rows 10000 --steps 1000 --warmup 20 --host"; do
  for e in 0 1 0 1; do
    if [ $e = 1 ]; then export TRAG_EUCLID_ONE_WG=1; else unset TRAG_EUCLID_ONE_WG; fi
    $B $c > gpurun_out/eu.json 2> gpurun_out/eu.err
    python3 -c "
import json; r=json.load(open('gpurun_out/eu.json')); print('one_wg=$e', '$c', 'ms/step %.4f scan %.4f' % (r['ms_per_step'], r['roofline']['kernel_avg_ms']), r['parity']['uncertified_queries'], r['parity']['exact_scan_keys_kept_from'])"
  done
done
