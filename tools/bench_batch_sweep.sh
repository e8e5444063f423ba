mkdir -p gpurun_out/r2c
for B in 8192 32768 65536 131072; do
  python bench.py --no-dense-blocks --no-cpu-baseline --batch $B > gpurun_out/r2c/bench_B$B.json 2> gpurun_out/r2c/bench_B$B.err
  python -c "
import json; d=json.load(open('gpurun_out/r2c/bench_B$B.json')); print($B, 'P', d['time_partitions'], 'value', round(d['value']), 'ms/step', round(d['ms_per_step'],2), 'fact/it', d['factorizations_per_iteration']); print('   ', d['roofline']['kernel_ms_per_iteration'])"
done
