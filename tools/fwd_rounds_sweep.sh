# Sequential sweep: inertia-correction rounds per launch (DTO_FWD_ROUNDS) x batch size, headline loop and full solves.
# Run on the GPU box from the repo root; results under gpurun_out/r2d/
mkdir -p gpurun_out/r2d
for B in 131072 196608 262144; do
  for R in 0 1 2 3; do
    DTO_FWD_ROUNDS=$R python bench.py --no-dense-blocks --no-cpu-baseline --batch $B > gpurun_out/r2d/bench_R${R}_B$B.json 2> gpurun_out/r2d/bench_R${R}_B$B.err
    python -c "
import json; d=json.load(open('gpurun_out/r2d/bench_R${R}_B$B.json')); r=d['roofline']['kernel_ms_per_iteration']; print('B', $B, 'rounds/launch', $R, 'value', round(d['value']), 'it thr', round(d['iteration_throughput']['value']), 'conv', d['solve']['converged'], 'fwd ms', r['k_kkt_fwd_seq'], 'bwd', r['k_kkt_bwd_seq'])"
  done
done
