#!/bin/bash
# the three single-GPU bench lines with their CPU legs (gpurun_out/bench_config{2,3,4}.json)
O=gpurun_out; mkdir -p $O
timeout 600 python bench.py --steps ${STEPS2:-100} > $O/bench_config2.json 2> $O/bench_config2.err
timeout 400 python bench.py --config 3 --steps 40 > $O/bench_config3.json 2> $O/bench_config3.err
timeout 400 python bench.py --config 4 --steps 20 > $O/bench_config4.json 2> $O/bench_config4.err
for c in 2 3 4; do python - $O/bench_config$c.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d['config']['baseline_config'], round(d['value']), d['ms_per_step'], round(d['roofline']['frac'],4), 'parity', d.get('parity_check'))
    for k in ('cpu_baseline','cpu_baseline_numpy','cpu_baseline_numpy_socket','end_to_end','pointwise_end_to_end'):
        v=d.get(k)
        if v: print('   ', k, round(v['value'],1) if v.get('value') else v, v.get('cores'), (v.get('sample') or v.get('path') or v.get('error') or '')[:100])
except Exception as e:
    print('FAILED', sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
