cd /root/repo
mkdir -p gpurun_out/r03
python -m pytest tests/test_multirank_gpu.py -x -q -m gpu -k "dry_exchange or bench_multi_rank" 2>&1 | tail -15 > gpurun_out/r03/t_dry.log
(time python bench.py) > gpurun_out/r03/bench_default2.json 2> gpurun_out/r03/bench_default2.err
tail -4 gpurun_out/r03/t_dry.log; tail -4 gpurun_out/r03/bench_default2.err
python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r03/bench_default2.json') if l.startswith('{')][-1])
r=d['roofline']
print({k:r[k] for k in ('achieved','frac','kernel_ms','traffic_over_algorithmic','gather_ceiling_GBps','hbm_resident_frac') if k in r})
print(r.get('hbm_resident'))
print(d.get('cpu_baseline',{}).get('value'))
PY
