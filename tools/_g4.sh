cd /root/repo
mkdir -p gpurun_out/r03
python -m pytest tests/test_spmm_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r03/t_spmm2.log
OUT=gpurun_out/r03/sweep3.txt; : > $OUT
run() { echo "## $*" >> $OUT; timeout 600 python bench.py --no-cpu-baseline --no-probe --no-traffic --steps 8 --warmup 2 "$@" 2>>gpurun_out/r03/sweep3.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; a=d.get('adjoint',{})
print(json.dumps({'sched':d['config'].get('schedule'),'kernel_ms':round(r['kernel_ms'],3),'frac':round(r['frac'],4),'adjoint_ms':round(a.get('kernel_ms',0),3),'adjoint_frac':round(a.get('frac',0),4)}))" >> $OUT 2>&1; }
run
run --shape arxiv --d 1433
run --shape arxiv --d 1433 --slice-cols 64
run --shape arxiv --d 1433 --slice-cols 256
run --shape arxiv --d 3703
run --d 130
run --d 130 --variant 4
run --d 102
run --d 100
run --d 200
run --d 300
run --shape arxiv
run --shape lowdeg
run --shape lowdeg --d 64
run --shape hbm16m
tail -3 gpurun_out/r03/t_spmm2.log; cat $OUT
