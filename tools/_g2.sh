cd /root/repo
mkdir -p gpurun_out/r03
python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r03/t_all.log
OUT=gpurun_out/r03/sweep1.txt; : > $OUT
run() { echo "## $*" >> $OUT; timeout 600 python bench.py --no-cpu-baseline --no-probe --no-traffic --steps 8 --warmup 2 "$@" 2>>gpurun_out/r03/sweep1.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; a=d.get('adjoint',{})
print(json.dumps({'sched':d['config'].get('schedule'),'kernel_ms':round(r['kernel_ms'],3),'frac':round(r['frac'],4),'adjoint_ms':round(a.get('kernel_ms',0),3),'adjoint_frac':round(a.get('frac',0),4)}))" >> $OUT 2>&1; }
for lib in w6 w5 w6 w5; do export H2GCN_HIP_LIBRARY=$PWD/build/ab/lib_$lib.so; echo "# lib $lib" >> $OUT; run --d 100; run --d 200 --variant 4; run --shape arxiv --d 100; done
unset H2GCN_HIP_LIBRARY
run --shape arxiv --d 1433
run --shape arxiv --d 1433 --variant 4
run --shape arxiv --d 1433 --slice-cols 128
run --shape arxiv --d 1433 --slice-cols 256
run --shape arxiv --d 3703
run --d 130
run --d 130 --variant 4
run --d 200
run --d 256
run --shape arxiv
run --shape lowdeg
run --shape lowdeg --d 64
tail -3 gpurun_out/r03/t_all.log; cat $OUT
