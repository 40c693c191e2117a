cd /root/repo
mkdir -p gpurun_out/r03
bash tools/profile_bench.sh r03_arxiv_d1433_s64 --shape arxiv --d 1433 --steps 10 --warmup 2 > /dev/null 2>&1
bash tools/profile_bench.sh r03_arxiv_d1433_s128 --shape arxiv --d 1433 --slice-cols 128 --steps 10 --warmup 2 > /dev/null 2>&1
for t in r03_arxiv_d1433_s64 r03_arxiv_d1433_s128; do echo "== $t"; f=$(find gpurun_out/prof_$t/trace -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-220; done > gpurun_out/r03/prof_1433.txt
OUT=gpurun_out/r03/sweep2.txt; : > $OUT
run() { echo "## $*" >> $OUT; timeout 600 python bench.py --no-cpu-baseline --no-probe --no-traffic --steps 8 --warmup 2 "$@" 2>>gpurun_out/r03/sweep2.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; a=d.get('adjoint',{})
print(json.dumps({'sched':d['config'].get('schedule'),'kernel_ms':round(r['kernel_ms'],3),'frac':round(r['frac'],4),'adjoint_ms':round(a.get('kernel_ms',0),3),'adjoint_frac':round(a.get('frac',0),4)}))" >> $OUT 2>&1; }
run --d 130
run --d 200
run --d 200 --variant 4
run --d 300
run --d 300 --variant 4
run --d 132
run --shape arxiv --d 1433 --slice-cols 128
run --shape products --d 1433 --shape arxiv --slice-cols 64
cat gpurun_out/r03/prof_1433.txt; cat $OUT
