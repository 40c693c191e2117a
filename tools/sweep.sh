#!/bin/bash
# tools/sweep.sh <outfile> -- one bench line per tuning point (GPU box)
OUT=$1; : > $OUT
run() { echo "## $*" >> $OUT; timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:d[k] for k in ('value','ms_per_step')}|{'kernel_ms':d['roofline']['kernel_ms'],'frac':d['roofline']['frac']}))" >> $OUT 2>&1; }
run --slice-cols 128
run --slice-cols 64
run --slice-cols 32
run --slice-cols 0
run --slice-cols 128 --chunks 2
run --slice-cols 128 --chunks 4
run --variant 1
run --slice-cols 64 --rows-per-wave 1
run --slice-cols 64 --rows-per-wave 2
run --slice-cols 64 --long-row-threshold 256
run --shape arxiv
run --d 64
run --d 64 --slice-cols 32
run --d 256
run --d 256 --slice-cols 64
cat $OUT
