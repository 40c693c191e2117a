cd /root/repo
mkdir -p gpurun_out/r03
python -m pytest tests -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r03/t_all2.log
tail -5 gpurun_out/r03/t_all2.log
