cd /root/repo
mkdir -p gpurun_out/r03
for h in 64 100 128; do for mode in "" "--stock-classifier"; do python tools/epoch_products.py $h $mode 2>&1 | grep "train step" | sed "s/^/[${mode:-fused classifier}] /"; done; done > gpurun_out/r03/train_step_fused_vs_stock.txt
cat gpurun_out/r03/train_step_fused_vs_stock.txt
for h in 64 100 128; do bash tools/profile_train_step.sh r03_train_h$h $h > gpurun_out/r03/train_h$h.txt 2>&1; done
