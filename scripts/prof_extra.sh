#!/bin/bash
# GPU box: kernel tables beyond the headline configuration -- EPC-Net-L at batch 256 (configs[3]) and the bf16 training step.
export TMPDIR=/tmp
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out/extra
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/extra/l -- python3 $R/bench.py --arch epc-net-l --batch 256 --precision f32 --no-cpu-baseline --no-configs --no-rccl --in-flight 1 --regions 1 --steps 20 --warmup 5 > $R/gpurun_out/extra/l.log 2>&1
cp $(ls $R/gpurun_out/extra/l/*/*kernel_stats.csv | head -1) $R/gpurun_out/extra/epc_net_l_b256_kernel_stats.csv
cd /tmp && GRAPH=0 PRECISION=bf16 rocprofv3 --kernel-trace -d $R/gpurun_out/extra/prof_train -o run -- python3 $R/scripts/time_train_step.py > $R/gpurun_out/extra/prof_train.log 2>&1
cd $R
python scripts/rocpd_stats.py $(ls gpurun_out/extra/prof_train/*.db | head -1) 90 70 > gpurun_out/extra/train_bf16_stats.txt
rm -rf gpurun_out/extra/prof_train gpurun_out/extra/l
head -12 gpurun_out/extra/epc_net_l_b256_kernel_stats.csv | cut -c1-160; head -6 gpurun_out/extra/train_bf16_stats.txt
