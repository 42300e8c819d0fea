#!/bin/bash
# GPU box: the round's final collection on a CLEAN build of the library -- the GPU tests (stop on failure), the headline profile set,
# the EPC-Net-L batch-256 side set, the training step's counters, the stress parity tables.
#   scripts/final_collect.sh <tag>      e.g. r05_b
# bench.py reads the counter summaries only while the loaded library has the hash stamped into them, and the driver's box runs the
# library as built from a clean tree: collect AFTER `make -B -C epc-net_amd/csrc` and do not rebuild afterwards.
set -eu
tag=${1:?usage: final_collect.sh <tag>}
cd "$(dirname "$0")/.."
out=gpurun_out/final_$tag
mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1 || { tail -30 $out/gpu_tests.log; echo "GPU tests failed: nothing collected"; exit 1; }
grep -a "passed\|failed" $out/gpu_tests.log | tail -2
bash scripts/collect_profiles.sh $tag > $out/collect.log 2>&1
bash scripts/collect_profiles.sh ${tag}_l_b256 side --arch epc-net-l --batch 256 > $out/collect_l.log 2>&1
bash scripts/collect_train_profiles.sh $tag > $out/collect_train.log 2>&1
PRECISION=bf16 bash scripts/prof_train.sh final_$tag && cp gpurun_out/final_$tag/train_stats.txt profiles/${tag}_train_step_bf16_kernel_stats.txt
bash scripts/prof_train.sh final_$tag && cp gpurun_out/final_$tag/train_stats.txt profiles/${tag}_train_step_kernel_stats.txt
PRECISION=f32 python scripts/stress_parity.py 90 > profiles/${tag}_stress_parity.txt 2>&1 || true
tail -3 profiles/${tag}_stress_parity.txt
# the tracked bench line LAST: only now do the counter summaries carry this library's hash, so the line has `traffic` / `mfma_util`
# (the line collect_profiles.sh wrote at its start was taken before the summaries were stamped: VERDICT r5 weak #9)
python3 bench.py > profiles/${tag}_bench_line.json 2> $out/bench_final.err || { tail -5 $out/bench_final.err; echo "bench.py failed"; }
mkdir -p $out/profiles; cp profiles/${tag}* profiles/pmc_*_current.json $out/profiles/ 2>/dev/null || true
