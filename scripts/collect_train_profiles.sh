#!/bin/bash
# GPU box: counters of the quadruplet TRAINING step (BASELINE.json configs[2]) -> gpurun_out/prof_train_<tag>/, then
# scripts/summarise_train_profiles.py <tag> -> profiles/<tag>_train_pmc.json + profiles/pmc_train_current.json (what bench.py's train
# legs read for roofline.traffic / largest_kernels).
#   scripts/collect_train_profiles.sh <tag>
# Per (arithmetic, tuple size): one kernel-trace pass and three counter passes of the EAGER step (scripts/time_train_step.py with
# GRAPH=0: the same kernels on the same tensors as the replayed graph, one dispatch record each) -- FETCH_SIZE, WRITE_SIZE, and the
# matrix pipe's busy cycles beside GRBM_GUI_ACTIVE.  Counters never share a run with other trace domains, the program stands
# directly behind `--` (no env / shell hop), the settings travel in the environment of rocprofv3 itself.
set -u
tag=${1:-cur}
cd "$(dirname "$0")/.."
root=$PWD
out=$root/gpurun_out/prof_train_$tag
mkdir -p "$out"
export TMPDIR=/tmp GRAPH=0 WARM=3 STEPS=5
cd /tmp
for prec in bf16x6 bf16; do
  for neg in 14 18; do
    leg=${prec}_$((neg + 4))
    export PRECISION=$prec NEG=$neg
    rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$leg/stats" -- python3 $root/scripts/time_train_step.py > "$out/$leg.stats.log" 2>&1
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/$leg/fetch" -- python3 $root/scripts/time_train_step.py > "$out/$leg.fetch.log" 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/$leg/write" -- python3 $root/scripts/time_train_step.py > "$out/$leg.write.log" 2>&1
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/$leg/mfma" -- python3 $root/scripts/time_train_step.py > "$out/$leg.mfma.log" 2>&1
  done
done
cd "$root"
python3 scripts/summarise_train_profiles.py "$tag"
