#!/bin/bash
# GPU box: collect the artefacts bench.py's roofline object cites, into gpurun_out/prof_<tag>/.
#   scripts/collect_profiles.sh <tag> [side <extra bench.py arguments>]
#   e.g.  scripts/collect_profiles.sh r05_l side --arch epc-net-l --batch 256      (a SIDE collection: per-tag summaries only, the
#   *_current.json files bench.py reads stay those of the headline configuration)
# Three SEPARATE rocprofv3 runs of the same bench command (kernel stats; FETCH_SIZE; WRITE_SIZE): the two TCC counters do
# not fit one pass, and counters are never combined with other trace domains (MI355X_MICROARCH.md, HBM section).
set -u
tag=${1:-cur}
side=""; extra=""
if [ "${2:-}" == "side" ]; then side=side; shift 2; extra="$*"; fi
cd "$(dirname "$0")/.."
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
args="bench.py --steps 20 --warmup 5 --no-cpu-baseline --in-flight 1 --no-configs --no-rccl --regions 1 $extra"
if [ -z "$side" ]; then python3 bench.py > "$out/bench_line.json" 2> "$out/bench.err"; fi       # the un-profiled line: bench.py's own defaults
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 $root/$args > "$out/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --in-flight 1 --no-configs --no-rccl --regions 1 $extra > "$out/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --in-flight 1 --no-configs --no-rccl --regions 1 $extra > "$out/pmc_write.log" 2>&1
# Compute-side counters, two more passes of the same command: the matrix pipe's busy cycles and the vector
# ALU's issue / active-lane counters (pass A), LDS and wait-state counters (pass B).  8 SQ slots per pass; GRBM_GUI_ACTIVE
# rides in the independent GRBM block.  Counter names are checked against `rocprofv3 -L` first: a name this ROCm does not
# know is dropped from the pass instead of failing it.
rocprofv3 -L > "$out/counters_available.txt" 2>&1
pick() { for c in "$@"; do grep -qw "$c" "$out/counters_available.txt" && printf "%s " "$c"; done; }
setA=$(pick SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE)
setB=$(pick SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_SALU GRBM_GUI_ACTIVE)
echo "pass A: $setA" > "$out/pmc_compute_sets.txt"; echo "pass B: $setB" >> "$out/pmc_compute_sets.txt"
cd /tmp
[ -n "$setA" ] && rocprofv3 --kernel-trace --pmc $setA --output-format csv -d "$out/pmc_compA" -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --in-flight 1 --no-configs --no-rccl --regions 1 $extra > "$out/pmc_compA.log" 2>&1
[ -n "$setB" ] && rocprofv3 --kernel-trace --pmc $setB --output-format csv -d "$out/pmc_compB" -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --in-flight 1 --no-configs --no-rccl --regions 1 $extra > "$out/pmc_compB.log" 2>&1
cd "$root"
python3 scripts/summarise_profiles.py "$tag" $side
