#!/bin/bash
# GPU box: collect the artefacts bench.py's roofline object cites, into gpurun_out/prof_<tag>/.
#   scripts/collect_profiles.sh <tag>        (then: python scripts/summarise_profiles.py <tag> on either side)
# Three SEPARATE rocprofv3 runs of the same bench command (kernel stats; FETCH_SIZE; WRITE_SIZE): the two TCC counters do
# not fit one pass, and counters are never combined with other trace domains (MI355X_MICROARCH.md, HBM section).
set -u
tag=${1:-cur}
cd "$(dirname "$0")/.."
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
args="bench.py --steps 20 --warmup 5 --no-cpu-baseline --in-flight 1"
python3 bench.py > "$out/bench_line.json" 2> "$out/bench.err"        # the un-profiled line: bench.py's own defaults
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 $root/$args > "$out/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --in-flight 1 > "$out/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --in-flight 1 > "$out/pmc_write.log" 2>&1
cd "$root"
python3 scripts/summarise_profiles.py "$tag"
