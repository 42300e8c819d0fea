#!/bin/bash
# GPU box: the inference step with the one-lane and the four-lane kNN form, alternating on the SAME box (boxes differ by 1-2 %).
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for q in 0 1; do
    for arch in epc-net epc-net-l; do
      b=64; [ $arch == epc-net-l ] && b=256
      EPC_KNN_QUAD=$q python bench.py --arch $arch --batch $b --precision f32 --no-cpu-baseline --no-configs --no-rccl --in-flight 1 --regions 3 --steps 100 2>/dev/null \
        | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('rep $rep quad $q $arch', d['ms_per_step'], {k: round(v, 4) for k, v in d['stage_ms'].items() if v})"
    done
  done
done
