#!/bin/bash
# GPU box: A/B of a variant library's inference step (bench.py's headline leg only): value, ms per step, the four block stages
for i in 1 2; do for v in "" "$1"; do
  echo "lib ${v:-default}"
  EPCNET_LIB=${v:+$PWD/build_variants/lib_$v.so} python bench.py --no-configs --no-cpu-baseline --precision f32 --regions 3 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], {k:v for k,v in d["stage_ms"].items() if k.startswith("block") or k=="knn"})'
done; done
