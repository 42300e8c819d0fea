mkdir -p gpurun_out/r4p
# NOTE: collect on a CLEAN build of the library (make -B -C epc-net_amd/csrc): bench.py reads the counters only while the loaded
# library has the hash stamped here, and the driver builds from scratch (objects left by ad-hoc hipcc runs gave another hash).
python -m pytest tests -m gpu -x -q > gpurun_out/r4p/gpu_tests.log 2>&1; grep -a "passed\|failed" gpurun_out/r4p/gpu_tests.log | tail -2
bash scripts/collect_profiles.sh r04_b > gpurun_out/r4p/collect.log 2>&1
bash scripts/prof_train.sh r4p
python scripts/stress_parity.py 90 > gpurun_out/r4p/stress_fast.txt 2>&1
PRECISION=f32 python scripts/stress_parity.py 90 > gpurun_out/r4p/stress_f32.txt 2>&1
tail -3 gpurun_out/r4p/stress_f32.txt
mkdir -p gpurun_out/r4p/profiles; cp profiles/r04_b_* profiles/pmc_*_current.json gpurun_out/r4p/profiles/ 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/prof_r04_b/bench_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['stage_ms']); c=d['configs']; print(c['train_step']['ms_per_step'], c['train_step_bf16']['ms_per_step'], c['epc_net_l_b256']['ms_per_step'], c['retrieval']['value'])"
