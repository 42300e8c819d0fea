"""bench.py's N > 1 code path, run once before the driver's 8-GPU run does: `bench.py --gpus 2 --backend gloo
--same-device` -- the script launches itself under torch.distributed.run, both ranks on cuda:0 (RCCL refuses two ranks on one GPU, so
the dry run's backend is gloo) -- must print ONE JSON line from rank 0, report the backend truthfully, run every `configs` leg on both
ranks without an error, and its sharded retrieval must book exactly what the one-rank run books (integer counters: SURVEY.md 8e).
bench.py starts as a CHILD of the test process.  Numbers from this run are not credit."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--regions", "1", "--settle-ms", "0",
           "--no-cpu-baseline", "--precision", "f32", "--in-flight", "1"] + list(flags)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert p.returncode == 0, "bench.py %s exited %d:\n%s" % (" ".join(flags), p.returncode, p.stderr.decode()[-3000:])
    assert len(lines) == 1, "expected ONE line on stdout, got %d:\n%s" % (len(lines), "\n".join(ln[:200] for ln in lines))
    return json.loads(lines[0])


def test_bench_runs_its_two_rank_code_on_one_device():
    two = _bench("--gpus", "2", "--backend", "gloo", "--same-device")
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["steps"] == 3
    assert two["backend"] == {"name": "gloo", "rccl": False, "ranks": 2, "same_device": True}
    legs = two["configs"]
    assert set(legs) >= {"train_step", "train_step_bf16", "train_step_22", "train_step_bf16_22", "epc_net_l_b256", "retrieval"}
    for name, leg in legs.items():
        assert "error" not in leg, (name, leg)
    r2 = legs["retrieval"]
    assert r2["rccl_exercised"] is False and r2["rccl_ranks"] == 0 and r2["ranks"] == 2 and r2["backend"] == "gloo"
    assert "three graphs" in legs["train_step"]["workload"] and legs["train_step_22"]["clouds_per_tuple"] == 22
    # the one-rank run of the same script: the sharded evaluation's results do not depend on the sharding
    one = _bench("--gpus", "1", "--backend", "gloo")
    r1 = one["configs"]["retrieval"]
    assert one["backend"]["ranks"] == 1 and r1["ranks"] == 1
    for key in ("ave_recall_at_1", "ave_one_percent_recall", "average_similarity", "queries_ranked", "ordered_pairs"):
        assert r1[key] == r2[key], (key, r1[key], r2[key])
    # (no assertion on the rates: three-step regions on a box that has just run another GPU process are dominated by one dispatch
    # stall -- 2 k clouds/s was seen for a run that does 58 k -- and numbers from a dry run are not credit anyway)
    assert two["value"] > 0 and one["value"] > 0
