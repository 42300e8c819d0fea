"""Config c5 at Oxford scale on one GPU (synthetic): 23 runs x 400 database clouds + 23 x 120 query clouds from HOST
memory -> descriptors -> all ordered run pairs ranked (evaluate.py:293-332 protocol).  Prints the wall times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
E = bench.pkg("engine"); R = bench.pkg("retrieval")
dev = torch.device("cuda:0")
arch = os.environ.get("ARCH", "epc-net")
store = bench.build_store(arch, dev, 0)
eng = E.InferenceEngine(arch, bench.PARAMS, store, outer=bench.OUTER, micro_batch=64 if arch == "epc-net" else 256)
runs, n_db, n_q = 23, 400, 120
rng = np.random.default_rng(0)
db = rng.uniform(-1, 1, (runs * n_db, 4096, 3)).astype(np.float32)
qs = rng.uniform(-1, 1, (runs * n_q, 4096, 3)).astype(np.float32)
R.get_latent_vectors(eng, db[:128], batch_size=64)            # warm-up (weights packed, workspaces allocated)
t0 = time.perf_counter()
dvec = R.get_latent_vectors(eng, db, batch_size=64 if arch == "epc-net" else 256)
qvec = R.get_latent_vectors(eng, qs, batch_size=64 if arch == "epc-net" else 256)
t1 = time.perf_counter()
truth_rng = np.random.default_rng(1)
truths = {(m, n): [sorted(truth_rng.choice(n_db, size=truth_rng.integers(0, 5), replace=False).tolist()) for _ in range(n_q)]
          for m in range(runs) for n in range(runs) if m != n}
t2 = time.perf_counter()
res = R.evaluate_runs([dvec[i * n_db:(i + 1) * n_db] for i in range(runs)], [qvec[i * n_q:(i + 1) * n_q] for i in range(runs)],
                      lambda m, n: truths[(m, n)])
t3 = time.perf_counter()
ncl = runs * (n_db + n_q)
print("c5 synthetic %s: %d clouds from host memory -> descriptors in %.2f s (%.0f clouds/s incl. H2D); %d run pairs ranked + "
      "recall bookkeeping in %.2f s; recall@1 %.2f %% (random truth)" % (arch, ncl, t1 - t0, ncl / (t1 - t0), runs * (runs - 1),
                                                                          t3 - t2, res["ave_recall"][0]))
