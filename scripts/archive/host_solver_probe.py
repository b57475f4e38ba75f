"""Where does the host sparse-LU callback spend its time?  (GPU box; prints timings)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hippyflow_amd as hf
from hippyflow_amd import workloads
nx, ny, k = 500, 400, 74
N = nx * ny
t = time.time(); prior = workloads.BiLaplacianPrior(nx, ny); print("factor", time.time() - t, "threads", prior.Rsolver.threads, flush=True)
X = np.asfortranarray(np.random.default_rng(0).standard_normal((N, k)))
S = prior.Rsolver
for thr in (1, 8, 32, 64):
    S.threads = thr; S._pool = None
    t = time.time(); S.solve_block(X); print("direct solve_block(74) threads=%d: %.2f s" % (thr, time.time() - t), flush=True)
    t = time.time(); S.solve_block(X[:, :32]); print("direct solve_block(32) threads=%d: %.2f s" % (thr, time.time() - t), flush=True)
t = time.time(); S.lu.solve(X[:, :8]); print("raw lu.solve 8 cols (threaded BLAS): %.2f s" % (time.time() - t))
from threadpoolctl import threadpool_limits
with threadpool_limits(limits=1):
    t = time.time(); S.lu.solve(X[:, :8]); print("raw lu.solve 8 cols (BLAS 1 thread): %.2f s" % (time.time() - t))
Xd = hf.MultiVector.from_dense(X)
Yd = hf.MultiVector(N, k)
for thr, chunk in ((32, 32), (32, 0), (1, 0), (8, 16)):
    S.threads = thr; S._pool = None
    op = hf.HostCallbackOperator(S, N, chunk_vectors=chunk)
    t = time.time(); op.matMvMult(Xd, Yd); hf.Context.default().synchronize()
    print("callback threads=%d chunk=%d: %.2f s" % (thr, chunk, time.time() - t), flush=True)
