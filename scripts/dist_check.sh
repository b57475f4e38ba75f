#!/bin/bash
# Functional check of the sharded (multi-rank) path on a ONE-GPU box: plain `python bench.py --gpus P` spawns P ranks
# that SHARE GPU 0; the native communicator then uses its p2p transport (HIP IPC staging buffers, reduction on the
# device) because RCCL refuses duplicate devices.  Timing is meaningless (P ranks time-slice one GPU).  What it proves:
# the launcher, the id exchange, the sample sharding, the all-reduce enqueued by the fused C solve and the k x k
# all-reduce of the Gram-form Rayleigh quotient give the eigenpairs of the FULL sample set (parity against the oracle
# over all samples).
#   bash scripts/dist_check.sh            (run from the repo root; the ranks are child processes of bench.py)
for W in as pod; do
for P in 2 4; do
  extra="--samples-total 64"; [ $W = pod ] && extra=""
  python bench.py --gpus $P --workload $W --steps 1 --warmup 1 $extra --no-cpu-baseline 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['config']['workload'][:8], 'ranks', d['n_gpus'], d['communicator'], 'units/rank', d['config'].get('samples_per_gpu', d['config'].get('snapshots_per_gpu')), 'eig rel-err vs oracle(all samples)', d['parity']['eig_rel_err_vs_oracle'],
      'angle', [v for k, v in d['parity'].items() if k.startswith('principal')][0])"
done
done
