#!/bin/bash
# Functional check of the sharded (multi-rank) path on a ONE-GPU box: P ranks share GPU 0 and all-reduce their device blocks
# through gloo (--backend gloo; timing is meaningless, the blocks go through the host).  What it proves: the sample
# sharding, the post-apply all-reduce hook inside the fused solve and the k x k all-reduce of the Gram-form Rayleigh
# quotient give the eigenpairs of the FULL sample set (parity against the oracle over all samples).
#   bash scripts/dist_check.sh            (run from the repo root; launches python as child processes)
for W in as pod; do
for P in 2 4; do
  extra="--samples-total 64"; [ $W = pod ] && extra=""
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $P --master-addr 127.0.0.1 --master-port $((29520 + P)) \
    bench.py --gpus $P --backend gloo --workload $W --steps 1 --warmup 1 $extra --no-cpu-baseline 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['config']['workload'][:8], 'ranks', d['n_gpus'], 'units/rank', d['config'].get('samples_per_gpu', d['config'].get('snapshots_per_gpu')), 'eig rel-err vs oracle(all samples)', d['parity']['eig_rel_err_vs_oracle'],
      'angle', [v for k, v in d['parity'].items() if k.startswith('principal')][0])"
done
done
