#!/bin/bash
# A/B of the half-height last round of the overlapped rank reduction (HFMI_NN_HALVE_LAST=0 switches it off): the 64-sample shard
# of config 4 with a one-rank RCCL communicator, three runs each, interleaved.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  for h in 1 0; do
    HFMI_NN_HALVE_LAST=$h timeout 600 python bench.py --samples-total 64 --no-cpu-baseline --no-check --dist-single 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('halve_last $h: step', round(d['ms_per_step'],3), 'ms', {k:round(v,3) for k,v in d['phases_ms_per_step'].items() if 'allreduce' in k or k=='apply_A'})"
  done
done
