#!/bin/bash
# Same-box A/B of two library builds on the un-profiled bench lines (box-to-box variation is 2-3 %, so builds are only comparable
# inside ONE job).  Build the other library first, e.g. from an older hfmi_gemm.hip:
#   git show <commit>:hippyflow_amd/csrc/hfmi_gemm.hip > /tmp/old.hip; hipcc ... -c /tmp/old.hip -o hippyflow_amd/build/hfmi_gemm_exp_old.o; link as
#   scripts/build_variant.sh does -> hippyflow_amd/build/libhfmi_old.so
for rep in 1 2; do for lib in hippyflow_amd/build/libhfmi_old.so hippyflow_amd/libhfmi.so; do for w in as pod; do HFMI_LIB=$lib python bench.py --workload $w --no-cpu-baseline --no-check --no-literal 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib'.split('/')[-1], '$w', round(d['ms_per_step'],2), round(d['roofline']['frac'],3), [ (k['kernel'],round(k['ms_per_step'],2)) for k in d['kernels'][:3]])"; done; done; done
