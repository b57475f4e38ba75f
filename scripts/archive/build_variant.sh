#!/bin/bash
# Experimental build of libhfmi with extra defines for ONE translation unit (timing experiments; results may be garbage):
#   bash scripts/build_variant.sh tn1 hfmi_gemm.hip -DTN_EXP=1     ->  hippyflow_amd/build/libhfmi_tn1.so
# then   HFMI_LIB=hippyflow_amd/build/libhfmi_tn1.so python scripts/tn_probe.py
name=$1; tu=$2; shift; shift
R=$(cd "$(dirname "$0")/.." && pwd)
B=$R/hippyflow_amd/build
python3 -c "import sys; sys.path.insert(0, '$R'); from hippyflow_amd import _build; _build.build(verbose=False)"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$R/hippyflow_amd/csrc "$@" -c $R/hippyflow_amd/csrc/$tu -o $B/${tu%.hip}_$name.o || exit 1
objs=""
for o in $B/hfmi_*.o; do
  case $o in *_tn[0-9]*.o|*_exp*.o|*_$name.o) continue;; esac
  [ "$o" = "$B/${tu%.hip}.o" ] && o=$B/${tu%.hip}_$name.o
  objs="$objs $o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libhfmi_$name.so $objs -ldl -lrt -lpthread && echo built $B/libhfmi_$name.so
