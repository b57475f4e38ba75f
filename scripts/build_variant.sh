#!/bin/bash
# Experimental build of libhfmi with extra defines for ONE translation unit (timing experiments; results may be garbage):
#   bash scripts/build_variant.sh p0 hfmi_gemm_nn.hip -DNN_PIPE3=0     ->  hippyflow_amd/build/libhfmi_p0.so
# then   HFMI_LIB=hippyflow_amd/build/libhfmi_p0.so python scripts/nn_pod_ab.py
name=$1; tu=$2; shift; shift
R=$(cd "$(dirname "$0")/.." && pwd)
B=$R/hippyflow_amd/build
python3 -c "import sys; sys.path.insert(0, '$R'); from hippyflow_amd import _build; _build.build(verbose=False)"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$R/hippyflow_amd/csrc "$@" -c $R/hippyflow_amd/csrc/$tu -o $B/${tu%.hip}_$name.o || exit 1
objs=""
for src in $(python3 -c "import sys; sys.path.insert(0, '$R'); from hippyflow_amd import _build; print(' '.join(_build.SOURCES))"); do
  o=$B/$(basename ${src%.hip}).o
  [ "$(basename $src)" = "$tu" ] && o=$B/${tu%.hip}_$name.o
  objs="$objs $o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libhfmi_$name.so $objs -ldl -lrt -lpthread && echo built $B/libhfmi_$name.so
