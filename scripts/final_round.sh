#!/bin/bash
# Everything a round's evidence consists of, on ONE box and ONE build:  bash scripts/final_round.sh r04j
#   1. the -m gpu test suite                      -> <tag>_gputests.log
#   2. kernel traces + PMC passes                 -> scripts/profile_round.sh, scripts/pmc_workload.sh; profiles/pmc_traffic.json of THIS
#      build is made from them (gpurun_out/<tag>_pmc_traffic.json: copy it over profiles/pmc_traffic.json when the job is back)
#   3. un-profiled bench lines (configs 4 / 3 / 2 with roofline.traffic from step 2, the 64-sample shard with and without a
#      one-rank RCCL communicator, 2 ranks sharing the GPU)                   -> <tag>_bench_*.json
#   4. the un-profiled kernel point               -> <tag>_kernel_point.json
#   5. kernel timeline of the shard step          -> scripts/shard_profile.sh
#   6. the eigensolve beyond the BASELINE sizes   -> scripts/eig_corner_trace.sh; beyond 256: scripts/eig_large_time.py, eig_large_trace.sh
#   7. the 8-rank rehearsal on one GPU and the driver's default command (headline + extras)
# Summaries land in gpurun_out/; copy what is to be judged into profiles/.
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out; mkdir -p $out
cd $R
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $out/${tag}_gputests.log
bash scripts/profile_round.sh $tag > $out/${tag}_profile_round.log 2>&1
for w in as pod kle; do MIN_MS=0.05 bash scripts/pmc_workload.sh $w $tag > $out/${tag}_pmc_$w.log 2>&1; done
cd $R
python profiles/make_pmc_traffic.py $tag $out > $out/${tag}_make_pmc_traffic.log 2>&1 && cp profiles/pmc_traffic.json $out/${tag}_pmc_traffic.json
for w in as pod kle; do
  timeout 900 python bench.py --workload $w --headline-only > $out/${tag}_bench_$w.json 2> $out/${tag}_bench_$w.err
done
timeout 600 python bench.py --samples-total 64 --no-cpu-baseline > $out/${tag}_bench_as_shard64.json 2>/dev/null
timeout 600 python bench.py --samples-total 64 --no-cpu-baseline --dist-single > $out/${tag}_bench_as_shard64_dist1.json 2>/dev/null
timeout 600 python bench.py --gpus 2 --samples-total 128 --steps 2 --warmup 1 --no-cpu-baseline > $out/${tag}_bench_as_2ranks_one_gpu.json 2>/dev/null
# the 8-rank rehearsal of the driver's --gpus 8 run: all 512 samples, 64 per rank, the eight ranks sharing this box's one GPU (p2p transport)
timeout 900 python bench.py --gpus 8 --steps 2 --warmup 1 --no-cpu-baseline > $out/${tag}_bench_as_8ranks_one_gpu.json 2> $out/${tag}_bench_as_8ranks_one_gpu.err
# the driver's own command: headline + the extra evidence keys (kernel point, configs 3 / 2, shard step) in ONE line
timeout 900 python bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.err
timeout 600 python scripts/kernel_point.py > $out/${tag}_kernel_point.log 2>&1 && cp $out/kernel_point.json $out/${tag}_kernel_point.json
bash scripts/shard_profile.sh ${tag}_shard64 > $out/${tag}_shard64_timeline.txt 2>&1
bash scripts/shard_profile.sh ${tag}_shard64_dist1 --samples-total 64 --dist-single > $out/${tag}_shard64_dist1_timeline.txt 2>&1
bash scripts/eig_corner_trace.sh $tag 138 160 192 224 256 > $out/${tag}_eig_corner.txt 2>&1
# the whole-GPU eigensolver beyond 256 (hfmi_eig_blocked.hip): wall times next to numpy.linalg.eigh with the phase split (host BLAS kept
# inside the container's CPU quota: scripts/eig_stall_diagnosis.py says why), 20 back-to-back calls per size, traces, counters
HFMI_EIG_LARGE_TIMING=1 timeout 900 python scripts/eig_large_time.py 300 512 1024 2048 4096 8192 --blas-threads=8 > $out/${tag}_eig_large.txt 2>&1
timeout 600 python scripts/eig_large_time.py 1024 2048 4096 8192 --no-host --reps=20 --diag --blas-threads=8 > $out/${tag}_eig_large_20calls.txt 2>&1
timeout 300 python scripts/eig_large_time.py 16384 --no-host --reps=2 --blas-threads=8 >> $out/${tag}_eig_large_20calls.txt 2>&1
timeout 300 python scripts/eig_stall_diagnosis.py 2048 > $out/${tag}_eig_stall_diagnosis.txt 2>&1
timeout 300 python scripts/dgemm_rate.py > $out/${tag}_dgemm_rate.txt 2>&1
./hippyflow_amd/build/tile_stride_probe > $out/${tag}_tile_stride_probe.txt 2>&1
bash scripts/eig_large_trace.sh $tag 512 1024 2048 4096 8192 > $out/${tag}_eig_large_trace.log 2>&1
for n in 4096 8192; do bash scripts/pmc_eig.sh $tag $n > $out/${tag}_pmc_eig_n$n.txt 2>&1; done
cd $R
cat $out/${tag}_gputests.log
ls $out | grep "^${tag}_" | wc -l
