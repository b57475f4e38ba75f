"""Where does a 20-80 ms stall of an eigensolve come from?  Reads the csv output of
  rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python3 scripts/eig_large_time.py N --no-host --reps=30
and prints (1) every HIP API call longer than THRESH ms with its neighbours, (2) every gap longer than THRESH ms between consecutive
GPU activities (kernels + copies) with the API calls that were in flight during the gap.  Usage: python scripts/find_stall.py DIR [THRESH_MS]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
thr = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 5e6


def rows(pattern):
    out = []
    for f in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        with open(f) as fh:
            out += list(csv.DictReader(fh))
    return out


api = rows("*hip_api_trace.csv")
ker = rows("*kernel_trace.csv")
cpy = rows("*memory_copy_trace.csv")
print("records: %d api, %d kernels, %d copies" % (len(api), len(ker), len(cpy)))
A = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in api))
G = sorted([(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in ker] +
           [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")) for r in cpy])
t0 = A[0][0] if A else 0
print("---- HIP API calls longer than %.1f ms" % (thr / 1e6))
for i, (s, e, f) in enumerate(A):
    if e - s > thr and i > 50:            # (skip start-up: module load, first allocations)
        print("  t=%.3f s  %-32s %.2f ms   before: %s | after: %s" % ((s - t0) / 1e9, f, (e - s) / 1e6, A[i - 1][2], A[i + 1][2] if i + 1 < len(A) else "-"))
print("---- GPU idle gaps longer than %.1f ms (after the first 100 activities)" % (thr / 1e6))
last_end = G[0][1] if G else 0
for i, (s, e, f) in enumerate(G):
    if i > 100 and s - last_end > thr:
        inflight = [(a[2], (a[1] - a[0]) / 1e6) for a in A if a[0] < s and a[1] > last_end and a[1] - a[0] > 1e6]
        print("  t=%.3f s  gap %.2f ms before %s (after %s); long API calls overlapping: %s" % ((s - t0) / 1e9, (s - last_end) / 1e6, f, G[i - 1][2], inflight[:6]))
    last_end = max(last_end, e)
