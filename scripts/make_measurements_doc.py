"""Write docs/measurements.md from a round's files:  python scripts/make_measurements_doc.py r06 [dir with the files, default profiles]
Every number in the document is read from a file under profiles/ (named in the text); the prose between the tables is fixed text."""
import csv
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src = sys.argv[2] if len(sys.argv) > 2 else "profiles"
out = []
P = out.append


def path(name):
    return os.path.join(src, "%s_%s" % (tag, name))


def last_json(name):
    try:
        lines = [ln for ln in open(path(name)).read().strip().splitlines() if ln.startswith("{")]
        return json.loads(lines[-1])
    except (OSError, IndexError, ValueError):
        return None


def text(name):
    try:
        return open(path(name)).read()
    except OSError:
        return ""


def f(v, fmt="%.2f"):
    return "--" if v is None else fmt % v


pmc = {}
try:
    pmc = json.load(open(os.path.join("profiles", "pmc_traffic.json")))
except (OSError, ValueError):
    pass

P("# Measurements, round 6 (files: `profiles/%s_*`; one box, one build per job; regenerate with `scripts/make_measurements_doc.py`)\n" % tag)
P("Build tag of the PMC traffic record (`profiles/pmc_traffic.json`): `%s`.  Host of the GPU boxes: EPYC 9575F, 2 x 64 cores, 256 hardware threads, "
  "**cgroup CPU quota 16** (`cpu.max = 1600000 100000`).  Roofs as in `docs/kernels.md`.\n" % pmc.get("build_tag"))

# ------------------------------------------------------------------ headline + workloads
P("## 1. The three BASELINE workloads (un-profiled lines) and the driver's default run\n")
P("| workload | ms/step (median) | GDoF*rank/s | dominant kernel: avg launch, TF, frac of 78.6 | frac of the Gaussian-operand MFMA ceiling | PMC: HBM bytes / algorithmic | eig rel-err vs oracle | CPU BLAS-3 best (threads) | CPU reference-style: best single setting / composite |")
P("|---|---|---|---|---|---|---|---|---|")
for w, label in (("as", "config 4 AS, 512 x (100 x 2e5), r = 64, p = 10"), ("pod", "config 3 POD, 2048 x 5e5, r = 128, p = 10"), ("kle", "config 2 KLE, Matern N = 1e5, B = M")):
    d = last_json("bench_%s.json" % w)
    if not d:
        P("| %s | (no file) |" % label)
        continue
    r = d.get("roofline", {})
    cb = d.get("cpu_baseline") or {}
    rs = cb.get("reference_style") or {}
    comp = (rs.get("composite_of_per_component_best_settings") or {}).get("value")
    P("| %s | **%.2f** (%s) | **%.4g** | `%s`: %.2f ms, %.1f TF, **%.3f** | %s | %s | %.1e | %.2e (%s) | %s / %s |" % (
        label, d["ms_per_step"], f(d.get("median_ms_per_step")), d["value"], r.get("kernel"), r.get("avg_launch_ms", 0), r.get("achieved", 0), r.get("frac", 0),
        f(r.get("frac_of_in_job_random_operand_peak"), "%.3f"),
        ("%.2f / %.2f GB = %.3f" % (r["traffic"] / 1e9, r["algorithmic_bytes_per_launch"] / 1e9, r["traffic_over_algorithmic"])) if r.get("traffic") else "--",
        (d.get("parity") or {}).get("eig_rel_err_vs_oracle", float("nan")), cb.get("value", float("nan")), cb.get("cores"),
        f(rs.get("best_single_setting_value"), "%.2e"), f(comp, "%.2e")))
d = last_json("bench_default.json")
if d:
    pk = d.get("device_peaks_measured", {})
    P("\nThe driver's default command (`profiles/%s_bench_default.json`): headline **%.2f ms/step**, %.4f GDoF*rank/s, `%s` %.3f of 78.6 TF = %s of the "
      "Gaussian-operand ceiling (%s while streaming); in-job peaks: MFMA constant operands %.1f TF (%.1f beside the copy), **Gaussian operands %.1f TF** "
      "(%.1f beside the copy), copy %.0f GB/s, read-only stream %s GB/s; extras took %s s." % (
          tag, d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], f(d["roofline"].get("frac_of_in_job_random_operand_peak"), "%.3f"),
          f(d["roofline"].get("frac_of_in_job_random_operand_peak_while_streaming"), "%.3f"), pk.get("mfma_f64_tflops", 0), pk.get("mfma_f64_tflops_while_streaming", 0),
          pk.get("mfma_f64_tflops_random_operands", 0), pk.get("mfma_f64_tflops_random_operands_while_streaming", 0), pk.get("hbm_copy_gbs", 0),
          f(pk.get("hbm_read_gbs"), "%.0f"), {k: round(v) for k, v in (d.get("extras_wall_seconds") or {}).items()}))
    P("\n| extra key | ms/step (median) | dominant kernel frac | eig rel-err | notes |")
    P("|---|---|---|---|---|")
    for key in ("config3", "config2", "shard64", "shard64_rccl_1rank", "as_prior_shard64"):
        e = d.get(key) or {}
        if "error" in e:
            P("| `%s` | error: %s | | | |" % (key, e["error"]))
            continue
        note = ""
        ph = e.get("phases_ms_per_step") or {}
        if key.startswith("shard64_rccl"):
            note = "exposed all-reduce %.3f ms, overlapped %.3f ms" % (ph.get("allreduce", 0), ph.get("allreduce_overlapped", 0))
        if key == "as_prior_shard64":
            sf = e.get("sptrsv_feasibility") or {}
            note = "host_function %.0f ms, host_d2h_wait %.1f, host_h2d %.1f; %s; SpTRSV estimate: %s + %s levels, %.0f ms/step at %s us per level" % (
                ph.get("host_function", 0), ph.get("host_d2h_wait", 0), ph.get("host_h2d", 0), (e.get("host_solver") or "")[:120], sf.get("levels_L"), sf.get("levels_U"),
                sf.get("level_scheduled_ms_per_step_latency_part", 0), sf.get("assumed_us_per_level"))
        P("| `%s` | %s (%s) | %s | %s | %s |" % (key, f(e.get("ms_per_step")), f(e.get("median_ms_per_step")), f((e.get("roofline") or {}).get("frac"), "%.3f"),
                                             f((e.get("parity") or {}).get("eig_rel_err_vs_oracle"), "%.1e"), note))
for name, label in (("bench_as_shard64.json", "64-sample shard (one GPU's share of the 8-GPU run)"), ("bench_as_shard64_dist1.json", "the same with a one-rank RCCL communicator"),
                    ("bench_as_8ranks_one_gpu.json", "config 4 as 8 ranks SHARING the one GPU (p2p transport; rehearsal of --gpus 8)")):
    e = last_json(name)
    if e:
        P("\n%s (`profiles/%s_%s`): %.2f ms/step, %s; eig rel-err %.1e." % (label, tag, name, e["ms_per_step"],
          "tn frac %.3f" % e["roofline"]["frac"] if e.get("roofline") else "", (e.get("parity") or {}).get("eig_rel_err_vs_oracle", float("nan"))))
if pmc.get("kernels"):
    P("\nPMC passes of this build (`profiles/%s_pmc_{as,pod,kle}_summary.json` -> `profiles/pmc_traffic.json`):\n" % tag)
    P("| kernel, shape | HBM bytes per launch | MFMA pipe busy | effective clock |")
    P("|---|---|---|---|")
    for k, v in pmc["kernels"].items():
        P("| `%s` | %.2f GB | %s | %s GHz |" % (k, v["hbm_bytes_per_launch"] / 1e9, f(v.get("mfma_pipe_util"), "%.3f"), f(v.get("effective_clock_ghz"), "%.2f")))

# ------------------------------------------------------------------ kernel point
kp = (d or {}).get("kernel_point") if d else None
if not kp:
    kpj = last_json("kernel_point.json")
    kp = kpj.get("kernel_point") if isinstance(kpj, dict) and "kernel_point" in kpj else None
P("\n## 2. North-star kernel point `G = X^T Omega`, N = 1e6, k = 138 (key `kernel_point` of the driver's line)\n")
if kp:
    pm = kp.get("peaks_measured_in_job") or {}
    P("In-job: read-only stream %s GB/s, MFMA on Gaussian operands %s TF.  `frac of measured` = max(bytes / read rate, flops / MFMA rate) / time: at the ridge both are loaded at once.\n" % (
        f(pm.get("hbm_read_gbs"), "%.0f"), f(pm.get("mfma_f64_tflops_random_operands"), "%.1f")))
    P("| n | kernel | ms | frac of 8 TB/s | frac of 78.6 TF | binding roof (spec) | frac of measured (binding) |")
    P("|---|---|---|---|---|---|---|")
    for r in kp["rows"]:
        P("| %d | %s | %.3f | %.3f | %.3f | %s | %s (%s) |" % (r["n"], r["kernel"], r["ms"], r["hbm_frac"], r["mfma_frac"], r["bound"], f(r.get("frac_of_measured"), "%.3f"), r.get("bound_measured", "--")))
P("\nStall breakdown of `k_tsgemm_ssb` at n = 48 / 64 / 96 / 138 (seven counter groups, `profiles/%s_ssb_stall_breakdown.json`; raw means per launch, per wave quad-cycle):\n" % tag)
try:
    sb = json.load(open(path("ssb_stall_breakdown.json")))
    P("| instance | ms | clock GHz | MFMA busy | SQ_WAIT_ANY | SQ_WAIT_INST_ANY | SQ_ACTIVE_INST_VALU | SQ_ACTIVE_INST_LDS | INSTS_MFMA : VALU : LDS : VMEM_RD : SALU |")
    P("|---|---|---|---|---|---|---|---|---|")
    for k, v in sorted(sb.items(), key=lambda kv: kv[1]["avg_duration_ms"]):
        if "ssb" not in k:
            continue
        c = v["raw_mean_counters"]
        wc = c.get("SQ_WAVE_CYCLES", 1)
        P("| `%s` | %.3f | %.2f | %.3f | %.3f | %.3f | %.3f | %.3f | %.3g : %.3g : %.3g : %.3g : %.3g |" % (
            k.replace("void ", "")[:28], v["avg_duration_ms"], v.get("effective_clock_ghz", 0), v.get("mfma_pipe_util", 0), c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc,
            c.get("SQ_ACTIVE_INST_VALU", 0) / wc, c.get("SQ_ACTIVE_INST_LDS", 0) / wc, c.get("SQ_INSTS_MFMA", 0), c.get("SQ_INSTS_VALU", 0) - c.get("SQ_INSTS_MFMA", 0),
            c.get("SQ_INSTS_LDS", 0), c.get("SQ_INSTS_VMEM_RD", 0), c.get("SQ_INSTS_SALU", 0)))
except (OSError, ValueError):
    P("(file missing)")

# ------------------------------------------------------------------ eigensolver
P("\n## 3. The whole-GPU eigensolver, 256 < n <= 16384 (`hfmi_sym_eig_small`, all eigenvectors, host matrix in, host matrices out)\n")
el = text("eig_large.txt")
rows = {}
for m in re.finditer(r"\[hfmi eig n=(\d+)\] ms: workspace \+ fills ([\d.]+) \| upload ([\d.]+) \| load ([\d.]+) \| tridiagonalisation ([\d.]+) \| leaves ([\d.]+) \| merges ([\d.]+) \| back-transformation ([\d.]+) \| output ([\d.]+)", el):
    rows.setdefault(int(m.group(1)), []).append([float(x) for x in m.groups()[1:]])
summ = {}
for m in re.finditer(r"n=(\d+)\s+sym_eig_small ([\d.]+) ms \(min of (\d+); median ([\d.]+), max ([\d.]+), max/min ([\d.]+)\)(?:\s+numpy.linalg.eigh ([\d.]+) ms\s+\| eig err ([\d.e+-]+)\s+orth ([\d.e+-]+)\s+resid ([\d.e+-]+))?", el):
    summ[int(m.group(1))] = m.groups()[1:]
r5 = {300: 3.47, 512: 5.57, 1024: 11.48, 2048: 28.64, 4096: 87.87, 8192: 342.52}
bars = {2048: 20, 4096: 50, 8192: 170}
P("`profiles/%s_eig_large.txt` (`scripts/eig_large_time.py`, phase split from `HFMI_EIG_LARGE_TIMING=1`, which adds a stream synchronisation per phase: the bench's `eig_large` key below is the un-instrumented time).  Median phase times of the timed calls, ms:\n" % tag)
P("| n | this build (min) | round 5 | bar | host `eigh` | upload | load + fills | tridiagonalisation (us / column) | leaves | merges | back-transformation | read-back | max err: eigenvalues / `V^T V - I` / residual |")
P("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for n in sorted(rows):
    ph = sorted(rows[n][1:] or rows[n], key=lambda r: sum(r))
    med = ph[len(ph) // 2]
    s_ = summ.get(n)
    P("| %d | **%s ms** | %s | %s | %s | %.2f | %.2f | %.1f (%.1f) | %.2f | %.2f | %.2f | %.2f | %s |" % (
        n, s_[0] if s_ else "--", r5.get(n, "--"), bars.get(n, "--"), ("%s ms" % s_[5]) if s_ and s_[5] else "--", med[1], med[0] + med[2], med[3], 1e3 * med[3] / n, med[4], med[5], med[6], med[7],
        ("%s / %s / %s" % (s_[6], s_[7], s_[8])) if s_ and s_[6] else "--"))
t20 = text("eig_large_20calls.txt")
if t20:
    P("\n20 back-to-back calls per size, host BLAS inside the CPU quota (`profiles/%s_eig_large_20calls.txt`):\n" % tag)
    P("```")
    for ln in t20.splitlines():
        if ln.startswith("n=") or ln.startswith("   ") or ln.startswith("cpu budget"):
            P(ln[:230])
    P("```")
eb = ((d or {}).get("eig_large") or {}).get("rows") if d else None
if eb:
    P("\nKey `eig_large` of the driver's line (un-instrumented, min of three calls; `hfmi_block_gram_eig` = the POD form: Gram matrix formed on the device, 128 eigenvectors returned):\n")
    P("| n | all eigenvectors, ms | POD form (128 vectors), ms | host `eigh`, ms | eigenvalue err vs host | orthonormality | residual |")
    P("|---|---|---|---|---|---|---|")
    for r in eb:
        P("| %d | %.2f | %.2f | %s | %s | %.1e | %.1e |" % (r["n"], r["ms"], r["gram_eig_128_ms"], f(r.get("host_eigh_ms"), "%.0f"), f(r.get("eig_err_rel_vs_host"), "%.1e"), r["orthonormality"], r["residual_rel"]))
for n in (4096, 8192):
    t = text("pmc_eig_n%d.txt" % n)
    if t:
        P("\nCounters of one solve at n = %d (`profiles/%s_pmc_eig_n%d_summary.json`; PMC passes serialise the launches, durations are the isolated ones; the clock column = GRBM_GUI_ACTIVE / duration is meaningful for launches of 0.1 ms and more only):\n" % (n, tag, n))
        P("```")
        for ln in t.splitlines():
            if ln.strip() and not ln.startswith("/opt") and "amdgpu.ids" not in ln:
                P(ln[:200])
        P("```")
    st = path("eig_large_n%d_kernel_stats.csv" % n)
    if os.path.exists(st):
        P("\nKernel trace at n = %d (`profiles/%s_eig_large_n%d_kernel_stats.csv`, warm-up + three timed solves):\n" % (n, tag, n))
        P("| kernel | calls | total ms | avg us |")
        P("|---|---|---|---|")
        for i, r in enumerate(csv.DictReader(open(st))):
            if i >= 12:
                break
            P("| `%s` | %s | %.1f | %.1f |" % (re.sub(r"^_ZN12_GLOBAL__N_1\d+|^_Z\d+", "", r["Name"].replace(".kd", ""))[:48], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
for name, title in (("dgemm_rate.txt", "General fp64 MFMA product of the eigensolver (`scripts/dgemm_rate.py`)"), ("tile_stride_probe.txt", "Tile-stride probe (`scripts/tile_stride_probe.hip`): the access patterns of `k_tri_bs` / `k_tri_b` alone, by leading dimension"),
                    ("eig_stall_diagnosis.txt", "Stall diagnosis (`scripts/eig_stall_diagnosis.py`): 30 calls at n = 2048 with a host matmul between calls"),
                    ("tn_tile_ab.txt", "The taller-tile experiment on `k_tsgemm_tn` (`scripts/tn_tile_ab.sh`; config 4, same box, interleaved)"),
                    ("pod_routes_time.txt", "The deterministic POD in its two exact forms (`scripts/pod_routes_time.py`): n x n Gram problem vs N x N state-dimension form"),
                    ("sync_price_probe.txt", "What one device-wide dependency costs (`scripts/sync_price_probe.hip`)"),
                    ("eig_tuning_ab.txt", "Threshold A/B of the eigensolver (`HFMI_EIG_SYM_MIN`, `HFMI_EIG_UNB_MAX`; defaults 3072 / 2048)")):
    t = text(name)
    if t:
        P("\n### %s -- `profiles/%s_%s`\n" % (title, tag, name))
        P("```")
        for ln in t.splitlines():
            if ln.strip() and "amdgpu.ids" not in ln:
                P(ln[:250])
        P("```")
open(os.path.join("docs", "measurements.md"), "w").write("\n".join(out) + "\n")
print("wrote docs/measurements.md: %d lines" % len(out))
