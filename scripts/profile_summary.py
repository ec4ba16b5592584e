#!/usr/bin/env python3
"""profiles/<tag>_summary.md from a rocprofv3 --kernel-trace --stats run of bench.py and the
un-profiled bench line.  usage: profile_summary.py <tag> <kernel_stats.csv> <bench.json> <bench_under_rocprof.json>"""
import csv
import json
import re
import sys

tag, stats_csv, bench_json, prof_json = sys.argv[1:5]
bench = json.load(open(bench_json))
prof = json.load(open(prof_json))
rows = list(csv.DictReader(open(stats_csv)))
OURS = ("conv3d_mfma", "conv_split", "conv_once", "conv_s2", "basicblock2d", "deconv_split", "conv_zs", "deconv3d", "cout1", "soft_argmin", "volume_", "pack_weights", "absmax", "spp_", "corr1d", "box_filter", "warp_", "stage_pair", "decoder_cat", "bn_")


def short(n):
    n = re.sub(r"^void\s+", "", n).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", n)[:80]


def plan_name(n):
    """rocprof's demangled template name -> the name bench.py uses"""
    m = re.match(r"conv3d_mfma_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>", n)
    if m:
        S, NT, TM, CK, KZ, K, DIL = map(int, m.groups())
        return ("conv3d_mfma_kernel<S=%d,NT=%d,TM=%d,CK=%d>" % (S, NT, TM, CK) if KZ == 3 else
                "conv2d_mfma_kernel<S=%d,NT=%d,TM=%d,K=%d,DIL=%d>" % (S, NT, TM, K, DIL))
    PR = {"3": "bf16x3", "2": "f16x2", "1": "f16"}
    m = re.match(r"conv_zs_kernel<(\d+)>", n)
    if m:
        return "conv3d_zs_%s_mfma_kernel" % PR[m.group(1)]      # (the <vol> launch shares the symbol)
    m = re.match(r"conv_split_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>", n)
    if m:
        PM, NT, TM, KZ, DIL, S, nsplit = m.groups()
        pr, NT, TM, KZ, DIL, S, nsplit = PR[PM], int(NT), int(TM), int(KZ), int(DIL), int(S), int(nsplit)
        if nsplit > 1 and KZ == 3:
            return "conv3d_%s_mfma_kernel<NT=%d,TM=%d>x%d" % (pr, NT, TM, nsplit)
        if nsplit > 1:
            return "conv2d_%s_mfma_kernel<NT=%d,TM=%d,DIL=%d>x%d" % (pr, NT, TM, DIL, nsplit)
        if S == 2:
            return "conv3d_%s_mfma_kernel<S=2,NT=%d,TM=%d>" % (pr, NT, TM)
        return ("conv3d_%s_mfma_kernel<NT=%d,TM=%d>" % (pr, NT, TM) if KZ == 3 else
                "conv2d_%s_mfma_kernel<NT=%d,TM=%d,DIL=%d>" % (pr, NT, TM, DIL))
    m = re.match(r"basicblock2d_kernel<(\d+), (\d+)>", n)
    if m:
        return "basicblock2d_%s_mfma_kernel<C=%s>" % (PR[m.group(1)], m.group(2))
    m = re.match(r"conv_once_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)>", n)
    if m:
        return "conv2d_%s_mfma_kernel<NT=%s,TM=%s,DIL=1>x%s,once" % (PR[m.group(1)], m.group(2), m.group(3), m.group(5))
    m = re.match(r"conv_s2_kernel<(\d+)>", n)
    if m:
        return "conv3d_%s_mfma_kernel<S=2,NT=2,TM=1>" % PR[m.group(1)]
    m = re.match(r"deconv_split_kernel<(\d+), (\d+)>", n)
    if m:
        return "deconv3d_%s_mfma_kernel<NT=%s>" % (PR[m.group(1)], m.group(2))
    m = re.match(r"deconv3d_mfma_kernel<(\d+), (\d+)>", n)
    if m:
        return "deconv3d_mfma_kernel<NT=%s,CK=%s>" % m.groups()
    n = re.sub(r"<.*$", "", n)
    return n


total = sum(float(r["TotalDurationNs"]) for r in rows)
ours = [r for r in rows if any(k in r["Name"] for k in OURS)]
ours_total = sum(float(r["TotalDurationNs"]) for r in ours)
roofs = bench.get("rooflines", {})
print("# Run %s — `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline`\n" % tag.upper())
print("PSMNet D=192, 384x1280, 1 MI355X.  Un-profiled bench line (`%s_bench.json`): **%.1f pairs/s**, %.2f ms/step (%s);"
      % (tag, bench["value"], bench["ms_per_step"], bench["config"].get("launch", "eager launches")))
cb = bench.get("cpu_baseline")
if cb:
    print("CPU baseline %.3f pairs/s on %d host threads; parity vs CPU oracle %.1e px." % (cb["value"], cb["cores"], bench["parity_max_abs_px_vs_cpu"]))
print("Same command under the profiler: %.1f pairs/s.\n" % prof["value"])
print("Kernels of this repository are %.1f %% of all GPU time in the trace (the rest: MIOpen/torch kernels of the BN + head calibration passes, image staging, layout copies).\n" % (100 * ours_total / total))
print("| kernel (this repo) | calls | avg us (rocprof) | avg us (bench HIP events) |")
print("|---|---|---|---|")
for r in sorted(ours, key=lambda r: -float(r["TotalDurationNs"])):
    s = short(r["Name"])
    ev = roofs.get(plan_name(s), {}).get("avg_launch_us")
    print("| `%s` | %s | %.1f | %s |" % (s, r["Calls"], float(r["AverageNs"]) / 1e3, "%.1f" % ev if ev else "-"))
d = bench["roofline"]
print("\nDominant kernel `%s`: %.1f %s = %.1f %% of the %.1f peak (%d launches/step, %.2f ms/step)."
      % (d["kernel"], d["achieved"], d["unit"], 100 * d["frac"], d["peak"], d["launches_per_step"], d["ms_per_step"]))
if "mfma_tflops_executed" in d:
    print("That peak is the dense 16-bit MFMA peak (%.0f TF/s) / %d MFMAs per fp32 product (%s); executed MFMA rate %.0f TF/s; "
          "%.2fx the fp32-input MFMA peak (157.3 TF/s) in algorithmic fp32 FLOP/s." % (d["mfma_peak_16bit"], d["mfmas_per_product"], d["mfma_dtype"], d["mfma_tflops_executed"], d["x_fp32_mfma_peak"]))
v = bench.get("cost_volume_build")
if v:
    print("Cost-volume build (the default forward no longer launches it; measured in the same model's materialising path): %.1f us -> %.0f GB/s = %.1f %% of 8 TB/s (north-star target >= 60 %%), %.1f %% of the 6.29 TB/s copy ceiling; back to back %.1f us = %.1f %%; %.1f MB algorithmic."
          % (v["in_forward_us"], v["achieved"], 100 * v["frac"], 100 * v["frac_of_copy_ceiling"], v["back_to_back_us"], 100 * v["back_to_back_frac"], v["algorithmic_bytes"] / 1e6))
v = roofs.get("volume_ndhwc_fwd_kernel")
if v:
    print("Cost-volume build: %.1f us by in-bench events -> %.0f GB/s = %.1f %% of 8 TB/s (north-star target >= 60 %%), %.1f %% of the 6.29 TB/s copy ceiling; PMC traffic %.1f MB vs %.1f MB algorithmic (profiles/r01_d_pmc.md)."
          % (v["avg_launch_us"], v["achieved"], 100 * v["frac"], 100 * v.get("frac_of_copy_ceiling", 0), (v.get("traffic") or 0) / 1e6, v["work_per_launch"] / 1e6))
