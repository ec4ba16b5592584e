import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], "pairs/s", d["ms_per_step"], "ms | hip", d.get("hip_path_ms_per_step"), "| other", d.get("other_ms_per_step", d.get("stock_torch_ms_per_step")), "| instrumented", (d.get("kernel_timing") or {}).get("instrumented_ms_per_step"))
for k, v in sorted(d.get("rooflines", {}).items(), key=lambda kv: -kv[1]["ms_per_step"]):
    print("%-46s %8.2f %-8s frac %.3f x%-3.0f avg %8.1f us %7.3f ms" % (k, v["achieved"], v["unit"], v["frac"], v["launches_per_step"], v["avg_launch_us"], v["ms_per_step"]))
if "cpu_baseline" in d: print(d["cpu_baseline"], d.get("parity_max_abs_px_vs_cpu"))
