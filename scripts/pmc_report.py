"""profiles/<tag>_pmc.md and profiles/traffic.json from the per-kernel counter table of scripts/pmc_passes.sh
over bench.py: `python scripts/pmc_report.py <tag> <table.md>` (run in the repository, on the commit measured)."""
import json
import os
import subprocess
import sys

tag, table = sys.argv[1:3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = [l for l in open(table).read().splitlines() if l.startswith("|")]
hdr = [c.strip() for c in rows[0].strip("|").split("|")]
data = {}
for l in rows[2:]:
    c = [x.strip() for x in l.strip("|").split("|")]
    data[c[0].strip("`")] = dict(zip(hdr[1:], map(float, c[1:])))
# rocprof's kernel name -> the plan name bench.py reports
NAMES = {
    "conv_zs_kernel<2>": "conv3d_zs_f16x2_mfma_kernel",
    "basicblock2d_kernel<2, 64>": "basicblock2d_f16x2_mfma_kernel<C=64>",
    "basicblock2d_kernel<2, 32>": "basicblock2d_f16x2_mfma_kernel<C=32>",
    "conv_once_kernel<2, 1, 2, 4, 2>": "conv2d_f16x2_mfma_kernel<NT=1,TM=2,DIL=1>x2,once",
    "conv_split_kernel<2, 1, 2, 1, 1, 1, 2>": "conv2d_f16x2_mfma_kernel<NT=1,TM=2,DIL=1>x2",
    "conv_split_kernel<2, 2, 2, 1, 1, 1, 2>": "conv2d_f16x2_mfma_kernel<NT=2,TM=2,DIL=1>x2",
    "conv_split_kernel<2, 4, 2, 1, 2, 1, 1>": "conv2d_f16x2_mfma_kernel<NT=4,TM=2,DIL=2>",
    "conv_split_kernel<2, 1, 4, 1, 1, 1, 1>": "conv2d_f16x2_mfma_kernel<NT=1,TM=4,DIL=1>",
    "conv_s2_kernel<2>": "conv3d_f16x2_mfma_kernel<S=2,NT=2,TM=1>",
    "conv_split_kernel<2, 2, 1, 3, 1, 2, 1>": "conv3d_f16x2_mfma_kernel<S=2,NT=2,TM=1>",
    "conv_split_kernel<2, 2, 2, 3, 1, 1, 1>": "conv3d_f16x2_mfma_kernel<NT=2,TM=2>",
    "conv_split_kernel<2, 1, 1, 3, 1, 1, 2>": "conv3d_f16x2_mfma_kernel<NT=1,TM=1>x2",
    "deconv_split_kernel<2, 1>": "deconv3d_f16x2_mfma_kernel<NT=1>",
    "deconv_split_kernel<2, 2>": "deconv3d_f16x2_mfma_kernel<NT=2>",
    "conv3d_cout1_zslide_kernel": "conv3d_cout1_zslide_kernel",
    "soft_argmin_up4_kernel<4>": "soft_argmin_up4_kernel",
    "spp_concat_kernel": "spp_concat_kernel", "spp_pool8_kernel": "spp_pool8_kernel",
    "spp_branches_kernel": "spp_branches_kernel", "absmax_kernel": "absmax_kernel",
    "volume_ndhwc_fwd_kernel<32>": "volume_ndhwc_fwd_kernel",
}
commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"]).decode().strip()
tpath = os.path.join(root, "profiles", "traffic.json")
tj = json.load(open(tpath))
lines = []
for k, n in NAMES.items():
    if k not in data:
        continue
    d = data[k]
    corr = 1 if k.startswith("spp_branches") else 2          # 4-B-per-lane readers: no doubling
    b = int((d["FETCH_SIZE"] * corr + d["WRITE_SIZE"]) * 1024)
    tj["kernels"][n] = {"bytes_per_launch": b,
                        "source": "profiles/%s_pmc.md (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over bench.py; commit %s)" % (tag, commit)}
    cyc = d["GRBM_GUI_ACTIVE"] / 8
    lines.append("| `%s` | `%s` | %.0f | %.1f %% | %.0f %% | %.3g | %.1f |" % (
        k, n, cyc, 100 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 100 * d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"],
        d["SQ_LDS_BANK_CONFLICT"], b / 1e6))
json.dump(tj, open(tpath, "w"), indent=1)
print("""# %s — PMC passes over the benchmark itself (commit %s)

`scripts/pmc_passes.sh OUT bench.py --no-cpu-baseline --steps 5 --warmup 2` on one MI355X: four separate
`rocprofv3 --kernel-trace --pmc <set> -- python3 bench.py ...` passes (SQ/GRBM set; LDS set; FETCH_SIZE; WRITE_SIZE),
reduced with `scripts/pmc_table.py` to the mean per dispatch of every kernel in the run (PSMNet D=192, 384x1280,
default precision f16x2), then `scripts/pmc_report.py`.  FETCH/WRITE in KiB.

%s

Readings (cycles = `GRBM_GUI_ACTIVE` / 8 XCDs; MFMA busy = `SQ_VALU_MFMA_BUSY_CYCLES` / (1024 SIMDs x cycles); waiting =
`SQ_WAIT_INST_ANY` / `SQ_WAVE_CYCLES`; HBM-side MB = (FETCH_SIZE x 2 + WRITE_SIZE) KiB, the gfx950 correction for
16-B-per-lane readers, written to `traffic.json` under the bench.py name):

| kernel | bench.py name | cycles | MFMA busy | waves waiting | LDS conflict cycles | HBM-side MB / launch |
|---|---|---|---|---|---|---|
%s
""" % (tag, commit, open(table).read().strip(), "\n".join(lines)))
