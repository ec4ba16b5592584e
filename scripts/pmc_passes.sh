#!/bin/bash
# PMC passes (separate passes, counters only with --kernel-trace: SQ/GRBM, LDS, FETCH_SIZE, WRITE_SIZE)
# over one python command, reduced to a per-kernel table of mean counter values per dispatch:
#   scripts/pmc_passes.sh OUTDIR script.py [args...]      (on the GPU box, from the repo root)
# FETCH_SIZE / WRITE_SIZE are in KiB-like units of the tool; on gfx950 FETCH_SIZE counts 64 B per
# 128-B request of a wide streaming read: double it (MI355X_MICROARCH.md, HBM).
set -e
out=$(realpath -m "$1"); shift
script=$(realpath "$1"); shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  rm -rf /tmp/pmc_$name
  rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_$name -o $name --output-format csv -- python3 "$script" $ARGS > "$out/$name.log" 2>&1
  mkdir -p "$out/$name" && find /tmp/pmc_$name -name "*counter_collection.csv" -exec cp {} "$out/$name/" \;
}
ARGS="$*"
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 "$root/scripts/pmc_table.py" "$out/sq" "$out/lds" "$out/fetch" "$out/write" > "$out/table.md"
cat "$out/table.md"
