#!/bin/bash
# PMC passes on the dominant kernels (separate passes: SQ/GRBM, LDS, FETCH_SIZE, WRITE_SIZE).
# usage (on the GPU box, from the repo root): bash scripts/pmc_s3.sh OUTDIR kernel [kernel ...]
set -e
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $R/$OUT/$name -o $name --output-format csv -- python3 $R/scripts/bench_kernels.py $KERNELS --iters 3 > $R/$OUT/$name.log 2>&1; }
KERNELS="$*"
cd $R
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 scripts/pmc_table.py $OUT/sq $OUT/lds $OUT/fetch $OUT/write > $OUT/table.md
cat $OUT/table.md
