#!/bin/bash
# rocprofv3 kernel trace (start / end timestamps of every dispatch) of one python command on the GPU box:
#   scripts/rocprof_trace.sh OUTDIR NAME script.py [args...]   -> OUTDIR/NAME_kernel_trace.csv.gz
set -e
out=$(realpath -m "$1"); name=$2; shift 2
script=$(realpath "$1"); shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rocprof_$name
rocprofv3 --kernel-trace --output-format csv -d /tmp/rocprof_$name -o $name -- python3 "$script" "$@" > "$out/$name.out" 2>&1
f=$(find /tmp/rocprof_$name -name "*kernel_trace.csv" | head -n 1)
if [ -z "$f" ]; then echo "no kernel_trace.csv produced"; tail -n 5 "$out/$name.out"; exit 1; fi
gzip -c "$f" > "$out/${name}_kernel_trace.csv.gz"
ls -la "$out/${name}_kernel_trace.csv.gz"
