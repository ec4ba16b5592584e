#!/bin/bash
# rocprofv3 kernel statistics of one python command on the GPU box:
#   scripts/rocprof_stats.sh OUTDIR NAME script.py [args...]
# -> OUTDIR/NAME_kernel_stats.csv (+ the program's stdout in OUTDIR/NAME.out).  The program itself
# follows `--` (python3 directly: no env / bash hop after the profiler's preload).
set -e
out=$(realpath -m "$1"); name=$2; shift 2
script=$(realpath "$1"); shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rocprof_$name
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rocprof_$name -o $name -- python3 "$script" "$@" > "$out/$name.out" 2>&1
f=$(find /tmp/rocprof_$name -name "*kernel_stats.csv" | head -n 1)
if [ -z "$f" ]; then echo "no kernel_stats.csv produced"; tail -n 5 "$out/$name.out"; exit 1; fi
cp "$f" "$out/${name}_kernel_stats.csv"
grep -v "rocprofv3\|simple_timer\|amdgpu.ids" "$out/$name.out" | tail -n 40
