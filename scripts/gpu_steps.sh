#!/bin/bash
# Run GPU steps one after the other on a gpurun box: `scripts/gpu_steps.sh OUTDIR "name|seconds|command" ...`
# Each step runs under `timeout -k 10`, its output goes to OUTDIR/name.log; a step that times out
# or is killed ends the sequence (no further GPU work after a hang); a step that merely fails does not.
out=$1; shift
mkdir -p "$out"
rc_all=0
for step in "$@"; do
  name=${step%%|*}; rest=${step#*|}; secs=${rest%%|*}; cmd=${rest#*|}
  echo "== $name (limit ${secs}s): $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > "$out/$name.log" 2>&1
  rc=$?
  echo "== $name rc=$rc"; tail -n 6 "$out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== $name timed out / was killed: stopping"; exit $rc; fi
  [ $rc -ne 0 ] && rc_all=$rc
done
exit $rc_all
