#!/bin/bash
# Same-box A/B of kernel builds: scripts/ab_builds.sh "<python command>" NAME=HIPCC_EXTRA ...
# builds dsmnet_amd/csrc/libdsmnet_hip_NAME.so for each variant (here, before the GPU run:
#   scripts/ab_builds.sh --build NAME="-DFOO=1" ...), then on the box runs the command once per
# variant with DSM_LIB_PATH pointing at it.
set -e
cd "$(dirname "$0")/.."
if [ "$1" == "--build" ]; then
  shift
  for v in "$@"; do
    name=${v%%=*}; extra=${v#*=}
    HIPCC_EXTRA="$extra" python dsmnet_amd/csrc/build.py --force > /dev/null
    cp dsmnet_amd/csrc/libdsmnet_hip.so dsmnet_amd/csrc/libdsmnet_hip_$name.so
    echo "built $name ($extra)"
  done
  python dsmnet_amd/csrc/build.py --force > /dev/null      # the default build back in place
  exit 0
fi
cmd=$1; shift
for name in "$@"; do
  echo "== $name"
  DSM_LIB_PATH=$PWD/dsmnet_amd/csrc/libdsmnet_hip_$name.so bash -c "$cmd"
done
