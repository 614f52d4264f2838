#!/bin/bash
# rocprofv3 kernel-trace summaries of the timed encode region for several builds of the library, same box:
#   bash tools/prof_ab.sh <tag> name=lib.so ...   ->  gpurun_out/<tag>_<name>_{4,1}/  (4 lanes / 1 lane)
set -o pipefail
T=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for nv in "$@"; do
  n=${nv%%=*}; l=${nv#*=}
  export FQGPU_LIB=$R/$l
  rm -rf $O/${T}_${n}_4 $O/${T}_${n}_1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_${n}_4 -o p -- python3 $R/bench.py --skip-cpu --skip-decode --skip-host --skip-other-data --skip-strong > $O/${T}_${n}_4.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_${n}_1 -o p -- python3 $R/bench.py --skip-cpu --skip-decode --skip-host --skip-other-data --skip-strong --lanes 1 --steps 2 > $O/${T}_${n}_1.log 2>&1 || exit 1
  echo "$n done"
done
