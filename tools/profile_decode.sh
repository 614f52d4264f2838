#!/bin/bash
# Counter passes over the decode walk (k_decode_both: 4 blocks of 16 MiB, two streams each = 8 lone waves), to be
# run on the GPU box from the repo root:   gpurun --timeout 600 -- 'bash tools/profile_decode.sh r03'
# Two rocprofv3 --pmc passes (counters never share a run with another trace domain); the per-kernel sums go to
# gpurun_out/<tag>_pmc_decode_walk.csv, which tools/refresh_profiles.py does not touch: copy it to profiles/ by hand.
set -o pipefail
T=${1:-r03}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${T}_decpmc_a $O/${T}_decpmc_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O/${T}_decpmc_a -o d -- python3 $R/tools/decode_time.py 16 > $O/${T}_decpmc_a.log 2>&1 && echo "pass a ok" &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/${T}_decpmc_b -o d -- python3 $R/tools/decode_time.py 16 > $O/${T}_decpmc_b.log 2>&1 && echo "pass b ok" &&
python3 - $O/${T}_decpmc_a $O/${T}_decpmc_b $O/${T}_pmc_decode_walk.csv "$(grep -h ns/symbol $O/${T}_decpmc_a.log | tail -1)" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float)
launches = 0
for d in sys.argv[1:3]:
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    seen = set()
    for r in csv.DictReader(open(f)):
        if "k_decode" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            seen.add(r["Dispatch_Id"])
    launches = max(launches, len(seen))
with open(sys.argv[3], "w") as o:
    o.write("# k_decode_both, tools/decode_time.py 16 (4 blocks x 16 MiB, 8 lone waves), sums over %d launches; %s\n" % (launches, sys.argv[4]))
    o.write("counter,sum\n")
    for k in sorted(acc):
        o.write("%s,%.6g\n" % (k, acc[k]))
print(open(sys.argv[3]).read())
PY
