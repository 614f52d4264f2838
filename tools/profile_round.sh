#!/bin/bash
# The profile passes of one round, to be run on the GPU box in ONE gpurun call from the repo root:
#   git rev-parse --short=12 HEAD > profiles/.head_commit; gpurun --timeout 1150 -- 'bash tools/profile_round.sh r04'
# Every pass is its own rocprofv3 run (counters never share a run with another trace domain); the
# summaries land under gpurun_out/<tag>_*/.  Behind passes 1-4 tools/make_traffic_json.py writes profiles/<tag>_traffic.json ON THE
# BOX, and the default bench.py run of the same call comes AFTER it, so the default line quotes its own round's file;
# tools/refresh_profiles.py (here, afterwards) copies everything into profiles/ and checks the quoted numbers.
set -o pipefail
T=${1:-r04}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --skip-cpu --skip-decode --skip-host --skip-other-data --skip-strong"
rm -rf $O/${T}_prof4 $O/${T}_prof1 $O/${T}_pmc_f $O/${T}_pmc_w $O/${T}_pmc_sq $O/${T}_pmc_insts
# 1. per-kernel time of the timed encode region, default lanes (4 blocks in flight)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof4 -o p4 -- $B > $O/${T}_prof4.log 2>&1 && echo "prof4 ok" &&
# 2. the same with one block in flight
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof1 -o p1 -- $B --lanes 1 --steps 2 > $O/${T}_prof1.log 2>&1 && echo "prof1 ok" &&
# 3./4. HBM bytes per launch: FETCH_SIZE and WRITE_SIZE in separate passes
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${T}_pmc_f -o f -- $B --steps 1 --warmup 1 > $O/${T}_pmc_f.log 2>&1 && echo "pmc_f ok" &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${T}_pmc_w -o w -- $B --steps 1 --warmup 1 > $O/${T}_pmc_w.log 2>&1 && echo "pmc_w ok" &&
python3 $R/tools/make_traffic_json.py $T > $O/${T}_traffic.log 2>&1 && echo "traffic json ok" &&
# the default run (what the driver runs), quoting the file just written
python3 $R/bench.py > $O/${T}_bench_default.log 2> $O/${T}_bench_default.err && echo "default ok" &&
# 5. wave-cycle breakdown, one block in flight
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/${T}_pmc_sq -o sq -- $B --steps 1 --warmup 1 --lanes 1 > $O/${T}_pmc_sq.log 2>&1 && echo "pmc_sq ok" &&
# 6. wave-instructions by class, one block in flight
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES --kernel-trace --output-format csv -d $O/${T}_pmc_insts -o insts -- $B --steps 1 --warmup 1 --lanes 1 > $O/${T}_pmc_insts.log 2>&1 && echo "pmc_insts ok"
ls $O/${T}_prof4 $O/${T}_pmc_f | head -20
