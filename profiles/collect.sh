#!/bin/bash
# Collects the judged artefacts of one round on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh r01_d
# bench lines (xenon1 stand-in with the CPU baseline, epb1), the rocprofv3 kernel-trace summary of the bench command, and
# the FETCH_SIZE / WRITE_SIZE counters in two separate --pmc passes.  Results land in gpurun_out/<tag>_*; copy them to
# profiles/ afterwards.
set -e
TAG=${1:-rxx}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --steps 5 --warmup 2 > $OUT/${TAG}_bench_xenon1_standin.json 2> $OUT/${TAG}_bench_xenon.err
python3 $ROOT/bench.py --workload epb1 --steps 5 --warmup 2 > $OUT/${TAG}_bench_epb1.json 2> $OUT/${TAG}_bench_epb1.err
python3 $ROOT/bench.py --workload sme3dc_standin --steps 5 --warmup 2 > $OUT/${TAG}_bench_sme3dc_standin.json 2> $OUT/${TAG}_bench_sme3dc.err
python3 $ROOT/bench.py --workload c5mini_standin --steps 5 --warmup 2 > $OUT/${TAG}_bench_c5mini_standin.json 2> $OUT/${TAG}_bench_c5mini.err
python3 $ROOT/bench.py --workload micro --steps 3 --warmup 1 > $OUT/${TAG}_bench_micro.json 2> $OUT/${TAG}_bench_micro.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o p -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu > $OUT/${TAG}_prof.log 2>&1
cp $OUT/${TAG}_prof/p_kernel_stats.csv $OUT/${TAG}_xenon1_standin_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -o p -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu > $OUT/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -o p -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu > $OUT/${TAG}_pmc_write.log 2>&1
python3 $ROOT/profiles/summarize_pmc.py $OUT/${TAG}_pmc_fetch_write_per_kernel.json $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write > $OUT/${TAG}_pmc_summary.txt
rm -rf $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_prof
tail -c 600 $OUT/${TAG}_bench_xenon1_standin.json; echo; head -5 $OUT/${TAG}_xenon1_standin_kernel_stats.csv; cat $OUT/${TAG}_pmc_summary.txt | head -40
