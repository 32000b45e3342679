#!/bin/bash
# Collects the judged artefacts of one round on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh r02_a [pmc|bench]
# "pmc":   the rocprofv3 kernel-trace summary of the default bench command and the FETCH_SIZE / WRITE_SIZE counters in two
#          separate --pmc passes (default workload and the large-front workload) -> copy the *_pmc_fetch_write_per_kernel.json
#          and *_kernel_stats.csv files to profiles/ and commit them: bench.py reads the newest one for `roofline.traffic`.
# "bench": the bench lines of every workload (CPU baseline legs included).
# Results land in gpurun_out/<tag>_*.
set -e
TAG=${1:-rxx}
WHAT=${2:-all}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
DEF=xenon1_colamd_standin
if [ "$WHAT" = pmc ] || [ "$WHAT" = all ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o p -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-large-front > $OUT/${TAG}_prof.log 2>&1
  cp $OUT/${TAG}_prof/p_kernel_stats.csv $OUT/${TAG}_${DEF}_kernel_stats.csv
  rm -rf $OUT/${TAG}_prof
  for W in $DEF c5mid_standin; do
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -o p -- python3 $ROOT/bench.py --workload $W --steps 1 --warmup 0 --no-cpu --no-large-front > $OUT/${TAG}_pmc_fetch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -o p -- python3 $ROOT/bench.py --workload $W --steps 1 --warmup 0 --no-cpu --no-large-front > $OUT/${TAG}_pmc_write.log 2>&1
    python3 $ROOT/profiles/summarize_pmc.py $OUT/${TAG}_${W}_pmc_fetch_write_per_kernel.json $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write > $OUT/${TAG}_${W}_pmc_summary.txt
    rm -rf $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write
  done
  head -12 $OUT/${TAG}_${DEF}_kernel_stats.csv
fi
if [ "$WHAT" = bench ] || [ "$WHAT" = all ]; then
  for W in $DEF xenon1_standin sme3dc_standin c5mini_standin epb1; do
    python3 $ROOT/bench.py --workload $W --steps 5 --warmup 2 > $OUT/${TAG}_bench_$W.json 2> $OUT/${TAG}_bench_$W.err
    tail -c 400 $OUT/${TAG}_bench_$W.json; echo
  done
  # (the reference needs ~5 minutes per run on this one: its time in the build container is in the fixture, fac_seconds)
  python3 $ROOT/bench.py --workload c5mid_standin --steps 3 --warmup 1 --no-cpu > $OUT/${TAG}_bench_c5mid_standin.json 2> $OUT/${TAG}_bench_c5mid_standin.err
  tail -c 400 $OUT/${TAG}_bench_c5mid_standin.json; echo
  python3 $ROOT/bench.py --workload micro --steps 3 --warmup 1 > $OUT/${TAG}_bench_micro.json 2> $OUT/${TAG}_bench_micro.err
  # configs[4] at full size (52 022 columns): ~3 s per factorization
  python3 $ROOT/bench.py --workload c5_standin --steps 2 --warmup 1 --no-cpu > $OUT/${TAG}_bench_c5_standin.json 2> $OUT/${TAG}_bench_c5_standin.err
  tail -c 400 $OUT/${TAG}_bench_c5_standin.json; echo
fi
