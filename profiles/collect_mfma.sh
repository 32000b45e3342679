#!/bin/bash
# MFMA-utilisation counters of the large-front update kernels (north_star: "rocprof counters ... MFMA utilisation on
# large-front updates"), one counter per --pmc pass as MI355X_MICROARCH.md prescribes:
#   bash profiles/collect_mfma.sh r03_a [workload]        (through gpurun, from the repo root)
# Writes gpurun_out/<tag>_<workload>_mfma_counters_per_kernel.json (raw sums per kernel) and ..._mfma_summary.txt.
# Copy both into profiles/ and commit them.
set -e
TAG=${1:-rxx}
W=${2:-c5mid_standin}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
DIRS=""
for C in SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE; do
  D=$OUT/${TAG}_pmc_$C
  rm -rf $D
  if rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o p -- python3 $ROOT/bench.py --workload $W --steps 1 --warmup 0 --no-cpu > $OUT/${TAG}_pmc_$C.log 2>&1; then
    DIRS="$DIRS $D"; echo "$C ok"
  else
    echo "$C FAILED (see ${TAG}_pmc_$C.log)"; tail -3 $OUT/${TAG}_pmc_$C.log
  fi
done
python3 $ROOT/profiles/summarize_pmc.py $OUT/${TAG}_${W}_mfma_counters_per_kernel.json $DIRS > $OUT/${TAG}_${W}_mfma_summary_raw.txt
python3 $ROOT/profiles/mfma_util.py $OUT/${TAG}_${W}_mfma_counters_per_kernel.json > $OUT/${TAG}_${W}_mfma_summary.txt
cat $OUT/${TAG}_${W}_mfma_summary.txt
for C in SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE; do rm -rf $OUT/${TAG}_pmc_$C; done
