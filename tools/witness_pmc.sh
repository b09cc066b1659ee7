#!/bin/bash
# PMC passes over gpv_witness_verify_dev at 4096 `step` proofs (on the GPU box): where the witness kernels' cycles go.
#   tools/witness_pmc.sh <tag>  ->  gpurun_out/<tag>/witness_pmc_{sq,mem}.txt
set -u
TAG=${1:-wpmc}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
summ() { db=$(find $1 -name '*_results.db' | head -1); [ -n "$db" ] && python $ROOT/tools/rocprof_summary.py $db --pmc; }
cd /tmp
( echo "# rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU --kernel-trace -- python tools/witness_rate.py --only 4096   (MI355X, $TAG)"
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU --kernel-trace -d $OUT/a -o p -- python $ROOT/tools/witness_rate.py --only 4096 > /dev/null 2> $OUT/a.err
  summ $OUT/a ) > $OUT/witness_pmc_sq.txt
( echo "# rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM --kernel-trace -- python tools/witness_rate.py --only 4096   (MI355X, $TAG)"
  timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM --kernel-trace -d $OUT/b -o p -- python $ROOT/tools/witness_rate.py --only 4096 > /dev/null 2> $OUT/b.err
  summ $OUT/b ) > $OUT/witness_pmc_mem.txt
rm -rf $OUT/a $OUT/b
tail -n 3 $OUT/a.err $OUT/b.err
