#!/bin/bash
# Runs on the GPU box (via gpurun): the rocprofv3 passes whose summaries are committed under profiles/.
#   tools/profile_round.sh r03p            -> gpurun_out/r03p/{kernel_stats,pmc_sq,pmc_fetch,pmc_write}.txt + bench_under_rocprof.json + traffic.json
# Kernel trace + stats in one pass; every PMC set in its own pass with --kernel-trace only (the pool refuses --pmc together with
# the HIP/HSA trace domains). The bench is cut down to the headline step (no CPU baseline, no side legs).
set -u
TAG=${1:-prof}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-heterogeneous --no-poseidon-gl-config --no-poseidon-gl --no-config-legs --no-clock-sample --no-exchange-probe"
summ() { db=$(find $1 -name '*_results.db' | head -1); [ -n "$db" ] && python $ROOT/tools/rocprof_summary.py $db ${2:-}; }
cd /tmp
( echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 <headline step only>   (MI355X, $TAG)"
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o p -- $B --steps 5 --warmup 1 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
  summ $OUT/stats "--skip-first 1" ) > $OUT/kernel_stats.txt
( echo "# rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU --kernel-trace -- python bench.py --steps 1 --warmup 1 <headline step only>   (MI355X, $TAG; 2 pipeline passes)"
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU --kernel-trace -d $OUT/sq -o p -- $B --steps 1 --warmup 1 > /dev/null 2> $OUT/sq.err
  summ $OUT/sq --pmc ) > $OUT/pmc_sq.txt
( echo "# rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 1 --warmup 1 <headline step only>   (MI355X, $TAG; KB, raw: x2 on gfx950)"
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o p -- $B --steps 1 --warmup 1 > /dev/null 2> $OUT/fetch.err
  summ $OUT/fetch --pmc ) > $OUT/pmc_fetch.txt
( echo "# rocprofv3 --pmc WRITE_SIZE --kernel-trace -- python bench.py --steps 1 --warmup 1 <headline step only>   (MI355X, $TAG; KB)"
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o p -- $B --steps 1 --warmup 1 > /dev/null 2> $OUT/write.err
  summ $OUT/write --pmc ) > $OUT/pmc_write.txt
# the same SQ pass with every kernel alone on one stream (GPV_OPT_SIDE_STREAM = 0): what k_merkle_leaves costs without the side stream underneath it
( echo "# rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python bench.py --no-side-stream --steps 1 --warmup 1 <headline step only>   (MI355X, $TAG)"
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/sq1 -o p -- $B --no-side-stream --steps 1 --warmup 1 > /dev/null 2> $OUT/sq1.err
  summ $OUT/sq1 --pmc ) > $OUT/pmc_sq_no_side_stream.txt
( echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-side-stream --steps 5 --warmup 1 <headline step only>   (MI355X, $TAG)"
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats1 -o p -- $B --no-side-stream --steps 5 --warmup 1 > $OUT/bench_no_side_stream_under_rocprof.json 2> $OUT/stats1.err
  summ $OUT/stats1 "--skip-first 1" ) > $OUT/kernel_stats_no_side_stream.txt
# HBM traffic of every kernel of the step, stamped with the build id of the library that was just profiled (bench.py checks it)
python $ROOT/tools/make_traffic_json.py $OUT $TAG step 8192 > $OUT/traffic.json
rm -rf $OUT/stats $OUT/sq $OUT/fetch $OUT/write $OUT/sq1 $OUT/stats1
for f in $OUT/*.err; do tail -n 2 $f; done
