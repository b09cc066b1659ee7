cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-heterogeneous --no-poseidon-gl-config --no-poseidon-gl"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof_r2n/fetch -o r2n -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof_r2n/write -o r2n -- $CMD > /dev/null 2>&1
for d in fetch write; do f=$(find gpurun_out/prof_r2n/$d -name "*.db" | head -1); python tools/rocprof_summary.py $f --pmc > gpurun_out/r2n_pmc_$d.txt; done
rm -rf gpurun_out/prof_r2n
grep -E "^k_merkle|^k_crown_level|^k_range" gpurun_out/r2n_pmc_fetch.txt gpurun_out/r2n_pmc_write.txt
