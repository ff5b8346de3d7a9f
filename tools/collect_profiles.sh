set -x
R=$PWD
export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof; mkdir -p $R/gpurun_out/prof
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/stats -o r01 -- python $R/bench.py --warmup 64 --no-cpu-baseline > $R/gpurun_out/prof/bench_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof/pmc_fetch -o r01 -- python $R/bench.py --steps 64 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof/pmc_write -o r01 -- python $R/bench.py --steps 64 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof/bench_write.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/filter -o r01 -- python $R/tools/bench_filter.py > $R/gpurun_out/prof/bench_filter.log 2>&1
for K in bpt psfpt; do
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof/pmc_fetch_$K -o r01 -- python $R/bench.py --renderer $K --steps 16 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof/pmc_write_$K -o r01 -- python $R/bench.py --renderer $K --steps 16 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
done
cd $R
python bench.py --warmup 64 > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err
tail -1 gpurun_out/bench_r01.json
find gpurun_out/prof -name "*.db" | xargs ls -la
python bench.py --renderer bpt > gpurun_out/bench_r01_bpt.json 2> gpurun_out/bench_r01_bpt.err
python bench.py --renderer psfpt > gpurun_out/bench_r01_psfpt.json 2> gpurun_out/bench_r01_psfpt.err
tail -c 600 gpurun_out/bench_r01_bpt.json; tail -c 600 gpurun_out/bench_r01_psfpt.json
