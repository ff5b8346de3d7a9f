#!/bin/bash
# round-2 GPU call A: the whole -m gpu suite (incl. the full-size configurations), the driver-form bench line, the default bench line,
# the harder scene, and the self-launched N=2 path dry-run on one GPU (gloo)
mkdir -p gpurun_out/r02a
( time python -m pytest tests -m gpu -x -q -s ) > gpurun_out/r02a/tests.log 2>&1
tail -5 gpurun_out/r02a/tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r02a/bench_driver.json 2> gpurun_out/r02a/bench_driver.err
python bench.py --no-cpu-baseline > gpurun_out/r02a/bench_default.json 2> gpurun_out/r02a/bench_default.err
python bench.py --workload testball-room --no-cpu-baseline > gpurun_out/r02a/bench_testball.json 2> gpurun_out/r02a/bench_testball.err
FPT_BENCH_FORCE_DEVICE=0 FPT_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r02a/bench_n2_gloo.json 2> gpurun_out/r02a/bench_n2_gloo.err
for f in bench_driver bench_default bench_testball bench_n2_gloo; do echo == $f; tail -c 1500 gpurun_out/r02a/$f.json; tail -3 gpurun_out/r02a/$f.err; done
