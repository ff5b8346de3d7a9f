# diagnostic counters for the traversal / shade kernels (separate passes; --kernel-trace only)
set -x
R=$PWD
export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc; mkdir -p $R/gpurun_out/pmc
cd /tmp
rocprofv3 -L > $R/gpurun_out/pmc/counters.txt 2>&1
B="python $R/bench.py --steps 64 --warmup 0 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES -d $R/gpurun_out/pmc/sq -o p -- $B > $R/gpurun_out/pmc/sq.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $R/gpurun_out/pmc/tcc -o p -- $B > $R/gpurun_out/pmc/tcc.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum -d $R/gpurun_out/pmc/tcp -o p -- $B > $R/gpurun_out/pmc/tcp.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM -d $R/gpurun_out/pmc/sq2 -o p -- $B > $R/gpurun_out/pmc/sq2.log 2>&1
cd $R
python - <<'PY'
import os, sqlite3
for d in ("sq","tcc","tcp","sq2"):
    p=os.path.join("gpurun_out/pmc",d)
    fs=[os.path.join(r,x) for r,_,f in os.walk(p) for x in f if x.endswith(".db")]
    if not fs: print(d,"no db"); print(open("gpurun_out/pmc/%s.log"%d).read()[-1500:]); continue
    cur=sqlite3.connect(fs[0]).cursor()
    q="select kernel_name,counter_name,count(*),avg(value),avg(duration) from counters_collection group by kernel_name,counter_name"
    for kn,cn,n,v,du in cur.execute(q):
        if "trace_kernel" in kn or "shade_kernel" in kn:
            print(d, kn.split("(")[0][-40:], cn, n, "%.4g"%v, "%.1f us"%(du/1e3))
PY
