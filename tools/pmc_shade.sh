#!/bin/bash
# SQ / TCC / TCP counters of the shading kernel over one 64-pass batch of the bench workload (python bench.py --steps 64 --warmup 0), one
# rocprofv3 --pmc pass per counter set
R=$PWD; export TMPDIR=/tmp
D=$R/gpurun_out/pmcs; rm -rf $D; mkdir -p $D
B="python $R/bench.py --steps 64 --warmup 0 --no-cpu-baseline"
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES -d $D/sq -o p -- $B > $D/sq.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $D/tcc -o p -- $B > $D/tcc.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum -d $D/tcp -o p -- $B > $D/tcp.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $D/sq2 -o p -- $B > $D/sq2.log 2>&1
cd $R
python - <<PY
import os, sqlite3
for d in ("sq","tcc","tcp","sq2"):
    p=os.path.join("$D",d)
    fs=[os.path.join(r,x) for r,_,f in os.walk(p) for x in f if x.endswith(".db")]
    if not fs: print(d,"no db"); print(open("$D/%s.log"%d).read()[-800:]); continue
    cur=sqlite3.connect(fs[0]).cursor()
    q="select kernel_name,counter_name,count(*),avg(value),avg(duration) from counters_collection group by kernel_name,counter_name"
    for kn,cn,n,v,du in cur.execute(q):
        if "shade_kernel" in kn or ("trace_kernel<3, false" in kn):
            print(d, kn.split("(")[0][-30:], cn, n, "%.5g"%v, "%.1f us"%(du/1e3))
PY
find $D -name "*.db" -delete
