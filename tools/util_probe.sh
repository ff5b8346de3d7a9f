#!/bin/bash
# usage: tools/util_probe.sh "<extra CXXFLAGS>" : VALU lane utilisation (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU x 4... see below) of the traversal kernel
cd $GRAFT_REPO_ROOT
rm -f fermat_amd/csrc/fpt_trace.o
make -s -C fermat_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -w $1" >/dev/null 2>&1
export TMPDIR=/tmp; R=$PWD; rm -rf $R/gpurun_out/util; mkdir -p $R/gpurun_out/util; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES -d $R/gpurun_out/util -o u -- python $R/bench.py --steps 64 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
cd $R
python - <<PY
import os, sqlite3
fs=[os.path.join(r,x) for r,_,f in os.walk("gpurun_out/util") for x in f if x.endswith(".db")]
cur=sqlite3.connect(fs[0]).cursor()
v={}
for kn,cn,n,val,du in cur.execute("select kernel_name,counter_name,count(*),avg(value),avg(duration) from counters_collection group by kernel_name,counter_name"):
    if "trace_kernel<3, false>" in kn: v[cn]=val; v["us"]=du/1e3
print("flags=[$1] MIXED launch %.0f us  INSTS_VALU %.3e  lane utilisation %.1f %%" % (v["us"], v["SQ_INSTS_VALU"], 100.0*v["SQ_THREAD_CYCLES_VALU"]/(v["SQ_ACTIVE_INST_VALU"]*4*64)))
PY
