#!/bin/bash
# usage: tools/build_variant.sh <suffix> "<extra CXXFLAGS>" [TRACE_OBJ]  -> fermat_amd/libfermat_pt_hip_<suffix>.so (a tuning variant of the
# product library built in a scratch copy of csrc/; select it at run time with FPT_LIB_PATH=fermat_amd/libfermat_pt_hip_<suffix>.so)
set -e
R=$(cd $(dirname $0)/.. && pwd)
S=$1; FLAGS=$2; TO=${3:-fpt_trace.o}
B=/tmp/fpt_build_$S
rm -rf $B; mkdir -p $B/fermat_amd $B/include; cp -r $R/fermat_amd/csrc $B/fermat_amd/; cp $R/include/*.h $B/include/
find $B -name "*.o" -delete
make -s -C $B/fermat_amd/csrc -j8 TRACE_OBJ=$TO LIBNAME=libfermat_pt_hip_$S.so ../libfermat_pt_hip_$S.so \
  CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -w $FLAGS"
cp $B/fermat_amd/libfermat_pt_hip_$S.so $R/fermat_amd/
ls -la $R/fermat_amd/libfermat_pt_hip_$S.so
