#!/bin/bash
# tools/build_variant.sh NAME "EXTRA_FLAGS" [source ...]   -- an experimental build of the library: the named kernel sources (default fpt_trace.hip) are compiled with
# EXTRA_FLAGS and linked with the standard objects into fermat_amd/variants/libfermat_NAME.so (git-ignored; travels to the GPU box).  Run a bench against it with
# FPT_LIB_PATH=fermat_amd/variants/libfermat_NAME.so python bench.py ...
set -e
cd "$(dirname "$0")/../fermat_amd/csrc"
NAME=$1; FLAGS=$2; shift 2 || true
SRCS=${@:-fpt_trace.hip}
[ -n "$NO_MAKE" ] || make -s -j8 all >/dev/null
mkdir -p ../variants ../../tools/_build/$NAME
STD="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wall -Wno-unused-function -Wno-unused-variable"
OBJS="fpt_trace.o fpt_pt.o fpt_filter.o fpt_bpt.o fpt_build.o fpt_build_lbvh.o fpt_api.o fpt_comm.o fpt_bpt_api.o fpt_psf_api.o fpt_bvh.o fpt_sequence.o fpt_lights.o host/fpt_renderer.o host/scene_io.o"
for s in $SRCS; do
	o=../../tools/_build/$NAME/$(basename ${s%.*}).o
	x=""; case $s in *.cpp) x="-x hip";; esac
	hipcc --offload-arch=gfx950 $STD $FLAGS $x -c $s -o $o &
	OBJS=$(echo $OBJS | sed "s#\b${s%.*}.o#$o#")
done
wait
hipcc --offload-arch=gfx950 -shared -o ../variants/libfermat_$NAME.so $OBJS -ldl
echo built fermat_amd/variants/libfermat_$NAME.so
