#!/bin/bash
# tools/buildvar.sh <tag> <file.hip> "<extra flags>": variant build of one kernel file linked with the current objects -> ab/libairlift_<tag>.so
TAG=$1; F=$2; X=$3
cd /root/repo/airlift_amd/csrc && mkdir -p ../../ab/obj_$TAG
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -mllvm -two-entry-phi-node-folding-threshold=200 $X -c $F -o ../../ab/obj_$TAG/$F.o || exit 1
OBJS=""; for o in build/*.o; do b=$(basename $o); if [ "$b" == "$F.o" ]; then OBJS="$OBJS ../../ab/obj_$TAG/$F.o"; else OBJS="$OBJS $o"; fi; done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab/libairlift_$TAG.so $OBJS -lz -lpthread && echo built ab/libairlift_$TAG.so
