#!/bin/bash
# build_km_variant.sh NAME "extra kmeans.hip flags"  -> tools/bin/lib_NAME.so  (the other objects are the in-tree ones); load with SC_LIB=
set -e
N=$1; KF=$2; C=streamchat_amd/csrc; T=$(mktemp -d)
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off"
make -s -C $C > /dev/null 2>&1
/opt/rocm/bin/hipcc $F $KF -c $C/kmeans.hip -o $T/kmeans.o 2>/dev/null
OBJS=$(ls $C/*.o | grep -v "kmeans.o")
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o tools/bin/lib_$N.so $OBJS $T/kmeans.o
rm -rf $T; echo tools/bin/lib_$N.so
