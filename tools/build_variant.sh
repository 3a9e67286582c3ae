#!/bin/bash
# build_variant.sh NAME "attention_decode flags" "gemv flags"  -> tools/bin/lib_NAME.so  (the other objects are the in-tree ones)
set -e
N=$1; AF=$2; GF=$3; C=streamchat_amd/csrc; T=$(mktemp -d)
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
make -s -C $C > /dev/null
/opt/rocm/bin/hipcc $F $AF -c $C/attention_decode.hip -o $T/attention_decode.o
/opt/rocm/bin/hipcc $F $GF -c $C/gemv.hip -o $T/gemv.o
OBJS=$(ls $C/*.o | grep -v "attention_decode.o\|gemv.o")
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o tools/bin/lib_$N.so $OBJS $T/attention_decode.o $T/gemv.o
rm -rf $T; echo tools/bin/lib_$N.so
