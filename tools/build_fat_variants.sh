#!/bin/bash
# timing ablations of k_attn_fat: tools/bin/lib_fat<N>.so = the in-tree library with attention_fat.hip compiled with -DFAT_ABL=<N>
# usage: tools/build_fat_variants.sh 1 2 4 ...   (then tools/ab_libs.sh "python tools/bench_attn.py llm49k" default fat1 fat2 ...)
cd "$(dirname "$0")/../streamchat_amd/csrc" || exit 1
make -s -j16 || exit 1
mkdir -p ../../tools/bin
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DFAT_ABL=$v -c attention_fat.hip -o /tmp/attention_fat_$v.o || exit 1
  objs=$(ls *.o | grep -v '^attention_fat.o$')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/bin/lib_fat$v.so $objs /tmp/attention_fat_$v.o || exit 1
  echo built lib_fat$v.so
done
