#!/bin/bash
# usage: tools/ab_libs.sh "<bench command>" lib1 lib2 ...   ("default" = in-tree lib); two interleaved rounds on one box
CMD=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    if [ $lib = default ]; then unset SC_LIB; else export SC_LIB=$PWD/tools/bin/lib_$lib.so; fi
    echo "== $lib round $round: $($CMD 2>&1 | grep '^{' | python -c "
import sys, json
print(' '.join('%s=%s' % (json.loads(l)['name'], json.loads(l)['TFLOPs']) for l in sys.stdin))")"
  done
done
