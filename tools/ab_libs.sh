#!/bin/bash
# Dev tool: time several libptrace builds on one box with a fixed geometry path
# (1 = LDS, 2 = scalar, 3 = hierarchy).  usage: ab_libs.sh PATH lib1.so lib2.so ... -- passes...
G=$1; shift
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for rep in 1 2; do
  for L in "${LIBS[@]}"; do
    echo "== $L"; PT_LIB=$L PT_GEOM=$G python tools/ab_env.py PT_NONE unset "$@" | sed "s/PT_NONE=unset //"
  done
done
