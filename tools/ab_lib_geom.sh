#!/bin/bash
# Dev tool: A/B two libptrace builds with a fixed geometry path (1 = LDS, 2 = scalar).
A=$1; B=$2; G=$3; shift 3
for rep in 1 2 3; do
  for L in $A $B; do
    echo "== $L"; PT_LIB=$L PT_GEOM=$G python tools/ab_env.py PT_NONE unset "$@" | sed "s/PT_NONE=unset //"
  done
done
