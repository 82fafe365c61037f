#!/bin/bash
# Dev tool: A/B two libptrace builds on several configs (tools/ab_geom.py prints LDS and scalar times).
A=$1; B=$2; shift 2
for rep in 1 2; do
  for L in $A $B; do echo "== $L"; PT_LIB=$L python tools/ab_geom.py "$@"; done
done
