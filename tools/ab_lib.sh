#!/bin/bash
# Dev tool: alternate two builds of libptrace (A/B) in separate processes on the same GPU.
# usage: tools/ab_lib.sh libA.so libB.so [passes...]
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for L in $A $B; do
    echo "== $L"; PT_LIB=$L python tools/ab_env.py PT_NONE unset "$@" | sed "s/PT_NONE=unset //"
  done
done
