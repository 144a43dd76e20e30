#!/bin/bash
# per-kernel times of the 512^3 dsmag step for the default library and every tools/variants/libcales_*.so named on the command line (timing experiments:
# variants may compute wrong fields -- nothing is validated here):  bash tools/abvariants.sh NOUN NOP ...
for v in base "$@" base; do
  if [ $v = base ]; then unset CALES_LIB; else export CALES_LIB=$PWD/tools/variants/libcales_$v.so; fi
  echo "== $v"; python3 tools/opbench.py --ops step --reps 6 2>&1 | tail -1 | cut -c1-420
done
