#!/bin/bash
# per-kernel times of the 512^3 dsmag step under different environment settings:  bash tools/envab.sh "CALES_KCHUNK=64" "CALES_GAUSSEL_MARCH=1" ...
for e in "" "$@" ""; do
  echo "== ${e:-default}"; env $e python3 tools/opbench.py --ops step --reps 6 2>&1 | tail -1 | cut -c1-330
done
