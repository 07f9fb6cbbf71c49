#!/bin/bash
for m in 8 16 32; do
  FRLW_EXTRA_HIPCC_FLAGS="-DFRLW_SLICE_MULT=$m" python -c "
import sys; sys.path.insert(0,'.')
from frlw_evd_amd import _build; _build.build(force=True)" > /dev/null 2>&1
  echo "SLICE_MULT=$m: $(python tools/exp.py 2>&1 | grep -E 'GEN1 TAF|full' | tr '\n' ' ')"
done
