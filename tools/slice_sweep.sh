#!/bin/bash
for m in 4 8 16; do
  FRLW_EXTRA_HIPCC_FLAGS="-DFRLW_SLICE_MULT=$m" python -c "
import sys; sys.path.insert(0,'.')
from frlw_evd_amd import _build; _build.build(force=True)" > /dev/null 2>&1
  for t in 6 7 8; do echo "SLICE_MULT=$m TWL=$t: $(FRLW_TWL=$t python tools/exp2.py 2>&1 | tail -1)"; done
done
