#!/bin/bash
# Build-time sweep of the conv kernel's BK on the GPU box (hipcc is available there too).
for cfg in "16 16" "32 16" "16 32" "32 32"; do
  set -- $cfg
  FRLW_EXTRA_HIPCC_FLAGS="-DCONV_BK_BIG=$1 -DCONV_BK_SMALL=$2" python -c "
import sys; sys.path.insert(0,'.')
from frlw_evd_amd import _build; _build.build(force=True)" > /dev/null 2>&1
  echo "BK_BIG=$1 BK_SMALL=$2: $(python tools/exp_det.py 2>&1 | grep 'engine fwd  B')"
done
