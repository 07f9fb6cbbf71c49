#!/usr/bin/env python3
"""`python generate_taf.py -raw_dir R -label_dir L -target_dir T -dataset gen1|gen4` -- the reference's offline
pre-processing command (generate_taf.py:78-243, README.md:56-73) on the gfx950 encoders: same flags, same output tree, same files
(frlw_evd_amd/generate.py; pinned by tests/golden/harness.npz, written by the reference's own script)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from frlw_evd_amd import generate  # noqa: E402

if __name__ == "__main__":
    generate.main("taf")
