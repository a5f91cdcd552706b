#!/usr/bin/env python3
"""The N = 768 GEMMs of the step (192 tiles of 128x256 = 75 % of the CUs; 256 tiles of 128x192): own tiles vs library."""
import os
import sys

sys.argv = [sys.argv[0], "none"] + sys.argv[1:]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm2_probe as g  # noqa: E402

TL = ["128x192", "128x256"]
for k in (768, 3072):
    g.run("fwd", "nt", g.T, 768, k, TL)
for k in (768, 2304, 3072, 18432):
    g.run("dgrad", "nn", g.T, 768, k, TL)
