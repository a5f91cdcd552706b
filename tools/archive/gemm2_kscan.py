#!/usr/bin/env python3
"""Fixed cost of a launch: [8192, 768] outputs on the 128x192 tile (256 tiles, one per CU) over the contraction depth."""
import os
import sys

sys.argv = [sys.argv[0], "none"]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm2_probe as g  # noqa: E402

for k in (64, 128, 256, 384, 512, 768, 1536, 3072):
    g.run("fwd", "nt", g.T, 768, k, ["128x192"])
for k in (64, 256, 768):
    g.run("dgrad", "nn", g.T, 768, k, ["128x192"])
