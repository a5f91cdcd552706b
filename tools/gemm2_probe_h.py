#!/usr/bin/env python3
"""Round 5: the four-wave 128x192 tile that runs TWO workgroups per CU (128x192h) against the tiles the engine uses today, on the
products with at least two tiles per CU (interleaved rounds, one process)."""
import os
import sys

sys.argv = [sys.argv[0], "none"] + sys.argv[1:]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm2_probe as g  # noqa: E402

for n, k, tl in [(2304, 768, ["128x192h", "128x192p", "256x192", "128x192"]), (3072, 768, ["128x192h", "256x192", "128x256", "128x192"]),
                 (18432, 768, ["128x192h", "256x256p", "256x256"]), (30528, 768, ["128x192h", "256x256"]),
                 (768, 768, ["128x192h", "128x192", "64x128"]), (768, 3072, ["128x192h", "128x192"])]:
    g.run("fwd", "nt", g.T, n, k, tl)
for n, k, tl in [(3072, 768, ["128x192h", "256x192", "128x256"]), (768, 768, ["128x192h", "128x192"]), (768, 2304, ["128x192h", "128x192"]),
                 (768, 3072, ["128x192h", "128x192"])]:
    g.run("dgrad", "nn", g.T, n, k, tl)
for m, n, tl in [(18432, 768, ["128x192h", "256x256"]), (30528, 768, ["128x192h", "128x256"])]:
    g.run("wgrad", "tn", m, n, g.T, tl)
