import os, sys
sys.argv = ["x", "none"]
exec(open("tools/gemm2_probe.py").read().split("TL = [")[0])
TL = ["256x192", "256x256", "128x256"]
run("fwd", "nt", T, 3072, 768, TL)
run("fwd", "nt", T, 18432, 768, TL)
run("fwd", "nt", T, 30528, 768, TL)
run("dgrad", "nn", T, 3072, 768, TL)
run("wgrad", "tn", 18432, 768, T, TL)
run("wgrad", "tn", 30528, 768, T, TL)
