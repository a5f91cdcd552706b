import os, sys
sys.argv = ["x", "none"]
exec(open("tools/gemm2_probe.py").read().split("TL = [")[0])
TL = ["256x192", "256x256"]
run("fwd", "nt", T, 3072, 768, TL)
run("fwd", "nt", T, 18432, 768, TL)
run("fwd", "nt", T, 30528, 768, TL)
run("dgrad", "nn", T, 3072, 768, TL)
run("wgrad", "tn", 18432, 768, T, TL)
run("wgrad", "tn", 30528, 768, T, TL)
run("wgrad", "tn", 768, 768, T, TL)
print("K scan (fixed cost per launch = intercept)")
for K in (64, 256, 768, 1536):
    run("kscan", "nt", T, 3072, K, ["256x192"])
for K in (64, 256, 768, 1536):
    run("kscan", "nt", T, 768, K, ["128x256"])
