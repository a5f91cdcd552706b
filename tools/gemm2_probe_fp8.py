import os, sys
sys.argv = ["x", "none"]
exec(open("tools/gemm2_probe.py").read().split("TL = [")[0])
for n, k in [(768, 768), (2304, 768), (3072, 768), (768, 3072), (18432, 768), (30528, 768)]:
    x, w = rnd(T, k), rnd(n, k)
    b = rnd(n)
    x8, sx = nnops.fp8_quantize(x); w8, sw = nnops.fp8_quantize(w)
    out = torch.empty((T, n), device=dev, dtype=torch.bfloat16)
    fns = {"lib bf16": lambda: torch.addmm(b, x, w.t()), "fp8 gemm": lambda: nnops.gemm_fp8_nt(x8, w8, sx, sw, bias=b, out=out),
           "quant x": lambda: nnops.fp8_quantize(x, out=x8)}
    res = {kk: [] for kk in fns}
    for _ in range(3):
        for kk, f in fns.items():
            res[kk].append(bench(f))
    fl = 2.0 * T * n * k
    print(f"fwd N={n:6d} K={k:5d}: " + " | ".join(f"{kk} {sorted(v)[1]:7.1f} us" + (f" {fl/sorted(v)[1]/1e6:6.0f} TF" if 'quant' not in kk else "") for kk, v in res.items()), flush=True)
