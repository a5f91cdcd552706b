"""Loading of tests/golden/vq_*.npz (see tests/golden/make_vq_golden.py for how they were made)."""
import glob
import hashlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_inputs(seed, B, S, K, D, regime):
    """Same recipe as tests/golden/make_vq_golden.py::make_inputs (big cases store only the recipe)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    z = rng.standard_normal((B, S, D), dtype=np.float32)
    if regime == "default_init":
        E = rng.uniform(-1.0 / K, 1.0 / K, size=(K, D)).astype(np.float32)
    elif regime == "separated":
        E = rng.standard_normal((K, D), dtype=np.float32)
    elif regime == "ties":
        E = rng.standard_normal((K, D), dtype=np.float32)
        E[K // 2:] = E[: K - K // 2]
    elif regime == "onto_codes":
        E = rng.standard_normal((K, D), dtype=np.float32)
        pick = rng.integers(0, K, size=(B, S))
        z = (E[pick] + 1e-3 * rng.standard_normal((B, S, D), dtype=np.float32)).astype(np.float32)
    else:
        raise ValueError(regime)
    g = rng.standard_normal((B, S, D), dtype=np.float32)
    c = np.float32(rng.uniform(0.5, 2.0))
    return z, E, g, c


# Cases too large for the per-case parametrised tests (each of those runs the single-threaded oracle several times): they have
# tests of their own (tests/test_oracle_golden.py::test_c_oracle_large_codebook_near_ties, tests/test_vq_gpu.py::test_k8192_default_*).
HEAVY = ("k8192_default",)


def case_names():
    return sorted(n for n in (os.path.basename(p)[3:-4] for p in glob.glob(os.path.join(GOLDEN, "vq_*.npz"))) if n not in HEAVY)


def load_case(name):
    f = dict(np.load(os.path.join(GOLDEN, f"vq_{name}.npz"), allow_pickle=False))
    c = {k: (v.item() if v.ndim == 0 else v) for k, v in f.items()}
    if "z" not in c:
        z, E, g, cc = make_inputs(c["seed"], c["B"], c["S"], c["K"], c["D"], c["regime"])
        assert np.float32(cc) == np.float32(c["c"])
        c.update(z=z, E=E, g=g)
    # inputs must be byte-identical to what the reference was fed
    assert _sha(c["z"]) == c["sha_z"] and _sha(c["E"]) == c["sha_E"] and _sha(c["g"]) == c["sha_g"], \
        f"golden inputs of {name} do not reproduce (numpy RNG stream changed?)"
    c["full"] = "z_q" in c
    return c


def ulp32(x):
    return np.spacing(np.abs(np.float32(x)))


def check_indices(c, idx, max_ulps=8.0):
    """idx must equal the reference's argmin, except on near-tie tokens where it must still be a minimiser:
    fp64 distance of the chosen code within `max_ulps` f32-ulps(|d|) of the fp64 minimum (SURVEY.md §7 hard part 1).
    Returns the number of tokens that differ from the reference."""
    idx = np.asarray(idx).reshape(-1)
    ref = c["idx"].astype(np.int64)
    diff = np.nonzero(idx != ref)[0]
    if diff.size == 0:
        return 0
    D = c["D"]
    z = c["z"].reshape(-1, D).astype(np.float64)
    E = c["E"].astype(np.float64)
    for n in diff:
        d = (z[n] ** 2).sum() + (E ** 2).sum(1) - 2.0 * E @ z[n]
        tol = max_ulps * float(ulp32(np.abs(d).max()))
        assert d[idx[n]] - d.min() <= tol, (
            f"{c['name']}: token {n}: chose code {idx[n]} (d={d[idx[n]]:.9g}) but minimum is {d.min():.9g} "
            f"at {d.argmin()} (reference chose {ref[n]}); gap {d[idx[n]] - d.min():.3g} > tol {tol:.3g}")
    return int(diff.size)


# Tokens whose chosen code may differ from the reference's: NONE, on every committed case (north_star: "quantized indices match the
# reference CPU PyTorch run bit-exact").  That includes the near-tie regimes -- c2_default (N = 8192, K = 512: 402 tokens with a
# top-2 gap below 1e-3, where the reference itself differs from the fp64 arg-min on 9 tokens) and k8192_default (K = 8192: 5310
# such tokens, 93 below 1e-5; the reference differs from fp64 on 124): oracle and HIP kernel follow the reference's f32 expression
# order ("kvq order v1") and have measured 0 flips on every box since round 2, so a single flip is a regression.
# (check_indices still says, for a failing token, whether the chosen code was at least a minimiser within 8 ulp.)
MAX_NEAR_TIE_FLIPS = {}


def check_flip_budget(c, ndiff):
    budget = MAX_NEAR_TIE_FLIPS.get(c["name"], 0)
    print(f"[golden] {c['name']}: {ndiff} of {c['B'] * c['S']} indices differ from the reference (budget {budget})")
    assert ndiff <= budget, f"{c['name']}: {ndiff} index flips against the reference exceed the recorded budget of {budget}"
