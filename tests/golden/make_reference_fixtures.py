"""Writes tests/golden/reference_kat.npz: known answers produced by the REFERENCE's own optimiser code
(lmcma_path_planner/src/lmcma.cpp, compiled where it lies into oracle/_ref/ by `make -C oracle ref`).

Run in the build container only (the reference tree does not travel):
    python tests/golden/make_reference_fixtures.py

The fixture holds data only: inputs (seeds, shapes, bounds, the fitness values that were fed back) and
the reference's outputs (random streams, constants, per-generation state, CRC32 digests of every
population).  tests/test_reference_pins.py replays the same inputs through oracle/lmcma_oracle.c and
demands the same bits.  glibc's qsort is a merge sort for these sizes (stable); the reference's tie
order is therefore candidate order, which the sort vectors below pin as well.
"""
import json
import os
import platform
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import pyref  # noqa: E402

OUT = os.path.join(HERE, "reference_kat.npz")

SEEDS = (1, 7, 42, 2147483646)
SHAPES = ((2, 6), (8, 6), (64, 16), (128, 64))


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a, dtype=np.float64).tobytes())


# objective functions: evaluated here once; the values fed back are stored, so the replay does not
# depend on how numpy sums on another machine
def sphere(X):
    out = []
    for x in X:  # accumulated left to right, like a C loop
        acc = 0.0
        for v in x:
            acc += float(v) * float(v)
        out.append(acc)
    return np.array(out)


def ellipsoid(X):
    n = X.shape[1]
    w = 10.0 ** (3.0 * np.arange(n) / max(n - 1, 1))
    return np.array([float(np.dot(w * x, x)) for x in X])


def toy(X):
    # the reference example's objective (example_lmcma.cpp:18-26): two Gaussian wells
    out = []
    for x, y in X:
        g1 = -4.0 * np.exp(-((x - 4.0) ** 2 + (y - 4.0) ** 2) / 2.0)
        g2 = -2.0 * np.exp(-((x - 2.0) ** 2 + (y - 2.0) ** 2) / 2.0)
        out.append(float(g1 + g2))
    return np.array(out)


OBJECTIVES = {"sphere": sphere, "ellipsoid": ellipsoid, "toy": toy}

# name, N, lambda (0 = the reference's default), objective, generations, sigma, seed, initial, lo, hi, arx generations kept
TRACES = [
    ("toy_2_6_bounds", 2, 0, "toy", 166, 1.0, 3, 0.0, -2.0, 15.0, (0, 1, 2, 80, 165)),
    ("sphere_2_6", 2, 6, "sphere", 60, 1.0, 1, 1.5, None, None, (0, 1, 2, 30, 59)),
    ("sphere_8_6_probe", 8, 6, "sphere", 50, 0.5, 42, 1.0, None, None, (0, 1, 2, 25, 49)),
    ("ellipsoid_8_6_bounds", 8, 6, "ellipsoid", 80, 1.0, 7, 3.0, -1.0, 5.0, (0, 1, 2, 40, 79)),
    ("sphere_8_6_uniform_start", 8, 6, "sphere", 30, 0.3, 2147483646, None, None, None, (0, 1, 29)),
    ("ellipsoid_64_16", 64, 16, "ellipsoid", 60, 1.0, 42, 1.0, None, None, (0, 1, 17, 59)),
    ("ellipsoid_64_16_bounds", 64, 16, "ellipsoid", 60, 2.0, 1, 1.0, -0.5, 4.0, (0, 1, 17, 59)),
    ("sphere_128_64", 128, 64, "sphere", 90, 1.0, 7, 1.0, None, None, (0, 66, 89)),
    ("ellipsoid_128_64_bounds", 128, 64, "ellipsoid", 90, 1.0, 42, 2.0, -3.0, 3.0, (0, 66, 89)),
]


def trace_lmcma(out, name, N, lam, objective, gens, sigma, seed, initial, lo, hi, keep):
    init = None if initial is None else np.full(N, float(initial))
    lo_a = None if lo is None else np.full(N, float(lo))
    hi_a = None if hi is None else np.full(N, float(hi))
    r = pyref.RefLMCMA(N, lam, initial=init, lo=lo_a, hi=hi_a, sigma=sigma, seed=seed)
    lam = r.lam
    f = OBJECTIVES[objective]
    fed = np.empty((gens, lam))
    xmean = np.empty((gens + 1, N))
    sig = np.empty(gens + 1)
    s = np.empty(gens + 1)
    bestf = np.empty(gens)
    digests = np.empty(gens + 1, dtype=np.uint32)
    iterators = np.full((gens, r.m), -1, dtype=np.int32)
    iterator_sz = np.empty(gens, dtype=np.int32)
    xmean[0], sig[0], s[0] = r.xmean, r.sigma, r.s
    digests[0] = crc(r.arx)
    kept = {}
    for g in range(gens):
        X = r.arx
        if g in keep:
            kept[g] = X
        fed[g] = f(X)
        r.tell_all(fed[g])
        xmean[g + 1], sig[g + 1], s[g + 1] = r.xmean, r.sigma, r.s
        bestf[g] = r.BestF
        digests[g + 1] = crc(r.arx)
        iterator_sz[g] = r.iterator_sz
        iterators[g, : r.iterator_sz] = r.iterator[: r.iterator_sz]
    p = "trace/%s/" % name
    out[p + "fed"] = fed
    out[p + "xmean"] = xmean
    out[p + "sigma"] = sig
    out[p + "s"] = s
    out[p + "bestf"] = bestf
    out[p + "arx_crc32"] = digests
    out[p + "iterator"] = iterators
    out[p + "iterator_sz"] = iterator_sz
    for g, X in kept.items():
        out[p + "arx_g%d" % g] = X
    k = r.iterator_sz
    it = r.iterator[:k]
    out[p + "final_v"] = r.v_arr[it]
    out[p + "final_pc_arr"] = r.pc_arr[it]
    out[p + "final_nj"] = r.nj[it]
    out[p + "final_lj"] = r.lj[it]
    out[p + "final_pc"] = r.pc
    assert r.counteval == gens * lam and r.itr == gens
    return dict(name=name, N=N, lam=lam, objective=objective, generations=gens, sigma=sigma, seed=seed,
                initial=initial, lo=lo, hi=hi, arx_generations=sorted(kept))


def trace_sibling(out, cls, name, N, lam, objective, gens, sigma, seed, initial, lo, hi):
    init = np.full(N, float(initial))
    lo_a = None if lo is None else np.full(N, float(lo))
    hi_a = None if hi is None else np.full(N, float(hi))
    r = cls(N, lam, initial=init, lo=lo_a, hi=hi_a, sigma=sigma, seed=seed)
    f = OBJECTIVES[objective]
    fed = np.empty((gens, r.lam))
    xmean = np.empty((gens + 1, N))
    sig = np.empty(gens + 1)
    digests = np.empty(gens + 1, dtype=np.uint32)
    xmean[0], sig[0], digests[0] = r.xmean, r.sigma, crc(r.arx)
    for g in range(gens):
        fed[g] = f(r.arx)
        r.tell_all(fed[g])
        xmean[g + 1], sig[g + 1], digests[g + 1] = r.xmean, r.sigma, crc(r.arx)
    p = "sibling/%s/" % name
    out[p + "fed"] = fed
    out[p + "xmean"] = xmean
    out[p + "sigma"] = sig
    out[p + "arx_crc32"] = digests
    out[p + "final_arx"] = r.arx
    out[p + "final_shape"] = r.diag if cls is pyref.RefSepCMA else r.factor
    return dict(name=name, kind=cls.kind, N=N, lam=r.lam, objective=objective, generations=gens, sigma=sigma, seed=seed,
                initial=initial, lo=lo, hi=hi)


def main():
    if not pyref.available():
        pyref.build()
    out = {}
    meta = {"generator": "tests/golden/make_reference_fixtures.py",
            "reference_file": "lmcma_path_planner/src/lmcma.cpp (lines 1-5, 7-835 compiled; see oracle/Makefile)",
            "compiler_flags": "g++ -std=c++11 -O2 -ffp-contract=off -fno-fast-math",
            "libc": " ".join(platform.libc_ver()), "machine": platform.machine(),
            "note": "qsort of this libc is a stable merge sort at these sizes; covariance/Eigen paths are not part of the build"}

    # random streams (lmcma.cpp:14-82)
    for seed in SEEDS:
        out["rng/uniform/%d" % seed] = pyref.uniform_stream(seed, 96)
        out["rng/gauss/%d" % seed] = pyref.gauss_stream(seed, 97)  # odd: ends on a cached value's partner

    # constants (lmcma.cpp:134-156, 266-272)
    consts = []
    for N, lam in SHAPES + ((2, 0), (10, 0), (128, 0)):
        r = pyref.RefLMCMA(N, lam, initial=np.zeros(N), seed=1)
        out["const/%d_%d/weights" % (N, lam)] = r.weights
        out["const/%d_%d/scalars" % (N, lam)] = np.array([r.mueff, r.c1, r.cc, r.K, r.M])
        consts.append(dict(N=N, lam_arg=lam, lam=r.lam, mu=r.mu, nvectors=r.m))
    meta["constants"] = consts

    # sort (lmcma.cpp:84-104): ties, signed zeros, infinities (NaN is undefined behaviour in `compare`)
    rng = np.random.default_rng(12345)
    sorts = {
        "ties": np.array([3.0, 1.0, 2.0, 1.0, 5.0, 1.0, -0.0, 0.0, 2.0, np.inf, -np.inf, np.inf]),
        "random_64": rng.normal(size=64),
        "quantised_257": np.round(rng.normal(size=257) * 4.0) / 4.0,
        "quantised_4096": np.round(rng.normal(size=4096) * 16.0) / 16.0,
    }
    for k, v in sorts.items():
        sv, idx = pyref.sort(v)
        out["sort/%s/in" % k] = v
        out["sort/%s/sorted" % k] = sv
        out["sort/%s/index" % k] = idx
    meta["sorts"] = sorted(sorts)

    # differentiationMatrix (lmcma.cpp:812-834)
    dms = []
    for n, order, dt in ((8, 2, 1.0), (32, 2, 1.0), (1, 2, 1.0), (2, 2, 1.0), (8, 1, 1.0), (8, 3, 1.0), (8, 0, 1.0), (8, 2, 0.5)):
        out["diffmat/%d_%d_%g" % (n, order, dt)] = pyref.differentiation_matrix(n, order, dt)
        dms.append([n, order, dt])
    meta["diffmats"] = dms

    # ask/tell traces of LMCMA (lmcma.cpp:172-205, 262-463)
    meta["traces"] = [trace_lmcma(out, *t) for t in TRACES]

    # three feedback terms per candidate, summed left to right (lmcma.cpp:186-188)
    r = pyref.RefLMCMA(4, 6, initial=np.full(4, 1.0), sigma=0.7, seed=11)
    terms = np.empty((20, 6, 3))
    xm = np.empty((20, 4))
    for g in range(20):
        for i in range(6):
            x = r.ask()
            t = np.array([float(np.dot(x, x)), 1e-17 * (i + 1), float(np.abs(x).sum())])
            terms[g, i] = t
            r.tell(t)
        xm[g] = r.xmean
    out["terms/fed"] = terms
    out["terms/xmean"] = xm
    out["terms/bestf"] = np.array([r.BestF])

    # the stop rule (lmcma.cpp:426-429): a sphere until sigma < 1e-20
    r = pyref.RefLMCMA(2, 6, initial=np.full(2, 0.5), sigma=0.1, seed=5)
    fed = []
    g_done = -1
    for g in range(4000):
        f = sphere(r.arx)
        fed.append(f)
        r.tell_all(f)
        if r.done():
            g_done = g
            break
    assert g_done > 0
    out["stop/fed"] = np.array(fed)
    out["stop/sigma_final"] = np.array([r.sigma])
    meta["stop"] = dict(N=2, lam=6, initial=0.5, sigma=0.1, seed=5, generation_done=g_done)

    # the sibling optimisers (lmcma.cpp:466-577, 579-754)
    meta["siblings"] = [
        trace_sibling(out, pyref.RefSepCMA, "sep_10_12", 10, 12, "ellipsoid", 60, 1.0, 5, 3.0, None, None),
        trace_sibling(out, pyref.RefSepCMA, "sep_8_default_bounds", 8, 0, "sphere", 40, 0.5, 9, 1.0, -0.5, 2.0),
        trace_sibling(out, pyref.RefCMAChol, "chol_8_16", 8, 16, "ellipsoid", 60, 1.0, 3, 2.0, None, None),
        trace_sibling(out, pyref.RefCMAChol, "chol_6_default_bounds", 6, 0, "sphere", 40, 0.5, 4, 1.0, -0.5, 2.0),
    ]

    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
