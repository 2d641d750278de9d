"""Seeded synthetic sample sets for the BRDF fitter.

The reference ships no benchmark inputs for the fitting path (its only data are 16-sample-per-surfel
captures, brdfdata.h:58), so the configurations of BASELINE.json are synthesised:

* cosine planes  c = 0.05 + 0.95 u          (three planes, the layout of struct extraData,
                                              brdfdata.cpp:962-966: [0,n) cos(L.N), [n,2n) cos(N.H),
                                              [2n,3n) cos(R.V) / cos(N.V))
* measurement    x = f(truth; planes) + 0.01 (u - 0.5)
* start          p0 = {0.5, 1, 1}  (brdfdata.cpp:1085); Ward {0.5, 0.5, 0.3}
* options        opts = {1e-3, 1e-15, 1e-15, 1e-20, 1e-6}, itmax = 100, bounds [0,100]^3
                 (brdfdata.cpp:1109-1117)

u is a counter-based stream -- splitmix64 over (seed + gamma*(index+1)) with the survey's seed
88172645463325252 -- so every surfel/plane/sample can be generated independently: ranks generate only
their own shard, and the device generator in csrc/ produces the same bits.  (The survey sketched a
sequential xorshift64; a counter-based stream is what sharding across GPUs needs.)
"""
from __future__ import annotations

import numpy as np

SEED = 88172645463325252
_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
PI = 3.1415926535897932384626433832795

P0 = {0: (0.5, 1.0, 1.0), 1: (0.5, 1.0, 1.0), 2: (0.5, 0.5, 0.3)}
TRUTH = {0: (0.35, 0.6, 24.0), 1: (0.35, 0.6, 24.0), 2: (0.35, 0.25, 0.15)}
OPTS = (1e-3, 1e-15, 1e-15, 1e-20, 1e-6)
ITMAX = 100
LB = (0.0, 0.0, 0.0)  # the application's box, brdfdata.cpp:1112-1113
UB = (100.0, 100.0, 100.0)


def bounds(model: int):
    """Box for the multi-surfel configurations.  Phong/Blinn-Phong: the application's [0,100]^3.  Ward divides
    by alpha^2, so a projected step that lands exactly on alpha = 0 turns the model into NaN and levmar (the
    reference's as much as this build's) stops with reason 7 / LM_ERROR; its roughness is kept >= 0.01."""
    return ((0.0, 0.0, 0.01), UB) if model == 2 else (LB, UB)


def uniform(seed: int, index: np.ndarray) -> np.ndarray:
    """u in [0,1) for 64-bit counters `index` (any shape)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + _GAMMA * (index.astype(np.uint64) + np.uint64(1))
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def model_value(model: int, p, c0, c1, c2):
    """numpy evaluation of the three models (only used to synthesise measurements)."""
    p0, p1, p2 = (np.asarray(v, dtype=np.float64) for v in p)
    if model == 0:
        return p0 * c0 + ((p2 + 2.0) / 2.0 * PI) * p1 * np.power(c2, p2)
    if model == 1:
        return p0 * c0 + p1 * np.power(c1, p2)
    if model == 2:
        a2 = p2 * p2
        t2 = (1.0 - c1 * c1) / (c1 * c1)
        spec = ((1.0 / (4.0 * PI * a2)) * np.exp(-(t2 * (1.0 / a2)))) * (1.0 / np.sqrt(c0 * c2))
        return c0 * (p0 / PI + p1 * spec)
    raise ValueError(f"unknown model {model}")


def surfel_truth(model: int, first: int, count: int, seed: int = SEED) -> np.ndarray:
    """Per-surfel ground truth for the multi-surfel configurations: kd,ks in U[0.1,0.9], n in U[2,62]
    (Ward: rho_d in U[0.1,0.9], rho_s in U[0.05,0.45], alpha in U[0.08,0.38])."""
    s = np.arange(first, first + count, dtype=np.uint64)
    u = [uniform(seed ^ 0x5DEECE66D, s * np.uint64(3) + np.uint64(j)) for j in range(3)]
    if model == 2:
        return np.stack([0.1 + 0.8 * u[0], 0.05 + 0.4 * u[1], 0.08 + 0.3 * u[2]], axis=1)
    return np.stack([0.1 + 0.8 * u[0], 0.1 + 0.8 * u[1], 2.0 + 60.0 * u[2]], axis=1)


def make_surfels(model: int, n: int, first: int = 0, count: int = 1, seed: int = SEED,
                 truth: np.ndarray | None = None):
    """Samples for surfels [first, first+count).

    Returns (angles[count,3,n], x[count,n], truth[count,3]), float64, C-contiguous.  The counter of
    surfel s, stream k (0..2 planes, 3 noise), sample i is (s*4 + k)*n + i.
    """
    s = np.arange(first, first + count, dtype=np.uint64)[:, None, None]
    k = np.arange(4, dtype=np.uint64)[None, :, None]
    i = np.arange(n, dtype=np.uint64)[None, None, :]
    u = uniform(seed, (s * np.uint64(4) + k) * np.uint64(n) + i)
    angles = np.ascontiguousarray(0.05 + 0.95 * u[:, :3, :])
    if truth is None:
        truth = surfel_truth(model, first, count, seed)
    truth = np.ascontiguousarray(np.broadcast_to(np.asarray(truth, dtype=np.float64), (count, 3)))
    f = model_value(model, (truth[:, 0:1], truth[:, 1:2], truth[:, 2:3]), angles[:, 0], angles[:, 1], angles[:, 2])
    x = np.ascontiguousarray(f + 0.01 * (u[:, 3, :] - 0.5))
    return angles, x, truth


def make_single(model: int, n: int, seed: int = SEED):
    """The single-material configurations (BASELINE.json configs 1-3): fixed truth TRUTH[model]."""
    a, x, t = make_surfels(model, n, 0, 1, seed, truth=np.asarray(TRUTH[model])[None, :])
    return a[0], x[0], t[0]
