"""Deterministic synthetic scenes of the benchmark configurations (definitions, not checkers).

SURVEY.md 8(d) C5: the reference's "Random Spheres" preset has 7 spheres; the 10 000-sphere stress scene is
defined by the build: splitmix64 seeded 0x5EED5EED, centres uniform in [-1.5, 1.5]^3 then rounded to binary32,
radii uniform in [0.01, 0.04] kept as doubles.  The oracle package has its own copy of these generators;
tests/test_host_logic.py checks that the two agree."""
import numpy as np


def synthetic_spheres(n=10000, seed=0x5EED5EED):
    """float64 [n, 4]: x, y, z (binary32-valued), r."""
    mask = (1 << 64) - 1
    state = seed & mask

    def uniform():
        nonlocal state
        state = (state + 0x9E3779B97F4A7C15) & mask
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
        z ^= z >> 31
        return (z >> 11) * (1.0 / 9007199254740992.0)

    out = np.zeros((n, 4), np.float64)
    for i in range(n):
        for k in range(3):
            out[i, k] = np.float32(-1.5 + 3.0 * uniform())
        out[i, 3] = 0.01 + 0.03 * uniform()
    return out


def synthetic_mixed_prims(n=40, seed=7):
    """Spheres, boxes and tori in turn, every second one with a rotation argument: dicts with type, pos, rot and
    r | half | radius (placed like SceneManager.createSphere / createBox / createTorus)."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        kind = ("sphere", "box", "torus")[i % 3]
        d = {"type": kind, "pos": [float(np.float32(v)) for v in rng.uniform(-1.3, 1.3, 3)],
             "rot": [float(np.float32(v)) for v in rng.uniform(-3.2, 3.2, 3)] if i % 2 else None}
        if kind == "sphere":
            d["r"] = float(rng.uniform(0.08, 0.3))
        elif kind == "box":
            d["half"] = [float(np.float32(v)) for v in rng.uniform(0.05, 0.3, 3)]
        else:
            d["radius"] = float(rng.uniform(0.1, 0.3))
        out.append(d)
    return out


def mixed_prims_as_triples(prims, make_transform):
    """(type, world_to_local, params) triples for Scene.loadPrims from the dicts above."""
    triples = []
    for d in prims:
        m = make_transform(*d["pos"], rotation=d["rot"])  # SceneManager.getTransform
        if d["type"] == "sphere":
            triples.append((0, m, [d["r"]]))
        elif d["type"] == "box":
            triples.append((1, m, d["half"]))
        else:
            triples.append((2, m, [d["radius"], d["radius"] / 4]))  # createTorus: minor = radius / 4
    return triples
