"""Seeded ray sets for the parity tests (same inputs go to the oracle and to the HIP path)."""
import numpy as np


def primary_grid(orc, w, h, seed=7):
    """One jittered primary ray per pixel of a w x h image, default camera of that aspect."""
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:h, 0:w]
    u = ((xs + rng.random((h, w))) / w).astype(np.float32).ravel()
    v = ((ys + rng.random((h, w))) / h).astype(np.float32).ravel()
    return orc.generate_rays(orc.camera_default(w, h), u, v)


def interior_rays(n, seed=11, extent=1.9):
    """Random origins inside the Cornell box with random (unnormalised) directions: what shadow
    and bounce rays look like, including rays that start inside the spheres."""
    rng = np.random.default_rng(seed)
    org = rng.uniform(-extent, extent, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    return org, d


def outside_rays(n, seed=13):
    """Origins well outside the scene box aimed near it: exercises the root-box rejection."""
    rng = np.random.default_rng(seed)
    org = rng.uniform(-8, 8, (n, 3)).astype(np.float32)
    tgt = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    return org, (tgt - org).astype(np.float32)


def axis_rays(n, seed=17):
    """Directions with exact zeros / tiny components: the |d| < 1e-8 slab rule and the
    parallel-plane divisions of Square::intersect."""
    rng = np.random.default_rng(seed)
    org = rng.uniform(-2.5, 2.5, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    kill = rng.integers(0, 3, n)
    d[np.arange(n), kill] = rng.choice(np.array([0.0, 1e-9, -1e-9, 1e-7], np.float32), n)
    two = rng.random(n) < 0.3
    d[two, (kill[two] + 1) % 3] = 0.0
    return org, d


def sphere_surface_rays(center, radius, n, seed=19):
    """Rays leaving / grazing a sphere surface (offset 1e-3 like the reference's bounce rays)."""
    rng = np.random.default_rng(seed)
    nrm = rng.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    sign = rng.choice([-1.0, 1.0], (n, 1))
    org = (np.asarray(center) + nrm * radius + sign * 1e-3 * nrm).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    return org, d


def mixed(orc, n_each=4000):
    """(origins, unit directions) as the fields of reference Ray objects."""
    parts = [primary_grid(orc, 96, 54), interior_rays(n_each), outside_rays(n_each), axis_rays(n_each),
             sphere_surface_rays((-1.0, -1.25, -0.5), 0.75, n_each), sphere_surface_rays((1.0, -1.25, 0.5), 0.75, n_each)]
    return np.concatenate([p[0] for p in parts]), orc.normalize_dirs(np.concatenate([p[1] for p in parts]))
