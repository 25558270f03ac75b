"""Seeded synthetic scan pairs for the ICP hot path.

The reference ships no data (no tests/, no datasets: CMakeLists.txt:1-46), so
bench.py and the tests use this generator.  It is deterministic across numpy
versions: the only random source is a counter-mode splitmix64 written out here.

Scene ("street canyon"): ground plane z=0 over [-60,60]^2, two long walls
y=+-8 m and two end walls x=+-60 m (height 6 m), 20 axis-aligned boxes
(1-4 m) on the ground; points are sampled uniformly by area.

Pose convention = the reference's (include/mola-fe-lidar/LidarOdometry.h:122,131):
the sought pose is `to` (local, queries) w.r.t. `from` (global, map), i.e.
g ~= T (+) l.  `make_pair` returns (map, local, T_gt).
"""
from __future__ import annotations

import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix(z: np.ndarray) -> np.ndarray:
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def splitmix_uniform(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """n uniform doubles in [0,1): value k of stream `stream` of `seed`."""
    with np.errstate(over="ignore"):
        s0 = _mix(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) * np.uint64(0xD1342543DE82EF95)
                  + np.uint64(stream) * np.uint64(0xA0761D6478BD642F) + np.uint64(1))
        k = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = _mix(s0 + k * _GAMMA)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def splitmix_normal(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """n standard normals (Box-Muller on two uniform streams)."""
    u1 = splitmix_uniform(seed, 2 * stream, n, offset)
    u2 = splitmix_uniform(seed, 2 * stream + 1, n, offset)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def pose_from_xyzypr(x, y, z, yaw, pitch, roll) -> np.ndarray:
    """4x4 of an MRPT-style TPose3D: R = Rz(yaw) Ry(pitch) Rx(roll)
    (the reference builds its guess this way: src/LidarOdometry.cpp:272-275)."""
    cy, sy = np.cos(yaw), np.sin(yaw)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cr, sr = np.cos(roll), np.sin(roll)
    T = np.eye(4)
    T[:3, :3] = [[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                 [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                 [-sp, cp * sr, cp * cr]]
    T[:3, 3] = [x, y, z]
    return T


#: ground-truth pose of SURVEY §8(d): (0.50, 0.20, 0.05 m; yaw 2, pitch 0.5, roll 0.3 deg)
T_GT_DEFAULT = pose_from_xyzypr(0.50, 0.20, 0.05, np.deg2rad(2.0), np.deg2rad(0.5), np.deg2rad(0.3))


class Scene:
    """List of axis-aligned rectangles (origin, edge u, edge v) with areas."""

    def __init__(self, scene_seed: int = 7, half: float = 60.0, wall_y: float = 8.0, wall_h: float = 6.0,
                 n_boxes: int = 20):
        rects = []
        # ground
        rects.append(((-half, -half, 0.0), (2 * half, 0, 0), (0, 2 * half, 0)))
        # long walls y = +-wall_y
        for s in (-1.0, 1.0):
            rects.append(((-half, s * wall_y, 0.0), (2 * half, 0, 0), (0, 0, wall_h)))
        # end walls x = +-half (between the long walls)
        for s in (-1.0, 1.0):
            rects.append(((s * half, -wall_y, 0.0), (0, 2 * wall_y, 0), (0, 0, wall_h)))
        # boxes
        u = splitmix_uniform(scene_seed, 0, 5 * n_boxes).reshape(n_boxes, 5)
        self.boxes = []
        for b in range(n_boxes):
            cx = -0.9 * half + 1.8 * half * u[b, 0]
            cy = -0.8 * wall_y + 1.6 * wall_y * u[b, 1]
            sx, sy, sz = 1.0 + 3.0 * u[b, 2], 1.0 + 3.0 * u[b, 3], 1.0 + 3.0 * u[b, 4]
            x0, y0 = cx - sx / 2, cy - sy / 2
            self.boxes.append((x0, y0, 0.0, x0 + sx, y0 + sy, sz))
            rects.append(((x0, y0, sz), (sx, 0, 0), (0, sy, 0)))          # top
            rects.append(((x0, y0, 0.0), (sx, 0, 0), (0, 0, sz)))         # y = y0
            rects.append(((x0, y0 + sy, 0.0), (sx, 0, 0), (0, 0, sz)))    # y = y1
            rects.append(((x0, y0, 0.0), (0, sy, 0), (0, 0, sz)))         # x = x0
            rects.append(((x0 + sx, y0, 0.0), (0, sy, 0), (0, 0, sz)))    # x = x1
        self.origin = np.array([r[0] for r in rects], dtype=np.float64)
        self.eu = np.array([r[1] for r in rects], dtype=np.float64)
        self.ev = np.array([r[2] for r in rects], dtype=np.float64)
        area = np.linalg.norm(np.cross(self.eu, self.ev), axis=1)
        self.cdf = np.cumsum(area) / area.sum()
        self.half, self.wall_y, self.wall_h = half, wall_y, wall_h

    def sample(self, n: int, seed: int, chunk: int = 1 << 20) -> np.ndarray:
        """n points (float64, shape (n,3)) sampled uniformly by area."""
        out = np.empty((n, 3), dtype=np.float64)
        for o in range(0, n, chunk):
            m = min(chunk, n - o)
            s = splitmix_uniform(seed, 0, m, o)
            a = splitmix_uniform(seed, 1, m, o)
            b = splitmix_uniform(seed, 2, m, o)
            k = np.minimum(np.searchsorted(self.cdf, s, side="right"), len(self.cdf) - 1)
            out[o:o + m] = self.origin[k] + a[:, None] * self.eu[k] + b[:, None] * self.ev[k]
        return out


_SCENE_CACHE: dict = {}


def default_scene() -> Scene:
    if "s" not in _SCENE_CACHE:
        _SCENE_CACHE["s"] = Scene()
    return _SCENE_CACHE["s"]


def make_pair(n_local: int, n_map: int, seed: int = 42, T_gt: np.ndarray | None = None,
              noise_sigma: float = 0.01, scene: Scene | None = None):
    """Returns (map_xyz float32 (3,M) SoA, local_xyz float32 (3,N) SoA, T_gt 4x4 float64).

    map   = n_map samples (seed);
    local = n_local independent samples (seed+1) + N(0, sigma^2) noise, moved by T_gt^-1,
    so that  map ~= T_gt (+) local  (init guess = identity is ~0.54 m / 2.1 deg away)."""
    scene = scene or default_scene()
    T_gt = T_GT_DEFAULT if T_gt is None else np.asarray(T_gt, dtype=np.float64)
    g = scene.sample(n_map, seed)
    p = scene.sample(n_local, seed + 1)
    if noise_sigma > 0:
        for c in range(3):
            for o in range(0, n_local, 1 << 20):
                m = min(1 << 20, n_local - o)
                p[o:o + m, c] += noise_sigma * splitmix_normal(seed + 1, 10 + c, m, o)
    Ti = np.linalg.inv(T_gt)
    l = p @ Ti[:3, :3].T + Ti[:3, 3]
    return (np.ascontiguousarray(g.T.astype(np.float32)), np.ascontiguousarray(l.T.astype(np.float32)), T_gt)


def lidar_scan(pose: np.ndarray, n_rings: int = 64, n_az: int = 1875, max_range: float = 80.0,
               sensor_h: float = 1.73, noise_sigma: float = 0.01, seed: int = 1,
               scene: Scene | None = None) -> np.ndarray:
    """KITTI-like spinning lidar model (HDL-64E geometry: 64 rings, +2..-24.8 deg) ray-cast
    against the scene.  `pose` = 4x4 vehicle pose in the world.  Returns the hits in the
    SENSOR frame, float32 SoA (3,K) (K ~ 100-120k), standing in for config 1's
    'one KITTI-00 scan pair' (no KITTI data exists in the image)."""
    scene = scene or default_scene()
    el = np.deg2rad(np.linspace(2.0, -24.8, n_rings))
    az = np.linspace(-np.pi, np.pi, n_az, endpoint=False)
    E, A = np.meshgrid(el, az, indexing="ij")
    d = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], axis=-1).reshape(-1, 3)
    Ts = np.array(pose, dtype=np.float64) @ pose_from_xyzypr(0, 0, sensor_h, 0, 0, 0)
    o = Ts[:3, 3]
    dw = d @ Ts[:3, :3].T
    best = np.full(len(dw), np.inf)
    # rectangles: origin + a*eu + b*ev, a,b in [0,1]
    for O, U, V in zip(scene.origin, scene.eu, scene.ev):
        nrm = np.cross(U, V)
        den = dw @ nrm
        with np.errstate(divide="ignore", invalid="ignore"):
            t = ((O - o) @ nrm) / den
        with np.errstate(invalid="ignore"):
            hit = o + t[:, None] * dw - O
        a = (hit @ U) / (U @ U)
        b = (hit @ V) / (V @ V)
        ok = (np.abs(den) > 1e-12) & (t > 0.5) & (t < max_range) & (a >= 0) & (a <= 1) & (b >= 0) & (b <= 1)
        best = np.where(ok & (t < best), t, best)
    keep = np.isfinite(best)
    r = best[keep]
    if noise_sigma > 0:
        r = r + noise_sigma * splitmix_normal(seed, 0, len(r))
    pts = d[keep] * r[:, None]
    return np.ascontiguousarray(pts.T.astype(np.float32))
