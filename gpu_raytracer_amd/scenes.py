"""Seeded synthetic scenes for the parity tests and the bench.

No glTF assets exist offline, so the BASELINE configs run on procedural
stand-ins with matching triangle counts:

  default_scene()   SceneBuilder::build_default_scene (shared/src/lib.rs:1242-1286): 6 spheres, 2 tris, 1 light
  cornell12()       Cornell box, 12 triangles, 1 point light (C0 / C1)
  sponza_like()     262,144 triangles, seed 0x53504F4E: atrium shell, two colonnade storeys, arches, cloth, plants
  bistro_like()     3,800,000 triangles, seed 0x42495354: street block, facades, trees with tiny-leaf foliage, props
  random_soup()     small random triangle soups for property tests

All randomness is a counter-based integer hash (splitmix64) mapped to f32 by
exact scaling, so a scene is a pure function of (seed, parameters).
"""
from dataclasses import dataclass, field

import numpy as np

from . import hostpack as H
from . import types as T

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _hash_u32(seed, stream, idx):
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = idx + np.uint64(stream) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(32)).astype(np.uint32)


def _u01(seed, stream, n_or_idx):
    idx = np.arange(n_or_idx, dtype=np.uint64) if np.isscalar(n_or_idx) else n_or_idx
    return (_hash_u32(seed, stream, idx) >> np.uint32(8)).astype(np.float32) / np.float32(16777216.0)


@dataclass
class Scene:
    name: str
    spheres: np.ndarray
    lights: np.ndarray
    vertices: np.ndarray
    triangles: np.ndarray
    materials: np.ndarray
    camera: np.ndarray
    meta: dict = field(default_factory=dict)

    @property
    def n_triangles(self):
        return len(self.triangles)


class _Mesh:
    """Accumulates indexed triangle meshes."""

    def __init__(self):
        self.v = []
        self.t = []
        self.m = []
        self.nv = 0

    def add(self, verts, tris, mat):
        verts = np.asarray(verts, dtype=np.float32).reshape(-1, 3)
        tris = np.asarray(tris, dtype=np.int64).reshape(-1, 3)
        self.v.append(verts)
        self.t.append(tris + self.nv)
        mat = np.asarray(mat, dtype=np.uint32)
        self.m.append(np.broadcast_to(mat, (len(tris),)).copy())
        self.nv += len(verts)
        return len(tris)

    @property
    def n_tris(self):
        return sum(len(t) for t in self.t)

    def finish(self):
        v = np.concatenate(self.v) if self.v else np.zeros((0, 3), np.float32)
        t = np.concatenate(self.t) if self.t else np.zeros((0, 3), np.int64)
        m = np.concatenate(self.m) if self.m else np.zeros((0,), np.uint32)
        va = np.zeros(len(v), dtype=T.VERTEX)
        va["position"] = v
        ta = np.zeros(len(t), dtype=T.TRIANGLE)
        ta["v0_index"], ta["v1_index"], ta["v2_index"] = t[:, 0], t[:, 1], t[:, 2]
        ta["material_id"] = m
        return va, ta


def _grid(origin, du, dv, nu, nv, flip=False, disp=None):
    """(nu x nv) quad grid spanned by du, dv from origin -> (verts, tris); 2*nu*nv triangles."""
    origin, du, dv = (np.asarray(a, dtype=np.float64) for a in (origin, du, dv))
    iu, iv = np.meshgrid(np.arange(nu + 1), np.arange(nv + 1), indexing="ij")
    p = origin + iu[..., None] * (du / nu) + iv[..., None] * (dv / nv)
    if disp is not None:
        p = p + disp(iu / nu, iv / nv)
    verts = p.reshape(-1, 3)
    i00 = (iu[:-1, :-1] * (nv + 1) + iv[:-1, :-1]).ravel()
    i10, i01, i11 = i00 + (nv + 1), i00 + 1, i00 + (nv + 1) + 1
    if flip:
        tris = np.concatenate([np.stack([i00, i01, i10], 1), np.stack([i10, i01, i11], 1)])
    else:
        tris = np.concatenate([np.stack([i00, i10, i01], 1), np.stack([i10, i11, i01], 1)])
    return verts, tris


def _tube(center_fn, radius_fn, n_seg, n_ring, closed_ring=True):
    """Surface of revolution around a curve: center_fn(s)->(n,3) frame origin, radius_fn(s, theta)->r.
    Axis is +Y unless center_fn returns (origin, ex, ez) frames.  2*n_seg*n_ring triangles."""
    s = np.linspace(0.0, 1.0, n_ring + 1)
    th = np.linspace(0.0, 2.0 * np.pi, n_seg + 1)[:-1]
    S, TH = np.meshgrid(s, th, indexing="ij")
    o, ex, ez = center_fn(S)
    r = radius_fn(S, TH)
    p = o + ex * (r * np.cos(TH))[..., None] + ez * (r * np.sin(TH))[..., None]
    verts = p.reshape(-1, 3)
    ir, it = np.meshgrid(np.arange(n_ring), np.arange(n_seg), indexing="ij")
    a = (ir * n_seg + it).ravel()
    b = (ir * n_seg + (it + 1) % n_seg).ravel()
    c, d = a + n_seg, b + n_seg
    tris = np.concatenate([np.stack([a, c, b], 1), np.stack([b, c, d], 1)])
    return verts, tris


def _box(lo, hi, n=(1, 1, 1)):
    """Axis-aligned box, outward-facing; faces tessellated n[axis] per side."""
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    d = hi - lo
    nx, ny, nz = n
    out_v, out_t, nvtx = [], [], 0
    faces = [
        (lo, (d[0], 0, 0), (0, 0, d[2]), nx, nz, False),                      # bottom (y = lo)  normal -y
        ((lo[0], hi[1], lo[2]), (d[0], 0, 0), (0, 0, d[2]), nx, nz, True),    # top              normal +y
        (lo, (d[0], 0, 0), (0, d[1], 0), nx, ny, True),                       # z = lo           normal -z
        ((lo[0], lo[1], hi[2]), (d[0], 0, 0), (0, d[1], 0), nx, ny, False),   # z = hi           normal +z
        (lo, (0, 0, d[2]), (0, d[1], 0), nz, ny, False),                      # x = lo           normal -x
        ((hi[0], lo[1], lo[2]), (0, 0, d[2]), (0, d[1], 0), nz, ny, True),    # x = hi           normal +x
    ]
    for o, du, dv, a, b, flip in faces:
        v, t = _grid(o, du, dv, a, b, flip)
        out_v.append(v)
        out_t.append(t + nvtx)
        nvtx += len(v)
    return np.concatenate(out_v), np.concatenate(out_t)


# --------------------------------------------------------------------------------------
def default_scene():
    """SceneBuilder::build_default_scene — shared/src/lib.rs:1242-1286; Camera::new :231-238."""
    materials = np.array([
        H.material_diffuse((0.8, 0.3, 0.3)),
        H.material_metallic((0.8, 0.8, 0.2), 0.1),
        H.material_glass((0.2, 0.3, 0.8), 1.5, 0.9),
        H.material_emissive((1.0, 1.0, 1.0), (0.5, 0.5, 1.0)),
    ], dtype=T.MATERIAL)
    spheres = np.array([
        ((0.0, 0.0, -1.0), 0.5, 0), ((-1.0, 0.0, -1.0), 0.5, 1), ((1.0, 0.0, -1.0), 0.5, 2),
        ((2.0, 0.0, -3.0), 0.5, 2), ((-2.0, 0.0, -4.0), 0.5, 1), ((-1.0, 2.0, -5.0), 0.5, 3),
    ], dtype=T.SPHERE)
    legacy = [
        ((0.0, 1.0, -2.0), (-0.5, 0.0, -2.0), (0.5, 0.0, -2.0), 0),
        ((1.5, 0.5, -3.0), (1.0, -0.5, -3.0), (2.0, -0.5, -3.0), 1),
    ]
    vertices, triangles = H.legacy_to_indexed(legacy)
    lights = np.array([H.light_point((5.0, 7.0, 4.0), (1.0, 1.0, 1.0), 1.0, np.inf)], dtype=T.LIGHT)
    return Scene("default", spheres, lights, vertices, triangles, materials, H.camera())


def empty_scene():
    return Scene("empty", np.zeros(0, T.SPHERE), np.zeros(0, T.LIGHT), np.zeros(0, T.VERTEX),
                 np.zeros(0, T.TRIANGLE), np.array([H.material_diffuse((0.5, 0.5, 0.5))], dtype=T.MATERIAL),
                 H.camera())


def single_triangle():
    vertices, triangles = H.legacy_to_indexed([((0.0, 1.0, -2.0), (-1.0, -1.0, -2.0), (1.0, -1.0, -2.0), 0)])
    lights = np.array([H.light_point((0.0, 0.0, 3.0), (1.0, 1.0, 1.0), 2.0)], dtype=T.LIGHT)
    materials = np.array([H.material_diffuse((0.2, 0.7, 0.4))], dtype=T.MATERIAL)
    return Scene("single_triangle", np.zeros(0, T.SPHERE), lights, vertices, triangles, materials, H.camera())


def cornell12():
    """Cornell box of SURVEY.md §8d C0: cube [-1,1]^3 open toward +Z, 10 wall triangles + 2-triangle
    emissive ceiling quad at y = 0.999, one point light at (0, 0.9, 0), camera (0,0,3.4) looking -Z,
    yfov 45 degrees.  Winding chosen so geometric normals face into the box."""
    white, red, green, light = 0, 1, 2, 3
    materials = np.array([
        H.material_diffuse((0.73, 0.73, 0.73)),
        H.material_diffuse((0.65, 0.05, 0.05)),
        H.material_diffuse((0.12, 0.45, 0.15)),
        H.material_emissive((1.0, 1.0, 1.0), (1.0, 1.0, 1.0)),
    ], dtype=T.MATERIAL)

    def quad(a, b, c, d, mat):  # two triangles a-b-c, a-c-d
        return [(a, b, c, mat), (a, c, d, mat)]

    L = []
    L += quad((-1, -1, 1), (1, -1, 1), (1, -1, -1), (-1, -1, -1), white)     # floor,   normal +y
    L += quad((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1), white)         # ceiling, normal -y
    L += quad((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), white)     # back,    normal +z
    L += quad((-1, -1, 1), (-1, -1, -1), (-1, 1, -1), (-1, 1, 1), red)       # left,    normal +x
    L += quad((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1), green)         # right,   normal -x
    L += quad((-0.25, 0.999, -0.25), (0.25, 0.999, -0.25), (0.25, 0.999, 0.25), (-0.25, 0.999, 0.25), light)
    vertices, triangles = H.legacy_to_indexed(L)
    lights = np.array([H.light_point((0.0, 0.9, 0.0), (1.0, 1.0, 1.0), 1.0)], dtype=T.LIGHT)
    cam = H.camera((0.0, 0.0, 3.4), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), 45.0)
    return Scene("cornell12", np.zeros(0, T.SPHERE), lights, vertices, triangles, materials, cam)


def random_soup(n_tris, seed=1, extent=4.0, size=0.6, n_materials=5, n_spheres=0, n_lights=2):
    """Random triangle soup in a cube in front of the default camera."""
    i = np.arange(n_tris, dtype=np.uint64)
    c = np.stack([_u01(seed, k, i) for k in range(3)], 1).astype(np.float64) * extent - extent / 2
    c[:, 2] -= extent / 2 + 1.0
    p = [c + (np.stack([_u01(seed, 3 + 3 * j + k, i) for k in range(3)], 1).astype(np.float64) - 0.5) * size
         for j in range(3)]
    verts = np.stack(p, 1).reshape(-1, 3)
    mesh = _Mesh()
    mats = (_hash_u32(seed, 20, i) % np.uint32(max(n_materials, 1))).astype(np.uint32)
    mesh.add(verts, np.arange(n_tris * 3).reshape(-1, 3), mats)
    vertices, triangles = mesh.finish()
    materials = np.array([
        H.material_new(tuple(_u01(seed, 30 + m, 3) * 0.8 + 0.1), 1.0 if m % 3 == 1 else 0.0, 0.5,
                       (0.3, 0.2, 0.1) if m % 4 == 3 else (0, 0, 0), 1.5, 0.5 if m % 5 == 2 else 0.0)
        for m in range(max(n_materials, 1))], dtype=T.MATERIAL)
    spheres = np.zeros(n_spheres, dtype=T.SPHERE)
    for s in range(n_spheres):
        u = _u01(seed, 40 + s, 4)
        spheres[s] = ((u[0] * 3 - 1.5, u[1] * 3 - 1.5, -2.0 - u[2] * 3), 0.2 + 0.4 * u[3], s % max(n_materials, 1))
    lights = []
    for l in range(n_lights):
        u = _u01(seed, 60 + l, 6)
        if l % 3 == 0:
            lights.append(H.light_point((u[0] * 8 - 4, u[1] * 8 - 2, 2.0 + u[2] * 3), (1.0, 0.9, 0.8), 1.5))
        elif l % 3 == 1:
            lights.append(H.light_directional((u[0] - 0.5, -1.0, u[2] - 0.5), (0.6, 0.7, 1.0), 0.8))
        else:
            lights.append(H.light_spot((u[0] * 4 - 2, 3.0, 1.0), (0.0, -1.0, -0.5), (1.0, 1.0, 1.0), 2.0, 20.0, 0.3, 0.5))
    lights = np.array(lights, dtype=T.LIGHT) if lights else np.zeros(0, T.LIGHT)
    return Scene(f"soup{n_tris}", spheres, lights, vertices, triangles, materials, H.camera())


# --------------------------------------------------------------------------------------
SPONZA_SEED = 0x53504F4E
BISTRO_SEED = 0x42495354


def _palette(seed, n, special):
    """n factor-only materials: mostly diffuse, a few metallic / glass / emissive given by `special`."""
    mats = []
    for m in range(n):
        u = _u01(seed, 900 + m, 4)
        albedo = tuple(0.15 + 0.7 * u[:3])
        kind = special.get(m, "diffuse")
        if kind == "metal":
            mats.append(H.material_metallic(albedo, 0.2 + 0.5 * float(u[3])))
        elif kind == "glass":
            mats.append(H.material_glass(albedo, 1.5, 0.8))
        elif kind == "emissive":
            mats.append(H.material_emissive((1.0, 1.0, 1.0), (4.0, 3.6, 3.0)))
        else:
            mats.append(H.material_diffuse(albedo))
    return np.array(mats, dtype=T.MATERIAL)


def _leaves(mesh, seed, stream, n, center, radius, size, mat):
    """n tiny randomly oriented triangles in an ellipsoid of `radius` around `center` (foliage cluster)."""
    if n <= 0:
        return
    i = np.arange(n, dtype=np.uint64)
    u = [_u01(seed, stream + k, i).astype(np.float64) for k in range(12)]
    # point in unit ball by cube-root radius * direction from two uniforms (no rejection)
    z = 2 * u[0] - 1
    phi = 2 * np.pi * u[1]
    rr = np.cbrt(u[2])
    s = np.sqrt(np.maximum(0.0, 1 - z * z))
    c = np.asarray(center, np.float64) + np.stack([s * np.cos(phi), z, s * np.sin(phi)], 1) * rr[:, None] * np.asarray(radius)
    a = c + (np.stack(u[3:6], 1) - 0.5) * size
    b = c + (np.stack(u[6:9], 1) - 0.5) * size
    d = c + (np.stack(u[9:12], 1) - 0.5) * size
    verts = np.stack([a, b, d], 1).reshape(-1, 3)
    mesh.add(verts, np.arange(n * 3).reshape(-1, 3), mat)


def sponza_like(n_triangles=262144, seed=SPONZA_SEED):
    """'sponza-like' atrium: 30 x 12 x 14 units, x in [-15,15], y in [0,12], z in [-7,7]."""
    mesh = _Mesh()
    M_FLOOR, M_WALL, M_CEIL, M_COL, M_ARCH, M_SLAB = 0, 1, 2, 3, 4, 5
    M_CLOTH0, M_LEAF0, M_POT, M_BRONZE, M_GLASS, M_LAMP = 6, 12, 16, 17, 18, 19
    # shell, inward facing, 0.5-unit grid (large architectural triangles)
    X0, X1, Y0, Y1, Z0, Z1 = -15.0, 15.0, 0.0, 12.0, -7.0, 7.0
    mesh.add(*_grid((X0, Y0, Z0), (30, 0, 0), (0, 0, 14), 60, 28, flip=True), M_FLOOR)       # normal +y
    mesh.add(*_grid((X0, Y1, Z0), (30, 0, 0), (0, 0, 14), 30, 14, flip=False), M_CEIL)      # normal -y
    mesh.add(*_grid((X0, Y0, Z0), (30, 0, 0), (0, 12, 0), 60, 24, flip=False), M_WALL)      # z=Z0, normal +z
    mesh.add(*_grid((X0, Y0, Z1), (30, 0, 0), (0, 12, 0), 60, 24, flip=True), M_WALL)       # z=Z1, normal -z
    mesh.add(*_grid((X0, Y0, Z0), (0, 0, 14), (0, 12, 0), 28, 24, flip=True), M_WALL)       # x=X0, normal +x
    mesh.add(*_grid((X1, Y0, Z0), (0, 0, 14), (0, 12, 0), 28, 24, flip=False), M_WALL)      # x=X1, normal -x
    # gallery slabs on both long sides at y = 5.6 .. 6.0, 3 units deep
    for zlo, zhi in ((Z0, Z0 + 3.0), (Z1 - 3.0, Z1)):
        mesh.add(*_box((X0, 5.6, zlo), (X1, 6.0, zhi), (60, 1, 6)), M_SLAB)
    # colonnades: 2 storeys x 2 sides x 13 columns, fluted shafts with entasis, base and capital boxes
    n_cols = 13
    col_x = np.linspace(X0 + 1.5, X1 - 1.5, n_cols)
    col_seg, col_ring = 32, 34
    for storey, (ybase, h) in enumerate(((0.0, 5.6), (6.0, 5.0))):
        for side, zc in enumerate((Z0 + 3.0, Z1 - 3.0)):
            for ci, xc in enumerate(col_x):
                r0 = 0.34 if storey == 0 else 0.27

                def center(S, xc=xc, zc=zc, ybase=ybase, h=h):
                    o = np.stack([np.full_like(S, xc), ybase + 0.3 + S * (h - 0.6), np.full_like(S, zc)], -1)
                    ex = np.broadcast_to(np.array([1.0, 0, 0]), o.shape)
                    ez = np.broadcast_to(np.array([0, 0, 1.0]), o.shape)
                    return o, ex, ez

                def radius(S, TH, r0=r0):
                    return r0 * (1.0 - 0.12 * S * S) * (1.0 + 0.035 * np.cos(16 * TH))

                mesh.add(*_tube(center, radius, col_seg, col_ring), M_COL)
                mesh.add(*_box((xc - 0.45, ybase, zc - 0.45), (xc + 0.45, ybase + 0.3, zc + 0.45), (2, 1, 2)), M_COL)
                mesh.add(*_box((xc - 0.45, ybase + h - 0.3, zc - 0.45), (xc + 0.45, ybase + h, zc + 0.45), (2, 1, 2)), M_COL)
    # arches between neighbouring columns (half tori), both storeys and sides
    arch_seg, arch_ring = 12, 20
    for storey, ytop in enumerate((5.3, 10.7)):
        for zc in (Z0 + 3.0, Z1 - 3.0):
            for ci in range(n_cols - 1):
                xa, xb = col_x[ci], col_x[ci + 1]
                xm, R = 0.5 * (xa + xb), 0.5 * (xb - xa) - 0.2

                def center(S, xm=xm, R=R, ytop=ytop, zc=zc):
                    ang = np.pi * S
                    o = np.stack([xm - R * np.cos(ang), ytop - 1.2 + 1.1 * np.sin(ang), np.full_like(S, zc)], -1)
                    # frame: tangent in the xy plane -> ex = in-plane normal, ez = +z
                    ex = np.stack([-np.cos(ang), np.sin(ang) * 1.1 / R, np.zeros_like(S)], -1)
                    ex = ex / np.linalg.norm(ex, axis=-1, keepdims=True)
                    ez = np.broadcast_to(np.array([0, 0, 1.0]), o.shape)
                    return o, ex, ez

                mesh.add(*_tube(center, lambda S, TH: 0.16 + 0.0 * S, arch_seg, arch_ring), M_ARCH)
    # hanging cloth: wavy fine grids across the atrium
    n_cloth, cn = 6, 68
    for k in range(n_cloth):
        u = _u01(seed, 100 + k, 8).astype(np.float64)
        xk = X0 + 4.0 + k * (22.0 / (n_cloth - 1))
        amp, ph = 0.25 + 0.2 * u[0], 6.28 * u[1]

        def disp(a, b, amp=amp, ph=ph):
            sag = -1.6 * np.sin(np.pi * b) * (0.7 + 0.3 * np.sin(np.pi * a))
            wave = amp * np.sin(9.0 * a + ph) * np.sin(7.0 * b + 0.5 * ph)
            return np.stack([wave, sag, 0.15 * np.sin(11 * a + ph)], -1)

        mesh.add(*_grid((xk, 10.4, Z0 + 3.3), (1.6, 0, 0), (0, 0, (Z1 - Z0) - 6.6), cn // 4, cn * 2, disp=disp), M_CLOTH0 + k)
    # bronze ornament (bumpy sphere) and a glass panel in the middle of the atrium
    def orn_center(S):
        o = np.stack([np.zeros_like(S), 0.6 + 1.8 * S, np.zeros_like(S)], -1)
        return o, np.broadcast_to(np.array([1.0, 0, 0]), o.shape), np.broadcast_to(np.array([0, 0, 1.0]), o.shape)
    mesh.add(*_tube(orn_center, lambda S, TH: 0.9 * np.sqrt(np.maximum(1e-4, S * (1 - S))) * 2 * (1 + 0.08 * np.sin(9 * TH) * np.sin(14 * S)) + 0.02, 64, 64), M_BRONZE)
    mesh.add(*_box((-0.4, 0.0, -0.4), (0.4, 0.6, 0.4), (2, 2, 2)), M_POT)
    mesh.add(*_box((6.0, 0.0, -1.5), (6.08, 3.0, 1.5), (1, 6, 6)), M_GLASS)
    # lamp panels under the ceiling (emissive)
    for xl in (-9.0, 0.0, 9.0):
        mesh.add(*_box((xl - 0.8, 11.7, -0.8), (xl + 0.8, 11.8, 0.8), (2, 1, 2)), M_LAMP)
    # potted plants: foliage clusters of tiny triangles fill the remaining budget
    pots = [(x, z) for x in (-12.0, -6.0, 6.0, 12.0) for z in (-3.2, 3.2)]
    for (x, z) in pots:
        mesh.add(*_box((x - 0.3, 0.0, z - 0.3), (x + 0.3, 0.5, z + 0.3), (1, 1, 1)), M_POT)
    remaining = n_triangles - mesh.n_tris
    if remaining < 0:
        raise ValueError(f"sponza_like: structural geometry already has {mesh.n_tris} > {n_triangles} triangles")
    per = [remaining // len(pots) + (1 if i < remaining % len(pots) else 0) for i in range(len(pots))]
    for i, ((x, z), n) in enumerate(zip(pots, per)):
        _leaves(mesh, seed, 200 + 16 * i, n, (x, 1.5, z), (0.8, 1.0, 0.8), 0.09, M_LEAF0 + i % 4)
    vertices, triangles = mesh.finish()
    assert len(triangles) == n_triangles
    materials = _palette(seed, 25, {M_BRONZE: "metal", 20: "metal", M_GLASS: "glass", M_LAMP: "emissive"})
    lights = np.array([
        H.light_directional((0.3, -1.0, 0.2), (1.0, 0.95, 0.85), 0.9),
        H.light_point((-9.0, 10.5, 0.0), (1.0, 0.9, 0.8), 9.0),
        H.light_point((0.0, 10.5, 0.0), (1.0, 0.9, 0.8), 9.0),
        H.light_point((9.0, 10.5, 0.0), (1.0, 0.9, 0.8), 9.0),
        H.light_point((0.0, 3.0, 0.0), (0.8, 0.9, 1.0), 3.0),
    ], dtype=T.LIGHT)
    cam = H.camera((X0 + 1.2, 2.0, 0.4), (1.0, 0.06, -0.02), (0.0, 1.0, 0.0), 60.0)
    cam["direction"] = cam["direction"] / np.linalg.norm(cam["direction"])
    return Scene("sponza_like", np.zeros(0, T.SPHERE), lights, vertices, triangles, materials, cam,
                 {"seed": seed, "synthetic": True})


def bistro_like(n_triangles=3800000, seed=BISTRO_SEED):
    """'bistro-like' street block: 100 x 100 units, facades, trees with tiny-leaf
    foliage, many small props; deep BVH and heavy divergence."""
    mesh = _Mesh()
    M_GROUND, M_ROAD, M_FAC0, M_TRUNK, M_LEAF0, M_PROP0, M_AWN, M_LAMP, M_GLASS = 0, 1, 2, 10, 11, 16, 22, 23, 24
    mesh.add(*_grid((-50, 0, -50), (100, 0, 0), (0, 0, 100), 200, 200, flip=True), M_GROUND)
    # two crossing streets are kept free; buildings on a 6 x 6 lot grid
    lots = [(-42.0 + 15.0 * i, -42.0 + 15.0 * j) for i in range(6) for j in range(6)
            if not (i in (2, 3) and j in (2, 3))]
    for b, (x, z) in enumerate(lots):
        u = _u01(seed, 100 + b, 4).astype(np.float64)
        w, d, h = 9.0 + 3.0 * u[0], 9.0 + 3.0 * u[1], 8.0 + 10.0 * u[2]
        mesh.add(*_box((x, 0.0, z), (x + w, h, z + d), (36, 36, 36)), M_FAC0 + b % 8)
        # awning strip on the street-facing side
        mesh.add(*_grid((x, 3.0, z + d), (w, 0, 0), (0, -0.6, 1.5), 24, 6), M_AWN)
    # trees along the streets
    tree_xy = []
    for k in range(34):
        t = -48.0 + k * (96.0 / 33)
        tree_xy += [(t, -7.0), (t, 7.0), (-7.0, t), (7.0, t)]
    n_trees = len(tree_xy)
    for k, (x, z) in enumerate(tree_xy):
        def center(S, x=x, z=z):
            o = np.stack([np.full_like(S, x) + 0.15 * np.sin(3 * S), 4.0 * S, np.full_like(S, z)], -1)
            return o, np.broadcast_to(np.array([1.0, 0, 0]), o.shape), np.broadcast_to(np.array([0, 0, 1.0]), o.shape)
        mesh.add(*_tube(center, lambda S, TH: 0.22 * (1 - 0.5 * S) * (1 + 0.1 * np.cos(5 * TH)), 16, 24), M_TRUNK)
    # small props (tables / chairs as boxes) scattered on the pavements
    n_props = 2400
    pu = [_u01(seed, 300 + k, n_props).astype(np.float64) for k in range(4)]
    for p in range(n_props):
        x, z = -48.0 + 96.0 * pu[0][p], (-9.5 if pu[1][p] < 0.5 else 8.5) + pu[2][p]
        if p % 2:
            x, z = z, x
        s = 0.25 + 0.35 * pu[3][p]
        mesh.add(*_box((x, 0.0, z), (x + s, 0.75 * s + 0.3, z + s), (2, 2, 2)), M_PROP0 + p % 6)
    # street lamps (emissive boxes)
    for k in range(12):
        t = -44.0 + 8.0 * k
        mesh.add(*_box((t, 5.0, -0.3), (t + 0.4, 5.3, 0.3), (1, 1, 1)), M_LAMP)
    mesh.add(*_box((-3.0, 0.0, 12.0), (3.0, 3.0, 12.1), (8, 8, 1)), M_GLASS)
    remaining = n_triangles - mesh.n_tris
    if remaining < 0:
        raise ValueError(f"bistro_like: structural geometry already has {mesh.n_tris} > {n_triangles} triangles")
    per = [remaining // n_trees + (1 if i < remaining % n_trees else 0) for i in range(n_trees)]
    for k, ((x, z), n) in enumerate(zip(tree_xy, per)):
        _leaves(mesh, seed, 1000 + 16 * k, n, (x, 5.2, z), (2.2, 1.8, 2.2), 0.07, M_LEAF0 + k % 5)
    vertices, triangles = mesh.finish()
    assert len(triangles) == n_triangles
    materials = _palette(seed, 25, {M_LAMP: "emissive", M_GLASS: "glass", 21: "metal", 17: "metal"})
    lights = np.array([
        H.light_directional((0.4, -1.0, 0.3), (1.0, 0.96, 0.9), 1.0),
        H.light_point((0.0, 6.0, 0.0), (1.0, 0.8, 0.6), 12.0),
        H.light_point((-30.0, 6.0, 0.0), (1.0, 0.8, 0.6), 12.0),
        H.light_point((30.0, 6.0, 0.0), (1.0, 0.8, 0.6), 12.0),
    ], dtype=T.LIGHT)
    cam = H.camera((-46.0, 1.7, 0.5), (1.0, 0.05, -0.03), (0.0, 1.0, 0.0), 60.0)
    cam["direction"] = cam["direction"] / np.linalg.norm(cam["direction"])
    return Scene("bistro_like", np.zeros(0, T.SPHERE), lights, vertices, triangles, materials, cam,
                 {"seed": seed, "synthetic": True})


SCENES = {
    "default": default_scene, "empty": empty_scene, "single_triangle": single_triangle,
    "cornell12": cornell12, "sponza_like": sponza_like, "bistro_like": bistro_like,
}
