"""Deterministic procedural stand-ins for the reference's OBJ meshes.

The reference checkout carries only the .mtl files of its models (its .gitignore drops *.obj), so every mesh
named by a scene or a benchmark configuration is synthesised here in closed form: surfaces of revolution,
boxes, ellipsoids and prisms, assembled per model name and tagged with the material names of the real .mtl
files.  Output is triangles only, outward counter-clockwise in OBJ space, with per-corner normals, rounded to
the six decimals an OBJ writer prints so that the text round trip through a loader is exact.

A mesh is returned in the form the reference's loader produces (OBJ_Loader.cpp:329-357): one vertex per face
corner in file order, indices 0..n-1, and one "mesh" (material group) per `usemtl` block.
"""
import math
import os
from dataclasses import dataclass, field

import numpy as np

ASSET_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "models")


@dataclass
class Part:
    material: str
    P: np.ndarray  # [nt, 3, 3] float64 corner positions
    N: np.ndarray  # [nt, 3, 3] float64 corner normals


@dataclass
class LoadedMesh:
    """Loader-shaped arrays: per-corner vertices in file order."""
    name: str
    positions: np.ndarray      # [n, 3] float32
    normals: np.ndarray        # [n, 3] float32
    texcoords: np.ndarray      # [n, 2] float32
    ka: np.ndarray             # [n, 3] float32 (material of the vertex's mesh)
    kd: np.ndarray
    ks: np.ndarray
    indices: np.ndarray        # [ni] uint32
    groups: list = field(default_factory=list)   # (material name, first triangle, triangle count)

    @property
    def triangle_count(self):
        return len(self.indices) // 3


# ----------------------------------------------------------------------------- MTL

def parse_mtl(path):
    """Returns {material name: {'Ka','Kd','Ks': (r,g,b), 'map_Kd': str}} following OBJ_Loader.cpp:743-902."""
    mats, cur = {}, None
    with open(path) as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "newmtl":
                cur = {"Ka": (0.0, 0.0, 0.0), "Kd": (0.0, 0.0, 0.0), "Ks": (0.0, 0.0, 0.0), "map_Kd": ""}
                mats[" ".join(tok[1:]) if len(tok) > 1 else "none"] = cur
            elif cur is not None and tok[0] in ("Ka", "Kd", "Ks") and len(tok) == 4:
                cur[tok[0]] = tuple(float(np.float32(float(v))) for v in tok[1:4])
            elif cur is not None and tok[0] == "map_Kd":
                cur["map_Kd"] = " ".join(tok[1:])
    return mats


# ----------------------------------------------------------------------------- building blocks

def _fix_winding(P, N):
    """Orders each triangle counter-clockwise around its (averaged) corner normal."""
    g = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0])
    flip = np.einsum("ij,ij->i", g, N.sum(axis=1)) < 0
    P = P.copy(); N = N.copy()
    P[flip] = P[flip][:, [0, 2, 1]]
    N[flip] = N[flip][:, [0, 2, 1]]
    return P, N


def _grid_tris(Pg, Ng):
    """Two triangles per cell of a [nu, nv, 3] grid."""
    a, b, c, d = Pg[:-1, :-1], Pg[1:, :-1], Pg[1:, 1:], Pg[:-1, 1:]
    na, nb, nc, nd = Ng[:-1, :-1], Ng[1:, :-1], Ng[1:, 1:], Ng[:-1, 1:]
    P = np.concatenate([np.stack([a, b, c], axis=2).reshape(-1, 3, 3), np.stack([a, c, d], axis=2).reshape(-1, 3, 3)])
    N = np.concatenate([np.stack([na, nb, nc], axis=2).reshape(-1, 3, 3), np.stack([na, nc, nd], axis=2).reshape(-1, 3, 3)])
    return P, N


def lathe(profile, segments, material, sub=1):
    """Surface of revolution about +Y of the polyline `profile` [(r, y), ...], each span split into `sub` rings.
    The surface normal points to the right of the profile's direction of travel (so list the outer wall
    bottom-to-top, the inner wall top-to-bottom)."""
    pts = np.asarray(profile, dtype=np.float64)
    rs, ys, nr, ny = [], [], [], []
    for i in range(len(pts) - 1):
        (r0, y0), (r1, y1) = pts[i], pts[i + 1]
        dr, dy = r1 - r0, y1 - y0
        ln = math.hypot(dr, dy) or 1.0
        for k in range(sub + 1):
            t = k / sub
            rs.append(r0 + dr * t); ys.append(y0 + dy * t); nr.append(dy / ln); ny.append(-dr / ln)
        # spans are emitted separately (hard edges between spans)
    rs, ys, nr, ny = map(np.asarray, (rs, ys, nr, ny))
    th = np.linspace(0.0, 2.0 * math.pi, segments + 1)
    c, s = np.cos(th), np.sin(th)
    Ps, Ns = [], []
    per = sub + 1
    for i in range(len(pts) - 1):
        sl = slice(i * per, (i + 1) * per)
        Pg = np.stack([np.outer(rs[sl], c), np.outer(ys[sl], np.ones_like(c)), np.outer(rs[sl], s)], axis=-1)
        Ng = np.stack([np.outer(nr[sl], c), np.outer(ny[sl], np.ones_like(c)), np.outer(nr[sl], s)], axis=-1)
        P, N = _grid_tris(Pg, Ng)
        Ps.append(P); Ns.append(N)
    P, N = np.concatenate(Ps), np.concatenate(Ns)
    area2 = np.linalg.norm(np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0]), axis=1)
    keep = area2 > 1e-12     # drops the collapsed triangles at the axis
    P, N = _fix_winding(P[keep], N[keep])
    return Part(material, P, N)


def box(lo, hi, material, sub=1):
    """Axis-aligned box with `sub` x `sub` cells per face."""
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    Ps, Ns = [], []
    u = np.linspace(0.0, 1.0, sub + 1)
    for axis in range(3):
        a1, a2 = (axis + 1) % 3, (axis + 2) % 3
        for side, val in ((-1.0, lo[axis]), (1.0, hi[axis])):
            Pg = np.zeros((sub + 1, sub + 1, 3)); Ng = np.zeros_like(Pg)
            Pg[..., axis] = val
            Pg[..., a1] = (lo[a1] + (hi[a1] - lo[a1]) * u)[:, None]
            Pg[..., a2] = (lo[a2] + (hi[a2] - lo[a2]) * u)[None, :]
            Ng[..., axis] = side
            P, N = _grid_tris(Pg, Ng)
            Ps.append(P); Ns.append(N)
    P, N = _fix_winding(np.concatenate(Ps), np.concatenate(Ns))
    return Part(material, P, N)


def ellipsoid(centre, radii, material, nu=16, nv=8):
    centre, radii = np.asarray(centre, float), np.asarray(radii, float)
    th = np.linspace(0.0, 2.0 * math.pi, nu + 1)
    ph = np.linspace(0.0, math.pi, nv + 1)
    d = np.stack([np.outer(np.sin(ph), np.cos(th)), np.outer(np.cos(ph), np.ones_like(th)), np.outer(np.sin(ph), np.sin(th))], axis=-1)
    Pg = centre + d * radii
    Ng = d / radii
    Ng = Ng / np.linalg.norm(Ng, axis=-1, keepdims=True)
    P, N = _grid_tris(Pg, Ng)
    area2 = np.linalg.norm(np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0]), axis=1)
    keep = area2 > 1e-12
    P, N = _fix_winding(P[keep], N[keep])
    return Part(material, P, N)


def transformed(part, rot=None, translate=(0.0, 0.0, 0.0), scale=1.0):
    R = np.eye(3) if rot is None else np.asarray(rot, float)
    P = (part.P * scale) @ R.T + np.asarray(translate, float)
    N = part.N @ R.T
    return Part(part.material, P, N)


def rot_axis(axis, angle):
    axis = np.asarray(axis, float); axis = axis / np.linalg.norm(axis)
    x, y, z = axis; c, s = math.cos(angle), math.sin(angle); C = 1 - c
    return np.array([[c + x * x * C, x * y * C - z * s, x * z * C + y * s],
                     [y * x * C + z * s, c + y * y * C, y * z * C - x * s],
                     [z * x * C - y * s, z * y * C + x * s, c + z * z * C]])


# ----------------------------------------------------------------------------- models

def _detail(d, base):
    return max(3, int(round(base * d)))


def gen_cube(detail=1.0):
    return [box((-1, -1, -1), (1, 1, 1), "Material", 1)]


def gen_chalice(detail=1.0):
    seg, sub = _detail(detail, 48), _detail(detail, 6)
    outer = [(0.0, 0.0), (0.45, 0.0), (0.45, 0.05), (0.12, 0.12), (0.07, 0.55), (0.12, 0.62), (0.42, 0.95), (0.48, 1.30)]
    inner = [(0.48, 1.30), (0.44, 1.30), (0.38, 0.98), (0.10, 0.70), (0.0, 0.68)]
    return [lathe(outer, seg, "Gold", sub), lathe(inner, seg, "Gold", sub)]


def gen_vase(detail=1.0):
    seg, n = _detail(detail, 160), _detail(detail, 70)
    ys = np.linspace(0.0, 1.5, n + 1)
    r = 0.28 + 0.22 * np.sin(ys * 2.3 + 0.3) ** 2 + 0.05 * np.cos(ys * 9.0)
    outer = [(0.0, 0.0)] + [(float(a), float(b)) for a, b in zip(r, ys)]
    inner = [(float(a) - 0.04, float(b)) for a, b in zip(r[::-1], ys[::-1]) if b > 0.08] + [(0.0, 0.08)]
    return [lathe(outer, seg, "Vase", 1), lathe([outer[-1], inner[0]], seg, "Vase", 1), lathe(inner, seg, "Vase", 1)]


def gen_crown(detail=1.0):
    seg, sub = _detail(detail, 256), _detail(detail, 24)
    band = [(0.52, 0.0), (0.62, 0.0), (0.64, 0.35), (0.60, 0.35), (0.50, 0.33), (0.52, 0.0)]
    parts = [lathe(band, seg, "Gold", sub)]
    spikes = 8
    cseg, csub = _detail(detail, 48), _detail(detail, 24)
    for k in range(spikes):
        a = 2.0 * math.pi * k / spikes
        cone = lathe([(0.0, 0.0), (0.11, 0.0), (0.0, 0.45)], cseg, "Gold", csub)
        parts.append(transformed(cone, translate=(0.61 * math.cos(a), 0.33, 0.61 * math.sin(a))))
        parts.append(ellipsoid((0.61 * math.cos(a), 0.80, 0.61 * math.sin(a)), (0.045, 0.045, 0.045), "Gold",
                               _detail(detail, 32), _detail(detail, 16)))
    return parts


def gen_crystal3(detail=1.0):
    parts = []
    sub = _detail(detail, 28)
    shards = [((0.0, 0.0, 0.0), 0.22, 1.25, (0, 0, 1), 0.0), ((0.28, 0.0, 0.10), 0.15, 0.85, (0, 0, 1), -0.45),
              ((-0.25, 0.0, -0.12), 0.17, 0.95, (1, 0, 0), 0.40), ((0.05, 0.0, 0.30), 0.12, 0.70, (1, 0, 0), -0.55),
              ((-0.10, 0.0, -0.32), 0.13, 0.60, (1, 0, 1), 0.50)]
    for (cx, cy, cz), r, h, axis, ang in shards:
        prof = [(0.0, 0.0), (r, 0.0), (r, h * 0.72), (0.0, h)]
        shard = lathe(prof, 6, "Green", sub)       # six flat sides with a pointed tip
        parts.append(transformed(shard, rot=rot_axis(axis, ang), translate=(cx, cy, cz)))
    return parts


def gen_table(detail=1.0):
    s = _detail(detail, 6)
    parts = [box((-1.0, 0.90, -0.6), (1.0, 1.0, 0.6), "Wood", s)]
    for x in (-0.9, 0.8):
        for z in (-0.5, 0.4):
            parts.append(box((x, 0.0, z), (x + 0.1, 0.90, z + 0.1), "DarkWood", max(1, s // 2)))
    return parts


def gen_book2_open(detail=1.0):
    s = _detail(detail, 4)
    left = transformed(box((-0.6, 0.0, -0.4), (0.0, 0.06, 0.4), "DarkRed", s), rot=rot_axis((0, 0, 1), 0.15))
    right = transformed(box((0.0, 0.0, -0.4), (0.6, 0.06, 0.4), "DarkRed", s), rot=rot_axis((0, 0, 1), -0.15))
    pages_l = transformed(box((-0.55, 0.06, -0.36), (0.0, 0.10, 0.36), "Beige", s), rot=rot_axis((0, 0, 1), 0.15))
    pages_r = transformed(box((0.0, 0.06, -0.36), (0.55, 0.10, 0.36), "Beige", s), rot=rot_axis((0, 0, 1), -0.15))
    clasp = box((-0.03, 0.0, -0.42), (0.03, 0.05, -0.40), "Golden", 1)
    return [left, right, pages_l, pages_r, clasp]


def gen_chest_gold(detail=1.0):
    s = _detail(detail, 6)
    body = box((-0.8, 0.0, -0.5), (0.8, 0.7, 0.5), "Wood", s)
    lid = transformed(lathe([(0.0, -0.8), (0.5, -0.8), (0.5, 0.8), (0.0, 0.8)], _detail(detail, 32), "Wood", s),
                      rot=rot_axis((0, 0, 1), math.pi / 2), translate=(0.0, 0.7, 0.0))
    band1 = box((-0.82, 0.0, -0.52), (-0.7, 0.72, 0.52), "Steel", 2)
    band2 = box((0.7, 0.0, -0.52), (0.82, 0.72, 0.52), "DarkSteel", 2)
    lock = box((-0.08, 0.45, 0.5), (0.08, 0.65, 0.56), "Gold", 1)
    return [body, lid, band1, band2, lock]


def gen_pizza(detail=1.0):
    return [lathe([(0.0, 0.0), (1.0, 0.0), (1.0, 0.08), (0.9, 0.12), (0.0, 0.10)], _detail(detail, 64), "Material.001", 2)]


def gen_pug(detail=1.0):
    nu, nv = _detail(detail, 32), _detail(detail, 16)
    return [ellipsoid((0, 0.45, 0), (0.55, 0.38, 0.35), "Beige", nu, nv), ellipsoid((0.6, 0.75, 0), (0.28, 0.26, 0.25), "Beige", nu, nv),
            ellipsoid((0.82, 0.70, 0), (0.12, 0.10, 0.12), "Brown", nu // 2, nv // 2), ellipsoid((0.55, 1.0, 0.15), (0.08, 0.12, 0.05), "Brown", nu // 2, nv // 2),
            ellipsoid((0.55, 1.0, -0.15), (0.08, 0.12, 0.05), "Brown", nu // 2, nv // 2)] + \
           [box((x, 0.0, z), (x + 0.12, 0.3, z + 0.12), "Beige", 1) for x in (-0.4, 0.3) for z in (-0.25, 0.13)]


def gen_bookcase(detail=1.0):
    s = _detail(detail, 4)
    parts = [box((-1.0, 0.0, -0.3), (1.0, 2.0, -0.25), "DarkWood.005", s), box((-1.0, 0.0, -0.3), (-0.95, 2.0, 0.3), "Wood.007", s),
             box((0.95, 0.0, -0.3), (1.0, 2.0, 0.3), "Wood.007", s)]
    covers = ["Cover2.001", "Cover3.001", "Cover4.001", "Sofa.001"]
    for shelf in range(4):
        y = 0.5 * shelf
        parts.append(box((-0.95, y, -0.25), (0.95, y + 0.05, 0.3), "Wood.007", s))
        for b in range(12):
            x = -0.9 + 0.15 * b
            hgt = 0.30 + 0.05 * ((b * 7 + shelf * 3) % 4)
            parts.append(box((x, y + 0.05, -0.2), (x + 0.12, y + 0.05 + hgt, 0.2), covers[(b + shelf) % 4], 1))
            parts.append(box((x + 0.01, y + 0.05 + hgt, -0.19), (x + 0.11, y + 0.06 + hgt, 0.19), "Pages.001", 1))
    return parts


def gen_chicken(detail=1.0):
    nu, nv = _detail(detail, 24), _detail(detail, 12)
    return [ellipsoid((0, 0.5, 0), (0.4, 0.35, 0.3), "Atlas", nu, nv), ellipsoid((0.35, 0.9, 0), (0.18, 0.2, 0.17), "Atlas", nu, nv),
            lathe([(0.0, 0.0), (0.06, 0.0), (0.0, 0.15)], 8, "Atlas", 1)]


def gen_woodlog(detail=1.0):
    log = transformed(lathe([(0.0, -1.0), (0.3, -1.0), (0.3, 1.0), (0.0, 1.0)], _detail(detail, 32), "Wood", _detail(detail, 8)),
                      rot=rot_axis((0, 0, 1), math.pi / 2), translate=(0, 0.3, 0))
    stem = lathe([(0.0, 0.0), (0.04, 0.0), (0.04, 0.2)], 8, "Mushroom_Bottom", 1)
    cap = ellipsoid((0, 0.22, 0), (0.12, 0.07, 0.12), "Mushroom_Top", _detail(detail, 16), _detail(detail, 8))
    return [log, transformed(stem, translate=(0.3, 0.58, 0.0)), transformed(cap, translate=(0.3, 0.58, 0.0))]


GENERATORS = {
    "cube": gen_cube, "Chalice": gen_chalice, "Vase": gen_vase, "Crown": gen_crown, "Crystal3": gen_crystal3,
    "Table": gen_table, "Book2_Open": gen_book2_open, "Chest_gold": gen_chest_gold, "Pizza": gen_pizza, "Pug": gen_pug,
    "BookCaseLargeBooks": gen_bookcase, "Chicken": gen_chicken, "WoodLog": gen_woodlog,
}


def _group_parts(parts):
    """Merges consecutive parts that share a material into one usemtl block, keeping file order."""
    groups = []
    for p in parts:
        if groups and groups[-1][0] == p.material:
            groups[-1] = (p.material, np.concatenate([groups[-1][1], p.P]), np.concatenate([groups[-1][2], p.N]))
        else:
            groups.append((p.material, p.P, p.N))
    return groups


def generate(name, detail=1.0, mtl_dir=ASSET_DIR):
    """Builds model `name` and returns it as loader-shaped arrays (see LoadedMesh)."""
    parts = GENERATORS[name](detail)
    mats = parse_mtl(os.path.join(mtl_dir, name + ".mtl"))
    groups = _group_parts(parts)
    pos, nrm, ka, kd, ks, ginfo = [], [], [], [], [], []
    first = 0
    for material, P, N in groups:
        P = np.round(P, 6) + 0.0
        N = np.round(N / np.maximum(np.linalg.norm(N, axis=-1, keepdims=True), 1e-30), 6) + 0.0
        m = mats[material]
        n = P.shape[0] * 3
        pos.append(P.reshape(-1, 3)); nrm.append(N.reshape(-1, 3))
        ka.append(np.tile(np.asarray(m["Ka"], np.float32), (n, 1)))
        kd.append(np.tile(np.asarray(m["Kd"], np.float32), (n, 1)))
        ks.append(np.tile(np.asarray(m["Ks"], np.float32), (n, 1)))
        ginfo.append((material, first, P.shape[0]))
        first += P.shape[0]
    positions = np.concatenate(pos).astype(np.float32)
    normals = np.concatenate(nrm).astype(np.float32)
    n = positions.shape[0]
    return LoadedMesh(name, positions, normals, np.zeros((n, 2), np.float32), np.concatenate(ka), np.concatenate(kd),
                      np.concatenate(ks), np.arange(n, dtype=np.uint32), ginfo)


def write_obj(mesh, path):
    """Writes `mesh` as Blender-style OBJ text (v / vn / usemtl / f a//a b//b c//c), one v and vn per corner."""
    with open(path, "w") as f:
        f.write(f"# synthetic stand-in for {mesh.name}.obj\nmtllib {mesh.name}.mtl\no {mesh.name}\n")
        for p in mesh.positions.astype(np.float64):
            f.write("v %.6f %.6f %.6f\n" % tuple(p))
        for q in mesh.normals.astype(np.float64):
            f.write("vn %.6f %.6f %.6f\n" % tuple(q))
        for material, first, count in mesh.groups:
            f.write(f"usemtl {material}\ns off\n")
            for t in range(first, first + count):
                a = 3 * t + 1
                f.write(f"f {a}//{a} {a + 1}//{a + 1} {a + 2}//{a + 2}\n")
