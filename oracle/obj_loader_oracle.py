"""CPU restatement of the reference's OBJ / MTL loader, for tests only.

TEST INFRASTRUCTURE - not part of the product.  Only tests/ may import this module; the product's loader is
cuda-ray-tracer_amd/host/OBJ_Loader.cpp, a different (single-pass) parser whose OUTPUT must equal what this restatement
produces, because the loader fixes vertex and triangle order and therefore which triangle wins an exact tie and which
triangle of a model is the never-tested last one.

Pinned: the reference holds no test, fixture or golden output for its loader, but its OBJ_Loader.cpp compiles in this image
(oracle/build_ref.sh -> oracle/_ref/libref.so), and tests/test_loader_oracle.py loads every case with the reference's own
objl::Loader as well: host loader, this restatement and the reference agree bit for bit.  The file follows the reference's
text statement by statement (file:line cited per
function, paths relative to the reference checkout), including the behaviour of its string helpers:

  algorithm::split       OBJ_Loader.h:193-229     algorithm::tail / firstToken   OBJ_Loader.h:232-266
  algorithm::getElement  OBJ_Loader.h:269-278     math:: / algorithm:: vector helpers   OBJ_Loader.cpp:96-174
  Loader::LoadFile       OBJ_Loader.cpp:183-465   Loader::GenVerticesFromRawOBJ  OBJ_Loader.cpp:467-574
  Loader::VertexTriangluation  OBJ_Loader.cpp:576-741   Loader::LoadMaterials    OBJ_Loader.cpp:743-902
  Mesh::Mesh(vertices, indices) keeps the vertices and drops the indices   OBJ_Loader.cpp:91-94

Arithmetic: IEEE binary32, one rounding per operation (numpy float32 scalars); std::stof is libc's strtof (called through
ctypes, so that decimal text is rounded to float once, not via double).  powf(x, 2) is x * x correctly rounded.
Where the reference would run into undefined behaviour (an index past the end of a vector, std::stof of an empty token)
this restatement raises LoaderUndefined; tests keep to inputs the reference defines.
The reference is a Windows program reading in text mode: "\\r\\n" reaches std::getline as "\\n".  load_file does the same.
"""
import ctypes
import ctypes.util
import os

import numpy as np

f32 = np.float32
_libc = ctypes.CDLL(ctypes.util.find_library("c") or "libc.so.6")
_libc.strtof.restype = ctypes.c_float
_libc.strtof.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p)]
_libc.strtol.restype = ctypes.c_long
_libc.strtol.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int]


class LoaderUndefined(Exception):
    """The reference's behaviour on this input is undefined (out-of-range access) or an uncaught C++ exception."""


def stof(s):
    """std::stof: strtof on the text, std::invalid_argument if nothing converts."""
    raw = s.encode("latin-1")
    end = ctypes.c_char_p()
    buf = ctypes.create_string_buffer(raw)
    v = _libc.strtof(buf, ctypes.byref(end))
    consumed = ctypes.cast(end, ctypes.c_void_p).value - ctypes.addressof(buf)
    if consumed == 0:
        raise LoaderUndefined(f"std::stof({s!r}) throws")
    return f32(v)


def stoi(s):
    raw = s.encode("latin-1")
    end = ctypes.c_char_p()
    buf = ctypes.create_string_buffer(raw)
    v = _libc.strtol(buf, ctypes.byref(end), 10)
    consumed = ctypes.cast(end, ctypes.c_void_p).value - ctypes.addressof(buf)
    if consumed == 0:
        raise LoaderUndefined(f"std::stoi({s!r}) throws")
    return int(v)


# ----------------------------------------------------------------------------- string helpers

def split(text, token):
    """OBJ_Loader.h:193-229, including its treatment of repeated separators (an empty entry per extra separator)."""
    out = []
    temp = ""
    n = len(token)
    i = 0
    while i < len(text):
        test = text[i:i + n]
        if test == token:
            if temp != "":
                out.append(temp)
                temp = ""
                i += n - 1
            else:
                out.append("")
        elif i + n >= len(text):
            temp += text[i:i + n]
            out.append(temp)
            break
        else:
            temp += text[i]
        i += 1
    return out


def _find_first_not_of(s, chars, start=0):
    if start is None:
        return None
    for i in range(start, len(s)):
        if s[i] not in chars:
            return i
    return None


def _find_first_of(s, chars, start=0):
    if start is None:      # std::string::npos as a start position finds nothing
        return None
    for i in range(start, len(s)):
        if s[i] in chars:
            return i
    return None


def _find_last_not_of(s, chars):
    for i in range(len(s) - 1, -1, -1):
        if s[i] not in chars:
            return i
    return None


def tail(line):
    """OBJ_Loader.h:232-247."""
    token_start = _find_first_not_of(line, " \t")
    space_start = _find_first_of(line, " \t", token_start)
    tail_start = _find_first_not_of(line, " \t", space_start)
    tail_end = _find_last_not_of(line, " \t")
    if tail_start is not None and tail_end is not None:
        return line[tail_start:tail_end + 1]
    if tail_start is not None:
        return line[tail_start:]
    return ""


def first_token(line):
    """OBJ_Loader.h:250-266."""
    if line != "":
        token_start = _find_first_not_of(line, " \t")
        token_end = _find_first_of(line, " \t", token_start)
        if token_start is not None and token_end is not None:
            return line[token_start:token_end]
        if token_start is not None:
            return line[token_start:]
    return ""


def get_element(elements, index):
    """OBJ_Loader.h:269-278."""
    idx = stoi(index)
    idx = len(elements) + idx if idx < 0 else idx - 1
    if idx < 0 or idx >= len(elements):
        raise LoaderUndefined(f"element {index!r} of {len(elements)}")
    return elements[idx]


# ----------------------------------------------------------------------------- vector helpers (float32)

def v3(x=0.0, y=0.0, z=0.0):
    return (f32(x), f32(y), f32(z))


def sub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def cross(a, b):          # OBJ_Loader.cpp:96-101
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def dot(a, b):            # OBJ_Loader.cpp:108-111
    return (a[0] * b[0]) + (a[1] * b[1]) + (a[2] * b[2])


def magnitude(a):         # OBJ_Loader.cpp:103-106; powf(x, 2) == x * x in float
    return np.sqrt((a[0] * a[0]) + (a[1] * a[1]) + (a[2] * a[2]))


def same_side(p1, p2, a, b):   # OBJ_Loader.cpp:131-140
    return bool(dot(cross(sub(b, a), sub(p1, a)), cross(sub(b, a), sub(p2, a))) >= 0)


def in_triangle(point, t1, t2, t3):     # OBJ_Loader.cpp:152-174
    if not (same_side(point, t1, t2, t3) and same_side(point, t2, t1, t3) and same_side(point, t3, t1, t2)):
        return False
    n = cross(sub(t2, t1), sub(t3, t1))                                  # GenTriNormal, :142-150
    with np.errstate(all="ignore"):
        m = magnitude(n)
        bn = (n[0] / m, n[1] / m, n[2] / m)                              # ProjV3, :120-124
        d = dot(point, bn)
        proj = (bn[0] * d, bn[1] * d, bn[2] * d)
        return bool(magnitude(proj) == 0)


def same(a, b):
    return bool(a[0] == b[0] and a[1] == b[1] and a[2] == b[2])


# ----------------------------------------------------------------------------- data

class Vertex:
    __slots__ = ("Position", "Normal", "TextureCoordinate")

    def __init__(self, position=None, normal=None, tex=None):
        self.Position = position if position is not None else v3()
        self.Normal = normal if normal is not None else v3()
        self.TextureCoordinate = tex if tex is not None else (f32(0), f32(0))

    def copy(self):
        return Vertex(self.Position, self.Normal, self.TextureCoordinate)

    def floats(self):
        return [*self.Position, *self.Normal, *self.TextureCoordinate]


class Material:
    def __init__(self):     # OBJ_Loader.cpp:80-87
        self.name = ""
        self.Ka = v3(); self.Kd = v3(); self.Ks = v3()
        self.Ns = f32(0); self.Ni = f32(0); self.d = f32(0)
        self.illum = 0
        self.map_Ka = self.map_Kd = self.map_Ks = self.map_Ns = self.map_d = self.map_bump = ""


class Mesh:
    def __init__(self, vertices=None):
        self.MeshName = ""
        self.Vertices = [v.copy() for v in vertices] if vertices else []
        self.Indices = []                # Mesh::Mesh(vertices, indices) never stores the indices (:91-94)
        self.MeshMaterial = Material()


# ----------------------------------------------------------------------------- triangulation

def vertex_triangulation(i_verts):
    """OBJ_Loader.cpp:576-741.  Returns the local index list."""
    o = []
    if len(i_verts) < 3:
        return o
    if len(i_verts) == 3:
        return [0, 1, 2]
    t_verts = [v.copy() for v in i_verts]

    def emit(a, b, c, limit):
        for j in range(limit):
            if same(i_verts[j].Position, a):
                o.append(j)
            if same(i_verts[j].Position, b):
                o.append(j)
            if same(i_verts[j].Position, c):
                o.append(j)

    rounds = 0
    while True:
        rounds += 1
        if rounds > 4 * len(i_verts) + 8:
            raise LoaderUndefined("the reference's ear clipping makes no progress on this polygon (it would loop for ever)")
        i = 0
        while i < len(t_verts):
            p_prev = t_verts[len(t_verts) - 1] if i == 0 else t_verts[i - 1]
            p_cur = t_verts[i]
            p_next = t_verts[0] if i == len(t_verts) - 1 else t_verts[i + 1]
            if len(t_verts) == 3:                                         # :630-645 (scans tVerts.size() entries of iVerts)
                emit(p_cur.Position, p_prev.Position, p_next.Position, len(t_verts))
                t_verts = []
                break
            if len(t_verts) == 4:                                         # :646-684
                emit(p_cur.Position, p_prev.Position, p_next.Position, len(i_verts))
                temp_vec = v3()
                for tv in t_verts:
                    if not same(tv.Position, p_cur.Position) and not same(tv.Position, p_prev.Position) and not same(tv.Position, p_next.Position):
                        temp_vec = tv.Position
                        break
                emit(p_prev.Position, p_next.Position, temp_vec, len(i_verts))
                t_verts = []
                break
            # :687-689: "angle <= 0 && angle >= 180" is never true, the vertex is never skipped here
            in_tri = False
            for j in range(len(i_verts)):                                 # :692-705
                q = i_verts[j].Position
                if in_triangle(q, p_prev.Position, p_cur.Position, p_next.Position) and not same(q, p_prev.Position) \
                        and not same(q, p_cur.Position) and not same(q, p_next.Position):
                    in_tri = True
                    break
            if in_tri:
                i += 1
                continue
            emit(p_cur.Position, p_prev.Position, p_next.Position, len(i_verts))     # :708-716
            for j in range(len(t_verts)):                                 # :719-726
                if same(t_verts[j].Position, p_cur.Position):
                    del t_verts[j]
                    break
            i = 0                                                         # :730 (i = -1, then the loop's increment)
        if len(o) == 0:
            break
        if len(t_verts) == 0:
            break
    return o


# ----------------------------------------------------------------------------- loader

class Loader:
    def __init__(self):
        self.LoadedMeshes = []
        self.LoadedVertices = []
        self.LoadedIndices = []
        self.LoadedMaterials = []

    @staticmethod
    def _lines(path):
        with open(path, "rb") as f:
            text = f.read().decode("latin-1")
        text = text.replace("\r\n", "\n")            # Windows text mode
        lines = text.split("\n")
        if lines and lines[-1] == "":                # std::getline yields nothing after a final newline
            lines.pop()
        return lines

    def gen_vertices_from_raw_obj(self, positions, tcoords, normals, curline):
        """OBJ_Loader.cpp:467-574."""
        o_verts = []
        v_vert = Vertex()                            # one object reused for every corner (:474)
        sface = split(tail(curline), " ")
        no_normal = False
        vtype = None
        for corner in sface:
            svert = split(corner, "/")
            if len(svert) == 1:
                vtype = 1
            if len(svert) == 2:
                vtype = 2
            if len(svert) == 3:
                vtype = 4 if svert[1] != "" else 3
            if vtype is None:
                raise LoaderUndefined("vtype read before it is set")
            if vtype in (1, 2, 3, 4) and len(svert) == 0:
                raise LoaderUndefined("empty corner")
            if vtype == 1:
                v_vert.Position = get_element(positions, svert[0])
                v_vert.TextureCoordinate = (f32(0), f32(0))
                no_normal = True
                o_verts.append(v_vert.copy())
            elif vtype == 2:
                v_vert.Position = get_element(positions, svert[0])
                v_vert.TextureCoordinate = get_element(tcoords, svert[1])
                no_normal = True
                o_verts.append(v_vert.copy())
            elif vtype == 3:
                v_vert.Position = get_element(positions, svert[0])
                v_vert.TextureCoordinate = (f32(0), f32(0))
                v_vert.Normal = get_element(normals, svert[2])
                o_verts.append(v_vert.copy())
            elif vtype == 4:
                v_vert.Position = get_element(positions, svert[0])
                v_vert.TextureCoordinate = get_element(tcoords, svert[1])
                v_vert.Normal = get_element(normals, svert[2])
                o_verts.append(v_vert.copy())
        if no_normal:                                # :562-573
            if len(o_verts) < 3:
                raise LoaderUndefined("face normal of fewer than three corners")
            a = sub(o_verts[0].Position, o_verts[1].Position)
            b = sub(o_verts[2].Position, o_verts[1].Position)
            normal = cross(a, b)
            for v in o_verts:
                v.Normal = normal
        return o_verts

    def load_file(self, path):
        """OBJ_Loader.cpp:183-465."""
        if path[-4:] != ".obj":
            return False
        if not os.path.isfile(path):
            return False
        self.LoadedMeshes = []; self.LoadedVertices = []; self.LoadedIndices = []
        positions, tcoords, normals = [], [], []
        vertices, indices = [], []
        mesh_mat_names = []
        listening = False
        meshname = ""
        for curline in self._lines(path):
            ft = first_token(curline)
            if ft == "o" or ft == "g" or curline[:1] == "g":              # :238-289
                if not listening:
                    listening = True
                    meshname = tail(curline) if ft in ("o", "g") else "unnamed"
                else:
                    if indices and vertices:
                        m = Mesh(vertices); m.MeshName = meshname
                        self.LoadedMeshes.append(m)
                        vertices = []; indices = []
                        meshname = tail(curline)
                    else:
                        meshname = tail(curline) if ft in ("o", "g") else "unnamed"
            if ft == "v":                                                 # :291-302
                s = split(tail(curline), " ")
                if len(s) < 3:
                    raise LoaderUndefined("v with fewer than three tokens")
                positions.append((stof(s[0]), stof(s[1]), stof(s[2])))
            if ft == "vt":                                                # :304-314
                s = split(tail(curline), " ")
                if len(s) < 2:
                    raise LoaderUndefined("vt with fewer than two tokens")
                tcoords.append((stof(s[0]), stof(s[1])))
            if ft == "vn":                                                # :316-327
                s = split(tail(curline), " ")
                if len(s) < 3:
                    raise LoaderUndefined("vn with fewer than three tokens")
                normals.append((stof(s[0]), stof(s[1]), stof(s[2])))
            if ft == "f":                                                 # :329-357
                v_verts = self.gen_vertices_from_raw_obj(positions, tcoords, normals, curline)
                for v in v_verts:
                    vertices.append(v.copy())
                    self.LoadedVertices.append(v.copy())
                i_indices = vertex_triangulation(v_verts)
                for li in i_indices:
                    indices.append((len(vertices) - len(v_verts)) + li)
                    self.LoadedIndices.append((len(self.LoadedVertices) - len(v_verts)) + li)
            if ft == "usemtl":                                            # :359-390
                mesh_mat_names.append(tail(curline))
                if indices and vertices:
                    m = Mesh(vertices)
                    m.MeshName = meshname + "_" + str(2)                  # the search loop of :370-377 leaves at i = 2
                    self.LoadedMeshes.append(m)
                    vertices = []; indices = []
            if ft == "mtllib":                                            # :392-419
                temp = split(path, "/")
                pathtomat = ""
                if len(temp) != 1:
                    for part in temp[:-1]:
                        pathtomat += part + "/"
                pathtomat += tail(curline)
                self.load_materials(pathtomat)
        if indices and vertices:                                          # :428-436
            m = Mesh(vertices); m.MeshName = meshname
            self.LoadedMeshes.append(m)
        for i, matname in enumerate(mesh_mat_names):                      # :441-455
            for mat in self.LoadedMaterials:
                if mat.name == matname:
                    if i >= len(self.LoadedMeshes):
                        raise LoaderUndefined("more usemtl lines than meshes")
                    self.LoadedMeshes[i].MeshMaterial = mat
                    break
        return not (not self.LoadedMeshes and not self.LoadedVertices and not self.LoadedIndices)

    def load_materials(self, path):
        """OBJ_Loader.cpp:743-902."""
        if path[-4:] != ".mtl":
            return False
        if not os.path.isfile(path):
            return False
        temp_material = Material()
        listening = False
        for curline in self._lines(path):
            ft = first_token(curline)
            if ft == "newmtl":
                if not listening:
                    listening = True
                else:
                    self.LoadedMaterials.append(temp_material)
                    temp_material = Material()
                temp_material.name = tail(curline) if len(curline) > 7 else "none"
            skip = False
            for key in ("Ka", "Kd", "Ks"):
                if ft == key:
                    temp = split(tail(curline), " ")
                    if len(temp) != 3:
                        skip = True                   # `continue`: the rest of the line's checks are skipped
                        break
                    setattr(temp_material, key, (stof(temp[0]), stof(temp[1]), stof(temp[2])))
            if skip:
                continue
            if ft == "Ns":
                temp_material.Ns = stof(tail(curline))
            if ft == "Ni":
                temp_material.Ni = stof(tail(curline))
            if ft == "d":
                temp_material.d = stof(tail(curline))
            if ft == "illum":
                temp_material.illum = stoi(tail(curline))
            for key in ("map_Ka", "map_Kd", "map_Ks", "map_Ns", "map_d"):
                if ft == key:
                    setattr(temp_material, key, tail(curline))
            if ft in ("map_Bump", "map_bump", "bump"):
                temp_material.map_bump = tail(curline)
        self.LoadedMaterials.append(temp_material)
        return True
