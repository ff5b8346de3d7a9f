"""Host-side scene front-end (plumbing, not the hot path).

Reads OBJ/MTL/camera files and applies the mesh pre-processing Fermat's RenderingContextImpl::init performs before
uploading (src/renderer.cu:735-744): compress_normals -> compress_tex -> unify_vertex_attributes -> apply_material_flags
(src/mesh/MeshStorage.cpp:246-299, 430-445, 651-840).  The output is the set of plain arrays that MeshView
(src/mesh/MeshView.h:96-145) exposes to the kernels; both the HIP product and the test oracle consume it as *input*.

MTL semantics follow src/mesh/MeshBase.cpp:492-713 and src/mesh/MeshStorage.cpp:150-172
(Ns -> roughness = 1/Ns, Ni -> ior, Ke/e -> emissive, Tr/d -> opacity, Td, r/Kr -> reflectivity, f -> flags,
map_* [-s sx sy] file).  Scene-loader parity with the reference's importers is a "next" row.
"""
from __future__ import annotations

import os
import numpy as np

# MeshMaterial, 208 bytes (src/mesh/MeshView.h:55-74); TextureReference = {u32 texture; pad; float2 scaling}
TEXREF_DTYPE = np.dtype([("texture", "<u4"), ("_pad", "<u4"), ("scaling", "<f4", (2,))])
MATERIAL_DTYPE = np.dtype([
    ("diffuse", "<f4", (4,)), ("diffuse_trans", "<f4", (4,)), ("ambient", "<f4", (4,)), ("specular", "<f4", (4,)),
    ("emissive", "<f4", (4,)), ("reflectivity", "<f4", (4,)),
    ("roughness", "<f4"), ("index_of_refraction", "<f4"), ("opacity", "<f4"), ("flags", "<i4"),
    ("ambient_map", TEXREF_DTYPE), ("diffuse_map", TEXREF_DTYPE), ("diffuse_trans_map", TEXREF_DTYPE),
    ("specular_map", TEXREF_DTYPE), ("emissive_map", TEXREF_DTYPE), ("bump_map", TEXREF_DTYPE)])
assert MATERIAL_DTYPE.itemsize == 208
INVALID_TEXTURE = 0xFFFFFFFF

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def default_material_params():
    """MeshMaterialParams::setToDefaultParams (src/mesh/MeshBase.cpp:354-412)."""
    return dict(name="null-material", diffuse=[0.7, 0.7, 0.7], diffuse_trans=[0.0, 0.0, 0.0], ambient=[0.2, 0.2, 0.2],
                specular=[0.0, 0.0, 0.0], emissive=[0.0, 0.0, 0.0], phong_exponent=0.0, index_of_refraction=1.0,
                opacity=1.0, reflectivity=[0.0, 0.0, 0.0], flags=0, maps={})


def load_mtl(path):
    """MTL reader restating MeshBase::loadMaterials (src/mesh/MeshBase.cpp:492-713): dispatch on the first characters of each token.
    Like the reference the returned list starts with a default-valued staging material ("null-material_0") that also lands
    in the material table; returns [] when the file cannot be opened."""
    if not os.path.exists(path):
        return []
    mats = [default_material_params()]
    mats[0]["name"] = "null-material_0"
    cur = mats[0]
    with open(path, "r", errors="replace") as f:
        for raw in f:
            line = raw.strip()
            if not line or line[0] == "#":
                continue
            tok = line.split()
            key = tok[0]
            if key[0] == "n":
                cur = default_material_params()
                cur["name"] = tok[2] if len(tok) > 2 else (tok[1] if len(tok) > 1 else "")     # sscanf("%s %s", buf2, buf2)
                mats.append(cur)
                continue
            fl = lambda i: float(tok[i])  # noqa: E731
            if key[0] == "N":
                if key[1:2] == "s":
                    cur["phong_exponent"] = fl(1)
                elif key[1:2] == "i":
                    cur["index_of_refraction"] = fl(1)
            elif key[0] == "T":
                if key[1:2] == "r":
                    cur["opacity"] = float(np.float32(1.0) - np.float32(fl(1)))
                elif key[1:2] == "d":
                    cur["diffuse_trans"] = [fl(1), fl(2), fl(3)]
            elif key[0] == "d":
                cur["opacity"] = fl(1)
            elif key[0] == "r":
                cur["reflectivity"] = [fl(1)] * 3
            elif key[0] == "e":
                cur["emissive"] = [fl(1), fl(2), fl(3)]
            elif key[0] == "f":
                cur["flags"] = int(tok[1])
            elif key[0] == "m":
                names = {"map_Ka": "ambient_map", "map_Kd": "diffuse_map", "map_Ks": "specular_map", "map_Ke": "emissive_map",
                         "map_Td": "diffuse_trans_map", "map_Bump": "bump_map", "map_bump": "bump_map"}
                if key in names:
                    scaling = [1.0, 1.0]
                    rest = tok[1:]
                    if rest and rest[0] == "-s":
                        scaling = [float(rest[1]), float(rest[2])]
                        rest = rest[3:]
                    cur["maps"][names[key]] = (rest[0] if rest else "", scaling)
            elif key[0] == "K":
                tgt = {"d": "diffuse", "s": "specular", "a": "ambient", "e": "emissive", "r": "reflectivity"}.get(key[1:2])
                if tgt:
                    cur[tgt] = [fl(1), fl(2), fl(3)]
    return mats


def load_camera(path):
    """-c camera file (src/renderer.cu:508-522): eye / aim / up / fov; dx = normalize(cross(aim-eye, up))."""
    vals = np.array(open(path).read().split(), dtype=np.float32)
    return make_camera(vals[0:3], vals[3:6], vals[6:9], float(vals[9]))


def make_camera(eye, aim, up, fov):
    eye = np.asarray(eye, np.float32); aim = np.asarray(aim, np.float32); up = np.asarray(up, np.float32)
    dx = np.cross(aim - eye, up).astype(np.float32)
    ln = np.float32(np.sqrt(np.float32((dx * dx).sum())))
    if ln > 0:
        dx = dx / ln
    return np.concatenate([eye, aim, up, dx, np.float32([fov])]).astype(np.float32)


class RawMesh:
    """Indexed triangle soup with separate position / normal / texcoord index streams (-1 = missing)."""

    def __init__(self):
        self.positions = np.zeros((0, 3), np.float32)
        self.normals = np.zeros((0, 3), np.float32)
        self.texcoords = np.zeros((0, 2), np.float32)
        self.v_idx = np.zeros((0, 3), np.int32)
        self.n_idx = np.zeros((0, 3), np.int32)
        self.t_idx = np.zeros((0, 3), np.int32)
        self.mat_idx = np.zeros((0,), np.int32)
        self.materials = []        # list of param dicts
        self.base_dir = "."

    def transformed(self, M):
        """transform() of src/mesh/MeshStorage.cpp:623-639 in fp32: points by M (row . (x,y,z,1), left to right), normals by the
        inverse transpose (adjugate / determinant), NOT re-normalised."""
        M = np.asarray(M, np.float32).reshape(4, 4)
        r = RawMesh()
        r.__dict__.update({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in self.__dict__.items()})
        P = self.positions.astype(np.float32)
        out = np.empty_like(P)
        for k in range(3):
            out[:, k] = ((M[k, 0] * P[:, 0] + M[k, 1] * P[:, 1]) + M[k, 2] * P[:, 2]) + M[k, 3] * np.float32(1.0)
        r.positions = out
        if len(self.normals):
            N = _inverse4(M).T.copy()
            V = self.normals.astype(np.float32)
            on = np.empty_like(V)
            for k in range(3):
                on[:, k] = (N[k, 0] * V[:, 0] + N[k, 1] * V[:, 1]) + N[k, 2] * V[:, 2]
            r.normals = on
        r.materials = list(self.materials)
        return r

    def with_per_triangle_normals(self):
        """add_per_triangle_normals (src/mesh/MeshStorage.cpp:449-480)"""
        r = RawMesh(); r.__dict__.update(self.__dict__)
        tri = self.v_idx
        du = (self.positions[tri[:, 0]] - self.positions[tri[:, 2]]).astype(np.float32)
        dv = (self.positions[tri[:, 1]] - self.positions[tri[:, 2]]).astype(np.float32)
        cr = np.stack([du[:, 1] * dv[:, 2] - du[:, 2] * dv[:, 1], du[:, 2] * dv[:, 0] - du[:, 0] * dv[:, 2],
                       du[:, 0] * dv[:, 1] - du[:, 1] * dv[:, 0]], 1).astype(np.float32)
        r.normals = _normalize_rows(cr)
        r.n_idx = np.arange(len(tri), dtype=np.int32)[:, None].repeat(3, 1)
        return r

    def with_per_triangle_texcoords(self):
        """add_per_triangle_texture_coordinates (src/mesh/MeshStorage.cpp:484-511)"""
        r = RawMesh(); r.__dict__.update(self.__dict__)
        r.texcoords = np.float32([[0, 0], [1, 0], [0, 1]])
        r.t_idx = np.tile(np.int32([0, 1, 2]), (len(self.v_idx), 1))
        return r

    @staticmethod
    def merge(meshes):
        """merge() of src/mesh/MeshStorage.cpp:514-620, applied left to right"""
        out = None
        for m in meshes:
            m2 = RawMesh(); m2.__dict__.update(m.__dict__)
            m2.materials = []
            for mat in m.materials:
                mat = dict(mat); mat["_base_dir"] = mat.get("_base_dir", m.base_dir)
                m2.materials.append(mat)
            if out is None:
                out = m2
                continue
            a, b = out, m2
            if (len(a.normals) > 0) != (len(b.normals) > 0):
                if len(a.normals) == 0:
                    a = a.with_per_triangle_normals() if len(a.v_idx) else a
                else:
                    b = b.with_per_triangle_normals()
            if (len(a.texcoords) > 0) != (len(b.texcoords) > 0):
                if len(a.texcoords) == 0:
                    a = a.with_per_triangle_texcoords() if len(a.v_idx) else a
                else:
                    b = b.with_per_triangle_texcoords()
            r = RawMesh()
            vo, no, to, mo = len(a.positions), len(a.normals), len(a.texcoords), len(a.materials)
            r.positions = np.concatenate([a.positions, b.positions]).astype(np.float32)
            r.normals = np.concatenate([a.normals.reshape(-1, 3), b.normals.reshape(-1, 3)]).astype(np.float32)
            r.texcoords = np.concatenate([a.texcoords.reshape(-1, 2), b.texcoords.reshape(-1, 2)]).astype(np.float32)
            r.v_idx = np.concatenate([a.v_idx, b.v_idx + vo]).astype(np.int32)
            # the reference offsets every stored index, -1 ("not provided") included, when the stream exists
            r.n_idx = np.concatenate([a.n_idx.reshape(-1, 3), b.n_idx.reshape(-1, 3) + no]).astype(np.int32)
            r.t_idx = np.concatenate([a.t_idx.reshape(-1, 3), b.t_idx.reshape(-1, 3) + to]).astype(np.int32)
            r.mat_idx = np.concatenate([a.mat_idx, b.mat_idx + mo]).astype(np.int32)
            r.materials = list(a.materials) + list(b.materials)
            r.base_dir = a.base_dir
            out = r
        return out if out is not None else RawMesh()


def _inverse4(M):
    """adjugate / determinant in fp32, same operation order as fermat_amd/csrc/host/scene_io.cpp::invert"""
    M = np.asarray(M, np.float32)
    f = np.float32

    def det3(r, c):
        a = lambda i, j: M[r[i], c[j]]  # noqa: E731
        return f(f(f(a(0, 0) * f(f(a(1, 1) * a(2, 2)) - f(a(1, 2) * a(2, 1)))) - f(a(0, 1) * f(f(a(1, 0) * a(2, 2)) - f(a(1, 2) * a(2, 0)))))
                 + f(a(0, 2) * f(f(a(1, 0) * a(2, 1)) - f(a(1, 1) * a(2, 0)))))
    C = np.zeros((4, 4), np.float32)
    for i in range(4):
        for j in range(4):
            rr = [x for x in range(4) if x != i]; cc = [x for x in range(4) if x != j]
            C[i, j] = f(-1.0 if (i + j) & 1 else 1.0) * det3(rr, cc)
    det = f(f(f(f(M[0, 0] * C[0, 0]) + f(M[0, 1] * C[0, 1])) + f(M[0, 2] * C[0, 2])) + f(M[0, 3] * C[0, 3]))
    if det == 0:
        return np.eye(4, dtype=np.float32)
    return (C.T / det).astype(np.float32)


def load_obj(path):
    """OBJ reader restating MeshBase::loadInfoFromObj/loadDataFromObj (src/mesh/MeshBase.cpp:728-1400): v / vn / vt / f (fan
    triangulation, negative indices), mtllib, usemtl, g.  Triangles are emitted GROUP BY GROUP, groups being "<g>:<usemtl>"
    names held in a std::map, i.e. in lexicographic order (MeshLoader::allocateData), not in file order.  Material 0 is the
    inserted default material (:749-756), material 1 the MTL staging default, then the library's materials."""
    m = RawMesh()
    m.base_dir = os.path.dirname(os.path.abspath(path))
    P, N, T = [], [], []
    m.materials = [default_material_params()]
    by_name = {"null-material": 0}
    material_count = 1
    groups = {}
    base, mat_name, cur = "null-group", "null-material", 0
    g = groups.setdefault(base, ([], [], [], []))
    with open(path, "r", errors="replace") as f:
        for raw in f:
            line = raw.strip()
            if not line or line[0] == "#":
                continue
            tok = line.split()
            k = tok[0]
            if k == "v":
                P.append([float(tok[1]), float(tok[2]), float(tok[3])])
            elif k == "vn":
                N.append([float(tok[1]), float(tok[2]), float(tok[3])])
            elif k == "vt":
                T.append([float(tok[1]), float(tok[2]) if len(tok) > 2 else 0.0])
            elif k[0] == "m":
                lib = tok[2] if len(tok) > 2 else tok[1]
                for mat in load_mtl(os.path.join(m.base_dir, lib)):
                    mat["_base_dir"] = m.base_dir
                    by_name.setdefault(mat["name"], len(m.materials))
                    m.materials.append(mat)
            elif k[0] == "u":
                mat_name = tok[2] if len(tok) > 2 else tok[1]
                if mat_name not in by_name:
                    by_name[mat_name] = material_count
                    material_count += 1
                cur = by_name[mat_name]
                g = groups.setdefault(base + ":" + mat_name, ([], [], [], []))
            elif k[0] == "g":
                base = tok[1] if len(tok) > 1 else ""
                g = groups.setdefault(base + ":" + mat_name, ([], [], [], []))
            elif k == "f":
                corners = []
                for c in tok[1:]:
                    parts = c.split("/")
                    vi = int(parts[0]); vi = vi - 1 if vi >= 0 else len(P) + vi
                    ti = -1; ni = -1
                    if len(parts) > 1 and parts[1]:
                        ti = int(parts[1]); ti = ti - 1 if ti >= 0 else len(T) + ti
                    if len(parts) > 2 and parts[2]:
                        ni = int(parts[2]); ni = ni - 1 if ni >= 0 else len(N) + ni
                    corners.append((vi, ti, ni))
                for i in range(1, len(corners) - 1):
                    a, b, c = corners[0], corners[i], corners[i + 1]
                    g[0].append([a[0], b[0], c[0]]); g[1].append([a[1], b[1], c[1]]); g[2].append([a[2], b[2], c[2]]); g[3].append(cur)
    VI, TI, NI, MI = [], [], [], []
    m.group_names, m.group_offsets = [], []
    for name in sorted(groups):
        gv, gt, gn, gm = groups[name]
        if not gm:
            continue
        m.group_names.append(name); m.group_offsets.append(len(MI))
        VI += gv; TI += gt; NI += gn; MI += gm
    m.group_offsets.append(len(MI))
    while len(m.materials) < material_count:
        m.materials.append(default_material_params())
    m.positions = np.array(P, np.float32).reshape(-1, 3)
    m.normals = np.array(N, np.float32).reshape(-1, 3)
    m.texcoords = np.array(T, np.float32).reshape(-1, 2)
    m.v_idx = np.array(VI, np.int32).reshape(-1, 3); m.n_idx = np.array(NI, np.int32).reshape(-1, 3)
    m.t_idx = np.array(TI, np.int32).reshape(-1, 3); m.mat_idx = np.array(MI, np.int32)
    return m


_PLY_TYPES = {"int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8",
              "char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4", "double": "f8"}
_PLY_INT_RANGE = {"i1": (-128, 127), "u1": (0, 255), "i2": (-32768, 32767), "u2": (0, 65535), "i4": (-2**63, 2**63 - 1), "u4": (0, 2**63 - 1)}


def load_ply(path):
    """PLY reader restating MeshBase::loadFromPly (src/mesh/MeshBase.cpp:191-340,1416-1520) over rply 1.01
    (src/mesh/rply-1.01/rply.c): magic "ply\n"; header words split at blanks; ascii / binary_little_endian / binary_big_endian
    1.0; after `end_header` ONE delimiter byte, then the data; every value passes through a double.  The loader's callbacks are a
    state machine and are reproduced as one: x, y write the current vertex and z advances it (same for nx ny nz and s t / u v);
    the first three entries of every face `vertex_indices` list write the current triangle and the third advances it; normal and
    texcoord indices equal the vertex indices; one group "null-group", one default material."""
    buf = open(path, "rb").read()
    if buf[:4] != b"ply\n":
        raise ValueError("Error opening ply file during first pass (%s)" % path)
    pos = 4
    blanks = b" \n\r\t"

    def word():
        nonlocal pos
        n = len(buf)
        while pos < n and buf[pos] in blanks:
            pos += 1
        if pos >= n:
            return None
        b = pos
        while pos < n and buf[pos] not in blanks:
            pos += 1
        w = buf[b:pos].decode("latin-1")
        if pos < n:
            pos += 1
        return w

    def skip_line():
        nonlocal pos
        e = buf.find(b"\n", pos)
        if e < 0:
            return False
        pos = e + 1
        return True

    def bad_header():
        raise ValueError("Error parsing ply header during first pass (%s)" % path)

    if word() != "format":
        bad_header()
    mode = word()
    if mode not in ("ascii", "binary_little_endian", "binary_big_endian") or word() != "1.0":
        bad_header()
    elements = []            # [name, count, [(prop name, type | ("list", len type, value type))]]
    w = word()
    while w != "end_header":
        if w in ("comment", "obj_info"):
            if not skip_line():
                bad_header()
            w = word()
        elif w == "element":
            name = word(); cnt = word()
            try:
                cnt = int(cnt.split()[0]) if cnt is not None else None
            except ValueError:
                cnt = None
            if name is None or cnt is None:
                bad_header()
            props = []
            w = word()
            while True:
                if w == "property":
                    t = word()
                    if t == "list":
                        lt, vt = word(), word()
                        if lt not in _PLY_TYPES or vt not in _PLY_TYPES:
                            bad_header()
                        t = ("list", _PLY_TYPES[lt], _PLY_TYPES[vt])
                    elif t in _PLY_TYPES:
                        t = _PLY_TYPES[t]
                    else:
                        bad_header()
                    pn = word()
                    if pn is None:
                        bad_header()
                    props.append((pn, t))
                    w = word()
                elif w in ("comment", "obj_info"):
                    if not skip_line():
                        bad_header()
                    w = word()
                else:
                    break
            elements.append([name, cnt, props])
        else:
            bad_header()

    def count_of(el, prop):
        for name, cnt, props in elements:
            if name == el:
                return cnt if any(p[0] == prop for p in props) else 0
        return 0

    nv, nn = count_of("vertex", "x"), count_of("vertex", "nx")
    nt = count_of("vertex", "s") or count_of("vertex", "u")
    ntri = count_of("face", "vertex_indices")
    P = np.zeros((nv, 3), np.float32); N = np.zeros((nn, 3), np.float32); T = np.zeros((nt, 2), np.float32)
    tri = np.zeros((ntri, 3), np.int32)
    vertex_el = next((i for i, e in enumerate(elements) if e[0] == "vertex"), -1)
    face_el = next((i for i, e in enumerate(elements) if e[0] == "face"), -1)
    cur = [0, 0, 0, 0]       # vertex, normal, texcoord, triangle cursors

    def bad_data():
        raise ValueError("Error parsing ply file (%s)" % path)

    endian = "<" if mode != "binary_big_endian" else ">"

    def value(t):
        nonlocal pos
        if mode == "ascii":
            wd = word()
            if wd is None:
                bad_data()
            try:
                if t in ("f4", "f8"):
                    v = float(wd)
                    lim = float(np.finfo(np.float32).max) if t == "f4" else float(np.finfo(np.float64).max)
                    if not (-lim <= v <= lim):
                        bad_data()
                    return v
                v = int(wd, 10)
            except ValueError:
                bad_data()
            lo, hi = _PLY_INT_RANGE[t]
            if not (lo <= v <= hi):
                bad_data()
            return float(v)
        dt = np.dtype(endian + t)
        if pos + dt.itemsize > len(buf):
            bad_data()
        v = float(np.frombuffer(buf, dt, 1, pos)[0])
        pos += dt.itemsize
        return v

    coord_of = {"x": 0, "y": 1, "z": 2}
    if nn:
        coord_of.update({"nx": 3, "ny": 4, "nz": 5})
    if nt:
        coord_of.update({"s": 6, "t": 7, "u": 6, "v": 7})
    for ei, (name, cnt, props) in enumerate(elements):
        for _ in range(cnt):
            for pn, t in props:
                if not isinstance(t, tuple):
                    v = value(t)
                    c = coord_of.get(pn, -1) if ei == vertex_el else -1
                    if c < 0:
                        continue
                    if c <= 2:
                        if cur[0] >= nv: bad_data()
                        P[cur[0], c] = np.float32(v); cur[0] += c == 2
                    elif c <= 5:
                        if cur[1] >= nn: bad_data()
                        N[cur[1], c - 3] = np.float32(v); cur[1] += c == 5
                    else:
                        if cur[2] >= nt: bad_data()
                        T[cur[2], c - 6] = np.float32(v); cur[2] += c == 7
                else:
                    ln = int(value(t[1]))
                    is_face = ei == face_el and pn == "vertex_indices"
                    for l in range(ln):
                        v = value(t[2])
                        if not is_face or l > 2:
                            continue
                        if cur[3] >= ntri: bad_data()
                        tri[cur[3], l] = int(v); cur[3] += l == 2
    m = RawMesh()
    m.base_dir = os.path.dirname(os.path.abspath(path))
    m.materials = [default_material_params()]
    m.positions, m.normals, m.texcoords = P, N, T
    m.v_idx = tri
    m.n_idx = tri.copy() if nn else np.full((ntri, 3), -1, np.int32)
    m.t_idx = tri.copy() if nt else np.full((ntri, 3), -1, np.int32)
    m.mat_idx = np.zeros(ntri, np.int32)
    m.group_names, m.group_offsets = ["null-group"], [0, ntri]
    return m


def load_model(path):
    """MeshBase::loadModel (src/mesh/MeshBase.cpp:446-460): dispatch on the case-sensitive extension"""
    ext = path.rsplit(".", 1)[1] if "." in path else ""
    if ext == "obj":
        return load_obj(path)
    if ext == "ply":
        return load_ply(path)
    raise ValueError("Unrecognized model file extension (%s)" % path)


def load_fa(path, cameras=None, dir_lights=None):
    """.fa scene scripts (src/mesh/fermat_loader.cpp:46-360): Begin/End transform stack, Transform/Translate/Scale/RotateX|Y|Z
    (new * top), LoadScene/LoadMesh (transform, merge, default-material replacement), LoadMaterials, SetMaterial, Camera,
    DirectionalLight.  Returns (RawMesh, cameras, dir_lights)."""
    cameras = [] if cameras is None else cameras
    dir_lights = [] if dir_lights is None else dir_lights
    if not path.endswith(".fa"):
        return load_model(path), cameras, dir_lights
    base_dir = os.path.dirname(os.path.abspath(path))
    text = open(path, "r", errors="replace").read()
    # token stream with line structure: '#' tokens and Camera/DirectionalLight consume the rest of their line
    toks = []
    for ln in text.split("\n"):
        ws = ln.split()
        i = 0
        while i < len(ws):
            if ws[i][0] == "#":
                break
            if ws[i] in ("Camera", "DirectionalLight"):
                toks.append((ws[i], ws[i + 1:])); break
            toks.append((ws[i], None)); i += 1
    f32 = np.float32
    stack = [np.eye(4, dtype=np.float32)]
    mesh = RawMesh(); mesh.base_dir = base_dir
    default_material = -1
    pos = 0

    def mul(a, b):
        r = np.zeros((4, 4), np.float32)
        for i in range(4):
            for j in range(4):
                s = f32(0)
                for k in range(4):
                    s = f32(s + f32(a[i, k] * b[k, j]))
                r[i, j] = s
        return r

    def nxt():
        nonlocal pos
        t = toks[pos][0]; pos += 1
        return t
    while pos < len(toks):
        cmd, rest = toks[pos]; pos += 1
        if cmd == "Begin":
            stack.append(stack[-1].copy())
        elif cmd == "End":
            if len(stack) > 1:
                stack.pop()
        elif cmd == "Transform":
            m = np.array([f32(nxt()) for _ in range(16)], np.float32).reshape(4, 4)
            stack[-1] = mul(m, stack[-1])
        elif cmd == "Translate":
            m = np.eye(4, dtype=np.float32); m[0, 3] = f32(nxt()); m[1, 3] = f32(nxt()); m[2, 3] = f32(nxt())
            stack[-1] = mul(m, stack[-1])
        elif cmd == "Scale":
            m = np.eye(4, dtype=np.float32); m[0, 0] = f32(nxt()); m[1, 1] = f32(nxt()); m[2, 2] = f32(nxt())
            stack[-1] = mul(m, stack[-1])
        elif cmd in ("RotateX", "RotateY", "RotateZ"):
            q = f32(f32(f32(nxt()) * f32(np.pi)) / f32(180.0))
            sn, cs = f32(np.sin(q)), f32(np.cos(q))
            m = np.eye(4, dtype=np.float32)
            if cmd[6] == "X":
                m[1, 1] = m[2, 2] = cs; m[1, 2] = -sn; m[2, 1] = sn
            elif cmd[6] == "Y":
                m[0, 0] = m[2, 2] = cs; m[2, 0] = -sn; m[0, 2] = sn
            else:
                m[0, 0] = m[1, 1] = cs; m[1, 0] = sn; m[0, 1] = -sn
            stack[-1] = mul(m, stack[-1])
        elif cmd in ("LoadScene", "LoadMesh"):
            name = nxt()
            full = name if os.path.exists(name) else os.path.join(base_dir, name)
            if not os.path.exists(full):
                raise FileNotFoundError('unable to find file "%s"' % name)
            other, _, _ = load_fa(full, cameras, dir_lights)
            other = other.transformed(stack[-1])
            tri0, nm = len(mesh.v_idx), len(mesh.materials)
            keep_dir = mesh.base_dir
            mesh = RawMesh.merge([mesh, other])
            mesh.base_dir = keep_dir
            if default_material != -1:
                sel = mesh.mat_idx[tri0:] == nm
                mesh.mat_idx[tri0:][sel] = default_material
        elif cmd == "LoadMaterials":
            name = nxt()
            full = name if os.path.exists(name) else os.path.join(base_dir, name)
            for mat in load_mtl(full):
                mat["_base_dir"] = os.path.dirname(os.path.abspath(full))
                mat["maps"].pop("bump_map", None)
                mesh.materials.append(mat)
        elif cmd == "SetMaterial":
            name = nxt()
            for i in range(len(mesh.materials) - 1, -1, -1):
                if mesh.materials[i]["name"] == name:
                    default_material = i
                    break
        elif cmd == "Camera":
            if not rest or rest[0] != "persp":
                continue
            c = dict(eye=[0, 0, -1], aim=[0, 0, 0], up=[0, 1, 0], fov=float(np.float32(np.pi) / np.float32(4)))
            i = 1
            while i < len(rest):
                if rest[i] in ("eye", "aim", "up"):
                    c[rest[i]] = [float(x) for x in rest[i + 1:i + 4]]; i += 4
                elif rest[i] == "fov":
                    c["fov"] = float(rest[i + 1]); i += 2
                else:
                    break
            cameras.append(make_camera(c["eye"], c["aim"], c["up"], c["fov"]))
        elif cmd == "DirectionalLight":
            d = np.zeros(6, np.float32)
            i = 0
            while rest and i < len(rest):
                if rest[i] in ("dir", "direction"):
                    v = np.float32([float(x) for x in rest[i + 1:i + 4]])
                    d[:3] = _normalize_rows(v[None])[0]; i += 4
                elif rest[i] == "color":
                    d[3:] = [float(x) for x in rest[i + 1:i + 4]]; i += 4
                else:
                    break
            dir_lights.append(d)
    return mesh, cameras, dir_lights


def load_scene(path):
    """the scene half of RenderingContextImpl::init (src/renderer.cu:690-870): load a .fa / .obj scene and pre-process it"""
    raw, cameras, dir_lights = load_fa(path)
    cam = cameras[0] if cameras else make_camera([0, 0, -1], [0, 0, 0], [0, 1, 0], float(np.float32(np.pi) / np.float32(4)))
    return Scene(raw, cam, dir_lights=np.array(dir_lights, np.float32).reshape(-1, 6) if dir_lights else None)


def load_tga(path):
    """Uncompressed / RLE true-colour TGA -> float4 texels (bytes/255, alpha 0) as src/renderer.cu:805-822 builds them.
    Row order is kept as stored, like contrib/cugar/image/tga.cpp's raw read."""
    d = np.fromfile(path, np.uint8)
    idlen, cmap, itype = int(d[0]), int(d[1]), int(d[2])
    w = int(d[12]) | (int(d[13]) << 8); h = int(d[14]) | (int(d[15]) << 8); bpp = int(d[16])
    off = 18 + idlen
    if cmap:
        raise ValueError("colour-mapped TGA unsupported")
    nb = bpp // 8
    if itype == 2:
        px = d[off:off + w * h * nb].reshape(h * w, nb)
    elif itype == 10:
        out = np.zeros((w * h, nb), np.uint8); i = 0; p = off
        while i < w * h:
            c = int(d[p]); p += 1
            n = (c & 0x7f) + 1
            if c & 0x80:
                out[i:i + n] = d[p:p + nb]; p += nb
            else:
                out[i:i + n] = d[p:p + n * nb].reshape(n, nb); p += n * nb
            i += n
        px = out
    else:
        raise ValueError("unsupported TGA type %d" % itype)
    rgb = px[:, [2, 1, 0]].astype(np.float32) / np.float32(255.0)     # BGR -> RGB
    tex = np.zeros((h, w, 4), np.float32); tex[..., :3] = rgb.reshape(h, w, 3)
    return tex


def _pack_normal(n):
    """cugar::pack_normal (contrib/cugar/linalg/vector_inl.h:748-790): 10:10:10 of saturate(n*0.5+0.5)*1023, truncated."""
    n = np.asarray(n, np.float32)
    e = n * np.float32(0.5) + np.float32(0.5)
    e = np.where(np.isnan(e), np.float32(0), np.clip(e, np.float32(0), np.float32(1))).astype(np.float32)
    q = (e * np.float32(1023.0)).astype(np.float32).astype(np.uint32)
    return (q[..., 0] | (q[..., 1] << 10) | (q[..., 2] << 20)).astype(np.uint32)


def _normalize_rows(v):
    v = np.asarray(v, np.float32)
    d = (v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]).astype(np.float32) + v[:, 2] * v[:, 2]
    ln = np.sqrt(d.astype(np.float32)).astype(np.float32)
    out = v.copy()
    nz = ln > 0
    out[nz] = (v[nz] / ln[nz, None]).astype(np.float32)
    return out


class Scene:
    """Pre-processed scene = the arrays behind MeshView + materials + textures + camera + directional lights."""

    def __init__(self, raw: RawMesh, camera, dir_lights=None, textures=None):
        self.camera = np.asarray(camera, np.float32)
        self.dir_lights = np.asarray(dir_lights if dir_lights is not None else np.zeros((0, 6)), np.float32).reshape(-1, 6)
        self.textures = []           # list of (H,W,4) float32
        self._tex_ids = {}
        self._build_materials(raw, textures or {})
        self._preprocess(raw)

    # -- materials: loadModel's translation (src/mesh/MeshStorage.cpp:150-172)
    def _texture_id(self, name, base_dir, provided):
        if not name:
            return INVALID_TEXTURE
        if name in self._tex_ids:
            return self._tex_ids[name]
        tex = None
        if name in provided:
            tex = np.ascontiguousarray(provided[name], np.float32)
        else:
            p = os.path.join(base_dir, name.replace("\\", "/"))
            if os.path.exists(p) and p.lower().endswith(".tga"):
                tex = load_tga(p)
        tid = len(self.textures)
        self.textures.append(tex)      # None => n_levels == 0 => lookups return the default value
        self._tex_ids[name] = tid
        return tid

    def _build_materials(self, raw, provided):
        mats = np.zeros(len(raw.materials), MATERIAL_DTYPE)
        for i, p in enumerate(raw.materials):
            m = mats[i]
            for k in ("diffuse", "diffuse_trans", "ambient", "specular", "emissive", "reflectivity"):
                m[k][:3] = np.float32(p[k]); m[k][3] = 0.0
            pe = np.float32(p["phong_exponent"])
            m["roughness"] = np.float32(1.0) / pe if pe != 0 else np.float32(1.0)
            m["index_of_refraction"] = p["index_of_refraction"]; m["opacity"] = p["opacity"]; m["flags"] = p["flags"]
            for k in ("ambient_map", "diffuse_map", "diffuse_trans_map", "specular_map", "emissive_map", "bump_map"):
                name, scaling = p["maps"].get(k, ("", [1.0, 1.0]))
                m[k]["texture"] = self._texture_id(name, p.get("_base_dir", raw.base_dir), provided) if k != "bump_map" or name else INVALID_TEXTURE
                m[k]["scaling"] = scaling
        self.materials = mats

    def _preprocess(self, raw):
        nt = len(raw.v_idx)
        # compress_tex (src/mesh/MeshStorage.cpp:268-299, src/mesh/MeshCompression.h:36-48)
        if len(raw.texcoords):
            # Bbox2f::insert keeps the first-seen value on ties (matters for the sign of zero)
            tc = raw.texcoords
            tmin = np.float32([tc[np.argmax(tc[:, c] == tc[:, c].min()), c] for c in range(2)])
            tmax = np.float32([tc[np.argmax(tc[:, c] == tc[:, c].max()), c] for c in range(2)])
            self.tex_bias = tmin; self.tex_scale = (tmax - tmin).astype(np.float32)
            comp = np.full((nt, 4), -1, np.int32)
            ti = raw.t_idx
            t = raw.texcoords[np.maximum(ti, 0)]                     # (nt,3,2)
            with np.errstate(divide="ignore", invalid="ignore"):
                tn = ((t - self.tex_bias) / self.tex_scale).astype(np.float32)
            tn[..., self.tex_scale == 0] = 0.0          # a degenerate axis (all u or all v equal) would be 0/0
            h = tn.astype(np.float16).view(np.uint16).astype(np.uint32)
            packed = (h[..., 0] | (h[..., 1] << 16)).astype(np.uint32).view(np.int32)
            comp[:, :3] = np.where(ti >= 0, packed, -1)
            self.texture_indices_comp = comp
        else:
            self.tex_bias = np.zeros(2, np.float32); self.tex_scale = np.ones(2, np.float32)
            self.texture_indices_comp = None
        # compress_normals + unify_vertex_attributes (src/mesh/MeshStorage.cpp:246-266, 651-840)
        tri_ids = np.arange(nt, dtype=np.int64)[:, None].repeat(3, 1)
        n_key = np.where(raw.n_idx >= 0, raw.n_idx.astype(np.int64), -tri_ids - 1)
        keys = np.stack([raw.v_idx.astype(np.int64).ravel(), n_key.ravel(), raw.t_idx.astype(np.int64).ravel()], 1)
        uniq, first, inv = np.unique(keys, axis=0, return_index=True, return_inverse=True)
        order = np.argsort(first, kind="stable")                     # first-seen order, as the std::map walk assigns ids
        rank = np.empty_like(order); rank[order] = np.arange(len(order))
        new_idx = rank[inv.ravel()].reshape(nt, 3).astype(np.int32)
        uk = uniq[order]
        nv = len(uk)
        vdata = np.zeros((nv, 4), np.float32)
        vdata[:, :3] = raw.positions[uk[:, 0]]
        normals = np.zeros((nv, 3), np.float32)
        has_n = uk[:, 1] >= 0
        if has_n.any():
            normals[has_n] = raw.normals[uk[has_n, 1]]
        if (~has_n).any():
            t = (-uk[~has_n, 1] - 1)
            tri = raw.v_idx[t]
            vp0, vp1, vp2 = raw.positions[tri[:, 0]], raw.positions[tri[:, 1]], raw.positions[tri[:, 2]]
            du = (vp0 - vp2).astype(np.float32); dv = (vp1 - vp2).astype(np.float32)
            cr = np.stack([du[:, 1] * dv[:, 2] - du[:, 2] * dv[:, 1], du[:, 2] * dv[:, 0] - du[:, 0] * dv[:, 2],
                           du[:, 0] * dv[:, 1] - du[:, 1] * dv[:, 0]], 1).astype(np.float32)
            normals[~has_n] = _normalize_rows(cr)
        vdata[:, 3] = _pack_normal(normals).view(np.float32)
        # MeshStorage::m_texture_data after unify: the raw texture coordinate of each unified vertex (0 where not provided)
        if len(raw.texcoords):
            td = np.zeros((nv, 2), np.float32)
            has_t = uk[:, 2] >= 0
            td[has_t] = raw.texcoords[uk[has_t, 2]]
            self.texture_data = np.ascontiguousarray(td)
        else:
            self.texture_data = None
        self.vertex_data = np.ascontiguousarray(vdata)
        # apply_material_flags (src/mesh/MeshStorage.cpp:430-445)
        vi = np.zeros((nt, 4), np.int32)
        vi[:, :3] = new_idx
        vi[:, 3] = self.materials["flags"][raw.mat_idx]
        self.vertex_indices = np.ascontiguousarray(vi)
        self.material_indices = np.ascontiguousarray(raw.mat_idx.astype(np.int32))
        self.num_triangles = nt
        self.num_vertices = nv
        self.bbox = (self.vertex_data[:, :3].min(0), self.vertex_data[:, :3].max(0))


def cornell_box(name="CornellBox-JP", camera_file="camera-frontal.txt"):
    d = os.path.join(DATA_DIR, "scenes", "CornellBox")
    raw = load_obj(os.path.join(d, name + ".obj"))
    return Scene(raw, load_camera(os.path.join(d, camera_file)))


def _affine(scale=(1, 1, 1), rot_y_deg=0.0, translate=(0, 0, 0)):
    a = np.deg2rad(rot_y_deg)
    R = np.array([[np.cos(a), 0, np.sin(a), 0], [0, 1, 0, 0], [-np.sin(a), 0, np.cos(a), 0], [0, 0, 0, 1]], np.float64)
    S = np.diag([scale[0], scale[1], scale[2], 1.0]); T = np.eye(4); T[:3, 3] = translate
    return T @ R @ S


def _checker(res=64, a=(0.9, 0.9, 0.9), b=(0.2, 0.3, 0.6), cells=8):
    y, x = np.mgrid[0:res, 0:res]
    m = (((x * cells) // res + (y * cells) // res) & 1).astype(bool)
    t = np.zeros((res, res, 4), np.float32)
    t[..., :3] = np.where(m[..., None], np.float32(a), np.float32(b))
    return t


def _uv_sphere(n_lat, n_lon, radius=1.0):
    """Procedural tessellated sphere with smooth normals and texcoords (a stand-in for bathroom2's dense props)."""
    m = RawMesh()
    th = np.linspace(0, np.pi, n_lat + 1); ph = np.linspace(0, 2 * np.pi, n_lon + 1)
    T, P = np.meshgrid(th, ph, indexing="ij")
    n = np.stack([np.sin(T) * np.cos(P), np.cos(T), np.sin(T) * np.sin(P)], -1).reshape(-1, 3)
    m.positions = (n * radius).astype(np.float32); m.normals = n.astype(np.float32)
    m.texcoords = np.stack([P / (2 * np.pi), T / np.pi], -1).reshape(-1, 2).astype(np.float32)
    i, j = np.meshgrid(np.arange(n_lat), np.arange(n_lon), indexing="ij")
    a = (i * (n_lon + 1) + j).ravel(); b = a + 1; c = a + (n_lon + 1); d = c + 1
    tris = np.concatenate([np.stack([a, c, b], 1), np.stack([b, c, d], 1)]).astype(np.int32)
    m.v_idx = tris; m.n_idx = tris.copy(); m.t_idx = tris.copy()
    m.mat_idx = np.ones(len(tris), np.int32)
    return m


def bathroom_standin(detail=1.0):
    """Stand-in for the MISSING models/bathroom2/bathroom.obj (BASELINE configs 3-4; .MISSING_LARGE_BLOBS).

    Geometry: a Cornell-box room scaled to bathroom2's extent seen from bathroom2's own camera
    (models/bathroom2/bathroom.fa:3), a shelf of instanced CornellBox-Glossy boxes laid out like the reference's own
    stand-in script models/bathroom2/bathroom_cornell.fa, and procedurally tessellated, textured glossy spheres that
    bring the triangle count to the ~1M range of a production interior.  Never silently used: callers name it.
    """
    d = os.path.join(DATA_DIR, "scenes", "CornellBox")
    room = load_obj(os.path.join(d, "CornellBox-JP.obj"))
    glossy = load_obj(os.path.join(d, "CornellBox-Glossy.obj"))
    parts = [room.transformed(_affine(scale=(20, 15, 22), translate=(-2, 0, 8)))]
    rng = np.random.default_rng(7)
    # instanced glossy boxes in rows, a la bathroom_cornell.fa
    for row in range(4):
        for k in range(8):
            parts.append(glossy.transformed(_affine(scale=(1.1, 1.1, 1.1), rot_y_deg=-52 + 7 * k,
                                                    translate=(-16 + 4.2 * k, 0.05 + 3.1 * row, -8 + 1.5 * row))))
    # tessellated textured spheres
    n_lat = max(8, int(180 * detail)); n_lon = 2 * n_lat
    tex = {"standin_checker.tga": _checker(256), "standin_stripes.tga": _checker(128, (0.8, 0.5, 0.2), (0.1, 0.1, 0.1), 16)}
    for s in range(6):
        sph = _uv_sphere(n_lat, n_lon, radius=1.6 + 0.5 * (s % 3))
        mat = default_material_params()
        mat.update(name="standin_sphere_%d" % s, diffuse=[0.8, 0.8, 0.8], specular=[0.6, 0.6, 0.6],
                   phong_exponent=float([8, 30, 100][s % 3]), index_of_refraction=1.5)
        mat["maps"] = {"diffuse_map": (list(tex)[s % 2], [4.0, 2.0])}
        if s == 5:   # one glass-like transmissive sphere exercises the transmission lobes
            mat.update(opacity=0.2, diffuse=[0.1, 0.1, 0.1], specular=[0.9, 0.9, 0.9], phong_exponent=200.0, maps={})
        sph.materials = [default_material_params(), mat]
        parts.append(sph.transformed(_affine(translate=(-12 + 4.5 * s, 2.5 + 0.4 * (s % 2), 10 + 2.0 * (s % 3)))))
    raw = RawMesh.merge(parts)
    cam = make_camera([-2.520284, 15.735250 * 0.6, 32.335594 * 0.8], [-1.976656, 14.700628 * 0.45, -2.417851], [0, 1, 0], 1.768946)
    return Scene(raw, cam, textures=tex)


def load_scene_native(path):
    """The same pre-processed Scene as load_scene(), produced by the C++ front-end of the product library
    (fermat_amd/csrc/host/scene_io.cpp through include/fermat_host.h: fpt_host_scene_load / fpt_host_scene_arrays) instead of this
    module's Python twin -- two orders of magnitude faster on multi-million-triangle .fa scenes; host code only, no GPU needed.
    tests/test_scene_io.py checks that the two produce identical arrays."""
    import ctypes as C
    from . import api

    class _SceneArrays(C.Structure):
        _fields_ = [("mesh", api.MeshView), ("textures", C.c_void_p), ("num_textures", C.c_uint32), ("dir_lights", C.c_void_p),
                    ("dir_lights_count", C.c_uint32), ("glossy_reflectance", C.c_void_p), ("camera", api.Camera), ("samples_dir", C.c_char_p)]

    L = api.lib()
    L.fpt_host_scene_load.restype = C.c_void_p; L.fpt_host_scene_load.argtypes = [C.c_char_p, C.c_char_p]
    L.fpt_host_scene_last_error.restype = C.c_char_p
    L.fpt_host_scene_free.argtypes = [C.c_void_p]
    L.fpt_host_scene_arrays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    h = L.fpt_host_scene_load(os.path.abspath(path).encode(), DATA_DIR.encode())
    if not h:
        raise RuntimeError("fpt_host_scene_load(%s): %s" % (path, L.fpt_host_scene_last_error().decode()))

    def arr(ptr, dtype, n):
        if not ptr or n == 0:
            return np.zeros(0, dtype)
        return np.frombuffer((C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr), dtype=dtype).copy()
    try:
        sa = _SceneArrays()
        if L.fpt_host_scene_arrays(h, None, C.byref(sa)) != 0:
            raise RuntimeError("fpt_host_scene_arrays failed")
        m = sa.mesh
        s = Scene.__new__(Scene)
        s.num_triangles, s.num_vertices = int(m.num_triangles), int(m.num_vertices)
        s.vertex_indices = arr(m.vertex_indices, np.int32, s.num_triangles * 4).reshape(-1, 4)
        s.vertex_data = arr(m.vertex_data, np.float32, s.num_vertices * 4).reshape(-1, 4)
        s.texture_indices_comp = arr(m.texture_indices_comp, np.int32, s.num_triangles * 4).reshape(-1, 4) if m.texture_indices_comp else None
        s.texture_data = arr(m.texture_data, np.float32, s.num_vertices * 2).reshape(-1, 2) if m.texture_data else None
        s.material_indices = arr(m.material_indices, np.int32, s.num_triangles)
        s.materials = arr(m.materials, MATERIAL_DTYPE, int(m.num_materials))
        s.tex_bias = np.float32(list(m.tex_bias)); s.tex_scale = np.float32(list(m.tex_scale))
        s.camera = np.frombuffer(bytes(sa.camera), np.float32).copy()
        s.dir_lights = arr(sa.dir_lights, np.float32, sa.dir_lights_count * 6).reshape(-1, 6)
        s.textures = []; s._tex_ids = {}
        tv = arr(sa.textures, np.dtype([("texels", "<u8"), ("res_x", "<u4"), ("res_y", "<u4")]), sa.num_textures)
        for t in tv:
            s.textures.append(arr(int(t["texels"]), np.float32, int(t["res_x"]) * int(t["res_y"]) * 4).reshape(int(t["res_y"]), int(t["res_x"]), 4)
                              if t["texels"] else None)
        s.bbox = (s.vertex_data[:, :3].min(0), s.vertex_data[:, :3].max(0))
        return s
    finally:
        L.fpt_host_scene_free(h)


def bathroom2_standin():
    """The stand-in for BASELINE configs 3-4's scene since round 4 (tools/gen_bathroom2_standin.py): the reference's OWN models/bathroom2/bathroom.mtl
    (23 materials, Kd / Ks maps, mirror, emitters), its textures and the camera of bathroom.fa, on procedural bathroom geometry -- bathroom.obj is absent from
    the reference checkout -- instanced through a .fa script: 493 objects, 1.8 M triangles.  Loaded by the C++ scene front-end."""
    return load_scene_native(os.path.join(DATA_DIR, "scenes", "bathroom2_standin", "bathroom2_standin.fa"))


def water_caustic_standin():
    """The stand-in for BASELINE configs[4]'s scene (tools/gen_water_caustic_standin.py): the reference's OWN models/water_caustic/water_caustic.mtl (Green / Red /
    Silver / Water / White / Light / Light2) and the camera of water_caustic.fa on procedural geometry -- water_caustic.obj is absent from the reference
    checkout: a closed Cornell-style room with a pool whose surface is a wavy height field in `Water` (Ns 1024, d 0: nearly specular, fully transmissive), Silver
    objects and two small, very bright quads in `Light` / `Light2`; 49 objects instanced through a .fa script, 864 576 triangles (819 200 of them the water surface).  Loaded by the C++ scene front-end."""
    return load_scene_native(os.path.join(DATA_DIR, "scenes", "water_caustic_standin", "water_caustic_standin.fa"))


def testball_room():
    """The harder stand-in for BASELINE configs 3-4 (tools/gen_testball_room.py): the bathroom2-sized room filled with ~225 instanced
    material-testball meshes, twelve textured / glossy / coated / transmissive materials, 4.9 M triangles.  Loaded from its .fa script by
    the C++ scene front-end."""
    return load_scene_native(os.path.join(DATA_DIR, "scenes", "testball_room", "testball_room.fa"))
