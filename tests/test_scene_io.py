"""CPU tests of the C++ scene front-end (fermat_amd/csrc/host/scene_io.cpp) against the independent Python
restatement in fermat_amd/scene.py: both follow src/mesh/{MeshBase,MeshStorage,fermat_loader}.cpp and must produce
byte-identical MeshView arrays (triangle order = the reference's group order, material table with the staging default,
fp32 transforms, fp16 texcoords, 10:10:10 normals)."""
import ctypes as C
import os

import numpy as np
import pytest

import fermat_amd as fa
from fermat_amd import api, scene

CORNELL = os.path.join(scene.DATA_DIR, "scenes", "CornellBox")


class SceneArrays(C.Structure):
    _fields_ = [("mesh", api.MeshView), ("textures", C.c_void_p), ("num_textures", C.c_uint32),
                ("dir_lights", C.c_void_p), ("dir_lights_count", C.c_uint32), ("glossy_reflectance", C.c_void_p),
                ("camera", api.Camera), ("samples_dir", C.c_char_p)]


def _lib():
    L = fa.lib()
    L.fpt_host_scene_load.restype = C.c_void_p
    L.fpt_host_scene_load.argtypes = [C.c_char_p, C.c_char_p]
    L.fpt_host_scene_last_error.restype = C.c_char_p
    L.fpt_host_scene_free.argtypes = [C.c_void_p]
    L.fpt_host_scene_arrays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.fpt_host_scene_counts.argtypes = [C.c_void_p, C.c_void_p]
    for f in ("fpt_host_scene_texture_name", "fpt_host_scene_material_name"):
        getattr(L, f).restype = C.c_char_p; getattr(L, f).argtypes = [C.c_void_p, C.c_uint32]
    L.fpt_host_scene_group_name.restype = C.c_char_p
    L.fpt_host_scene_group_name.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    return L


def _arr(ptr, dtype, n):
    if not ptr or n == 0:
        return np.zeros(0, dtype)
    return np.frombuffer((C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr), dtype=dtype).copy()


def load_cpp(path):
    L = _lib()
    h = L.fpt_host_scene_load(path.encode(), scene.DATA_DIR.encode())
    assert h, L.fpt_host_scene_last_error()
    sa = SceneArrays()
    assert L.fpt_host_scene_arrays(h, None, C.byref(sa)) == 0
    m = sa.mesh
    out = dict(
        nt=m.num_triangles, nv=m.num_vertices, nm=m.num_materials,
        vertex_indices=_arr(m.vertex_indices, np.int32, m.num_triangles * 4).reshape(-1, 4),
        vertex_data=_arr(m.vertex_data, np.uint32, m.num_vertices * 4).reshape(-1, 4),
        tex_comp=_arr(m.texture_indices_comp, np.int32, m.num_triangles * 4).reshape(-1, 4) if m.texture_indices_comp else None,
        material_indices=_arr(m.material_indices, np.int32, m.num_triangles),
        materials=_arr(m.materials, scene.MATERIAL_DTYPE, m.num_materials),
        tex_bias=np.float32(list(m.tex_bias)), tex_scale=np.float32(list(m.tex_scale)),
        camera=np.frombuffer(bytes(sa.camera), np.float32).copy(),
        dir_lights=_arr(sa.dir_lights, np.float32, sa.dir_lights_count * 6).reshape(-1, 6),
        glossy=_arr(sa.glossy_reflectance, np.float32, 32 ** 4))
    counts = (C.c_uint32 * 4)()
    L.fpt_host_scene_counts(h, counts)
    out["counts"] = list(counts)
    out["texture_names"] = [L.fpt_host_scene_texture_name(h, i).decode() for i in range(counts[2])]
    out["material_names"] = [L.fpt_host_scene_material_name(h, i).decode() for i in range(m.num_materials)]
    a, b = C.c_int32(), C.c_int32()
    out["groups"] = []
    for i in range(counts[3]):
        n = L.fpt_host_scene_group_name(h, i, C.byref(a), C.byref(b)).decode()
        out["groups"].append((n, a.value, b.value))
    tex = []
    tv = _arr(sa.textures, np.dtype([("texels", "<u8"), ("res_x", "<u4"), ("res_y", "<u4")]), sa.num_textures)
    for t in tv:
        tex.append(_arr(int(t["texels"]), np.float32, int(t["res_x"]) * int(t["res_y"]) * 4).reshape(int(t["res_y"]), int(t["res_x"]), 4) if t["texels"] else None)
    out["textures"] = tex
    L.fpt_host_scene_free(h)
    return out


def same_mesh(cpp, py):
    assert cpp["nt"] == py.num_triangles and cpp["nv"] == py.num_vertices and cpp["nm"] == len(py.materials)
    assert np.array_equal(cpp["vertex_indices"], py.vertex_indices)
    assert np.array_equal(cpp["vertex_data"], py.vertex_data.view(np.uint32))
    assert np.array_equal(cpp["material_indices"], py.material_indices)
    assert cpp["materials"].tobytes() == py.materials.tobytes()
    if py.texture_indices_comp is None:
        assert cpp["tex_comp"] is None
    else:
        assert np.array_equal(cpp["tex_comp"], py.texture_indices_comp)
        assert np.array_equal(cpp["tex_bias"], py.tex_bias) and np.array_equal(cpp["tex_scale"], py.tex_scale)


@pytest.mark.parametrize("name", ["CornellBox-JP", "CornellBox-Glossy"])
def test_obj_loader_matches_python(name):
    cpp = load_cpp(os.path.join(CORNELL, name + ".obj"))
    py = scene.Scene(scene.load_obj(os.path.join(CORNELL, name + ".obj")), scene.make_camera([0, 0, -1], [0, 0, 0], [0, 1, 0], 0.7853981852531433))
    same_mesh(cpp, py)
    # reference conventions: material 0 = inserted default, 1 = MTL staging default; groups "<g>:<usemtl>" in std::map order
    assert cpp["material_names"][0] == "null-material" and cpp["material_names"][1] == "null-material_0"
    names = [g[0] for g in cpp["groups"]]
    assert names == sorted(names) and all(":" in n for n in names)
    assert cpp["groups"][0][1] == 0 and cpp["groups"][-1][2] == cpp["nt"]
    assert all(cpp["groups"][i][2] == cpp["groups"][i + 1][1] for i in range(len(names) - 1))
    assert cpp["glossy"].shape == (32 ** 4,) and np.isfinite(cpp["glossy"]).all()


def test_triangle_order_is_group_order():
    """CornellBox-JP declares floor, ceiling, backWall, ... : the loader must emit backWall first (lexicographic group order)"""
    cpp = load_cpp(os.path.join(CORNELL, "CornellBox-JP.obj"))
    assert cpp["groups"][0][0] == "backWall:backWall"
    mats = [cpp["material_names"][i] for i in cpp["material_indices"]]
    assert mats[0] == "backWall" and "light" in mats


def _write_tga(path, w, h, rng, rle=False, bpp=24):
    px = rng.integers(0, 256, (h, w, bpp // 8), dtype=np.uint8)
    hdr = bytearray(18); hdr[2] = 10 if rle else 2; hdr[12] = w & 255; hdr[13] = w >> 8; hdr[14] = h & 255; hdr[15] = h >> 8; hdr[16] = bpp
    body = bytearray()
    if rle:
        flat = px.reshape(-1, bpp // 8)
        i = 0
        while i < len(flat):
            n = min(int(rng.integers(1, 9)), len(flat) - i)
            if rng.integers(0, 2):
                body.append(0x80 | (n - 1)); body += flat[i].tobytes(); flat[i:i + n] = flat[i]
            else:
                body.append(n - 1); body += flat[i:i + n].tobytes()
            i += n
        px = flat.reshape(h, w, bpp // 8)
    else:
        body += px.tobytes()
    open(path, "wb").write(bytes(hdr) + bytes(body))
    return px


FA = """# test scene
Camera persp eye 0.1 1.0 3.5 aim 0 1 0 up 0 1 0 fov 0.9
DirectionalLight direction 1.0 -0.5 1.0 color 8 8 7
LoadScene {cornell}/CornellBox-JP.obj
Begin
	RotateY -15
	Scale 0.5 0.75 0.5
	Translate 0.2 0.1 -0.3
	LoadScene {cornell}/CornellBox-Glossy.obj
	Begin
		RotateX 30
		RotateZ 12.5
		Transform 1 0 0 0.1  0 1 0.2 0  0 0 1 0  0 0 0 1
		LoadMesh quad.obj
	End
End
LoadMaterials extra.mtl
SetMaterial shiny
Begin
	Translate 0 2 0
	LoadScene plain.obj
End
"""


def _make_fa(tmp_path):
    rng = np.random.default_rng(3)
    _write_tga(str(tmp_path / "checker.tga"), 8, 4, rng)
    _write_tga(str(tmp_path / "glow.tga"), 5, 7, rng, rle=True, bpp=32)
    (tmp_path / "quad.mtl").write_text("newmtl tex\nKd 0.5 0.6 0.7\nNs 20\nNi 1.5\nmap_Kd -s 2 3 checker.tga\nKe 1 2 3\nmap_Ke glow.tga\nf 2\n"
                                       "newmtl glass\nTr 0.25\nTd 0.1 0.2 0.3\nr 0.4\nKs 0.3 0.3 0.3\nmap_Ks missing.tga\n")
    (tmp_path / "quad.obj").write_text("mtllib quad.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 0.5 1\nvn 0 0 1\nvn 0 1 0\nvt 0 0\nvt 2 0\nvt 2 -1\nvt 0 1.5\n"
                                       "g q\nusemtl tex\nf 1/1/1 2/2/1 3/3/1 4/4/1\nusemtl glass\nf 1//2 2//2 5//2\nf -1/1 -2/2 -3/3\n")
    (tmp_path / "extra.mtl").write_text("newmtl shiny\nKd 0.1 0.1 0.1\nKs 0.8 0.8 0.8\nNs 200\nNi 2.0\nmap_Bump bump.tga\n")
    (tmp_path / "plain.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 0 1\nv 1 0 1\nf 1 2 4 3\n")
    (tmp_path / "scene.fa").write_text(FA.format(cornell=CORNELL))
    return str(tmp_path / "scene.fa")


def test_fa_loader_matches_python(tmp_path):
    path = _make_fa(tmp_path)
    cpp = load_cpp(path)
    py = scene.load_scene(path)
    same_mesh(cpp, py)
    assert np.array_equal(cpp["camera"].view(np.uint32), py.camera.view(np.uint32))
    assert np.array_equal(cpp["dir_lights"].view(np.uint32), py.dir_lights.view(np.uint32)) and len(py.dir_lights) == 1
    # SetMaterial: plain.obj's default material (its index 0) is replaced by "shiny"
    shiny = cpp["material_names"].index("shiny")
    assert (cpp["material_indices"][-2:] == shiny).all()
    # textures: first-use order over the merged material table; missing files keep their slot with no levels
    assert cpp["texture_names"] == ["checker.tga", "glow.tga", "missing.tga"]
    assert cpp["textures"][2] is None and py.textures[2] is None
    for a, b in zip(cpp["textures"][:2], py.textures[:2]):
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    tex = cpp["materials"][cpp["material_names"].index("tex")]
    assert tex["flags"] == 2 and tex["diffuse_map"]["texture"] == 0 and tuple(tex["diffuse_map"]["scaling"]) == (2.0, 3.0)
    assert np.float32(tex["roughness"]) == np.float32(1.0) / np.float32(20.0)
    glass = cpp["materials"][cpp["material_names"].index("glass")]
    assert np.float32(glass["opacity"]) == np.float32(0.75) and tuple(glass["reflectivity"][:3]) == (np.float32(0.4),) * 3
    # material flags land in vertex_indices.w (shadow mask)
    assert (cpp["vertex_indices"][:, 3] == cpp["materials"]["flags"][cpp["material_indices"]]).all()


def test_tga_roundtrip_and_camera_file(tmp_path):
    L = _lib()
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (6, 9, 4), dtype=np.uint8)
    p = str(tmp_path / "out.tga").encode()
    assert L.fpt_host_write_tga(p, 9, 6, img.ctypes.data_as(C.c_void_p), 4) == 0
    back = scene.load_tga(p.decode())
    assert np.array_equal((back[..., :3] * 255.0 + 0.5).astype(np.uint8), img[..., :3])
    cam = api.Camera()
    assert L.fpt_host_load_camera(os.path.join(CORNELL, "camera-frontal.txt").encode(), C.byref(cam)) == 0
    ref = scene.load_camera(os.path.join(CORNELL, "camera-frontal.txt"))
    assert np.array_equal(np.frombuffer(bytes(cam), np.float32).view(np.uint32), ref.view(np.uint32))


def test_loader_errors():
    L = _lib()
    assert not L.fpt_host_scene_load(b"/nonexistent/scene.fa", scene.DATA_DIR.encode())
    assert b"unable to open" in L.fpt_host_scene_last_error()
    assert not L.fpt_host_scene_load(os.path.join(CORNELL, "CornellBox-JP.obj").encode(), b"/nonexistent")
    assert b"glossy_reflectance" in L.fpt_host_scene_last_error()


def test_malformed_obj_and_degenerate_texcoords(tmp_path):
    """face indices are range-checked before anything indexes with them; a texcoord axis of zero extent gives 0, not NaN"""
    L = _lib()
    d = str(tmp_path)
    base = "v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0.5 0.1\nvt 0.5 0.9\nvt 0.5 0.4\nvn 0 0 1\n"
    for name, face in (("too_large", "f 1/1/1 2/2/1 9/3/1\n"), ("tex_oob", "f 1/1/1 2/7/1 3/3/1\n"), ("normal_oob", "f 1/1/1 2/2/1 3/3/5\n"),
                       ("too_negative", "f -1/-1/-1 -2/-2/-1 -7/-3/-1\n"), ("zero", "f 0/1/1 2/2/1 3/3/1\n")):
        p = os.path.join(d, name + ".obj")
        open(p, "w").write(base + face)
        assert not L.fpt_host_scene_load(p.encode(), scene.DATA_DIR.encode()), name
        assert b"out of range" in L.fpt_host_scene_last_error(), (name, L.fpt_host_scene_last_error())
    # every u equal: tex_scale.x == 0; the compressed coordinates are finite halfs, identical in the C++ front-end and its Python twin
    p = os.path.join(d, "flat_u.obj")
    open(p, "w").write(base + "f 1/1/1 2/2/1 3/3/1\n")
    cpp = load_cpp(p)
    py = scene.load_scene(p)
    assert cpp["tex_scale"][0] == 0.0 and np.array_equal(cpp["tex_comp"], py.texture_indices_comp)
    h = (cpp["tex_comp"][:, :3].view(np.uint32) & 0xFFFF).astype(np.uint16).view(np.float16)
    assert np.isfinite(h.astype(np.float32)).all() and (h == 0).all()


def test_tga_palette_and_truncation(tmp_path):
    """colour-mapped TGAs honour the first-entry index and reject indices outside the stored palette; a truncated RLE stream
    is an error, not an image with uninitialised rows (checked through a material's map_Kd: a rejected texture has no levels)"""
    L = _lib()
    d = str(tmp_path)

    def scene_with(tga_bytes):
        open(os.path.join(d, "t.tga"), "wb").write(tga_bytes)
        open(os.path.join(d, "m.mtl"), "w").write("newmtl a\nKd 1 1 1\nmap_Kd t.tga\n")
        open(os.path.join(d, "s.obj"), "w").write("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nusemtl a\nf 1/1 2/2 3/3\n")
        return load_cpp(os.path.join(d, "s.obj"))["textures"]

    def hdr(itype, w, h, bpp, cmap=0, start=0, length=0, bits=0):
        return bytes([0, cmap, itype, start & 255, start >> 8, length & 255, length >> 8, bits, 0, 0, 0, 0, w & 255, w >> 8, h & 255, h >> 8, bpp, 0])
    pal = bytes([10, 20, 30, 40, 50, 60])                       # two BGR entries
    ok = scene_with(hdr(1, 2, 1, 8, 1, 4, 2, 24) + pal + bytes([4, 5]))       # indices 4, 5 -> entries 0, 1 (first-entry index 4)
    assert ok[0] is not None and np.allclose(ok[0][0, :, :3] * 255.0, [[30, 20, 10], [60, 50, 40]])
    bad = scene_with(hdr(1, 2, 1, 8, 1, 0, 2, 24) + pal + bytes([0, 2]))      # index 2 is outside a 2-entry palette
    assert bad[0] is None
    rle_ok = scene_with(hdr(10, 4, 1, 24) + bytes([0x83, 1, 2, 3]))           # one run of four pixels
    assert rle_ok[0] is not None and np.allclose(rle_ok[0][0, :, :3] * 255.0, [[3, 2, 1]] * 4)
    rle_cut = scene_with(hdr(10, 4, 2, 24) + bytes([0x83, 1, 2, 3]))          # second row missing
    assert rle_cut[0] is None


def test_cornell_box_jp_facts_written_out_by_hand():
    """Facts of models/CornellBox/CornellBox-JP.{obj,mtl} (shipped verbatim under fermat_amd/data/scenes/CornellBox) as the REFERENCE's loader must
    see them, written here as literals read off the two text files and the reference's rules -- not computed by fermat_amd/scene.py, the Python twin
    the other tests of this file compare the C++ front-end with:
      * 18 quads -> 36 triangles (fan triangulation, src/mesh/MeshBase.cpp:845-915);
      * materials are numbered in the order the .mtl file defines them (loadMaterials runs at `mtllib`, before any `usemtl`, :802-812): leftWall,
        rightWall, floor, ceiling, backWall, shortBox, tallBox, light -- after the default entries the loader inserts;
      * roughness = 1 / Ns (src/mesh/MeshStorage.cpp:163), index_of_refraction = Ni, diffuse = Kd, specular = Ks, emissive = Ke;
      * triangles are emitted group by group in std::map order of the keys "<g name>:<usemtl name>" (kKeepGroups, :103-120, :438-439).  The file
        says `usemtl shortBox` BEFORE `g shortBox` (and `usemtl tallBox` before `g tallBox`), so the boxes' faces belong to the groups
        "leftWall:shortBox" and "shortBox:tallBox", which sort between "leftWall:leftWall" and "light:light" / after "rightWall:rightWall"."""
    s = scene.load_scene_native(os.path.join(scene.DATA_DIR, "scenes", "CornellBox", "CornellBox-JP.obj"))
    assert s.num_triangles == 36
    names = ["leftWall", "rightWall", "floor", "ceiling", "backWall", "shortBox", "tallBox", "light"]
    Ns = [5.0, 5.0, 6.0, 1.0, 5.0, 5.0, 15.0, 1.0]
    Ni = [1.5, 1.5, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]
    Kd = [(0.63, 0.065, 0.05), (0.2, 0.25, 0.6)] + [(0.725, 0.71, 0.68)] * 5 + [(0.78, 0.78, 0.78)]
    Ks = [(0.5,) * 3, (0.5,) * 3, (0.7,) * 3, (0.0,) * 3, (0.5,) * 3, (0.6,) * 3, (0.6,) * 3, (0.0,) * 3]
    Ke = [(0.0,) * 3] * 7 + [(24.0,) * 3]
    first = len(s.materials) - 8                      # the default material(s) the loader inserts come first
    assert first >= 1
    m = s.materials[first:]
    for k in range(8):
        assert m["roughness"][k] == np.float32(1.0) / np.float32(Ns[k]), names[k]
        assert m["index_of_refraction"][k] == np.float32(Ni[k]), names[k]
        assert np.array_equal(m["diffuse"][k][:3], np.float32(Kd[k])) and np.array_equal(m["specular"][k][:3], np.float32(Ks[k])), names[k]
        assert np.array_equal(m["emissive"][k][:3], np.float32(Ke[k])), names[k]
    order = ["backWall"] * 2 + ["ceiling"] * 2 + ["floor"] * 2 + ["leftWall"] * 2 + ["shortBox"] * 12 + ["light"] * 2 + ["rightWall"] * 2 + ["tallBox"] * 12
    assert [names[i - first] for i in s.material_indices.tolist()] == order
    # geometry facts straight from the `v` lines: the room spans x in [-1.02, 1], y in [0, 1.99], z in [-1.04, 0.99]; the light quad lies at y = 1.98
    lo, hi = s.vertex_data[:, :3].min(0), s.vertex_data[:, :3].max(0)
    assert np.allclose(lo, [-1.02, 0.0, -1.04], atol=1e-6) and np.allclose(hi, [1.0, 1.99, 0.99], atol=1e-6)
    light = np.where(s.material_indices == first + 7)[0]
    assert np.allclose(s.vertex_data[s.vertex_indices[light, :3].reshape(-1), 1], 1.98, atol=1e-6)
