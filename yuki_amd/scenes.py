"""Synthetic scenes for the Path-integrator hot path.

No scene assets ship with the reference (`.MISSING_LARGE_BLOBS`, `res/*` ignored),
so every workload is generated here, deterministically, from integer hashes
(never numpy's global RNG).  A scene is a set of *world-space* flat arrays — the
exact data `yk_scene_create` consumes — so everything upstream of this module
(how the points were produced) is outside the parity boundary.

  cornell()      built-in Cornell box restated from yuki/src/scene/mod.rs:154-530
                 (image-textured back wall replaced by white matte)          cfg 1
  bunny_class()  69 312-triangle displaced cube-sphere, PLY defaults of
                 scene/mod.rs:99-152 (white Lambert, point light, camera)    cfg 2
  city()         instanced displaced icospheres in an open box, mixed
                 materials, rect + point lights (~1 M / ~10 M triangles)     cfg 3-5
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import abi

F = np.float32


# --------------------------------------------------------------------------- hashing
def _mix64(x):
    """splitmix64 finaliser on uint64 arrays."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x


def hash_u01(seed, *keys):
    """Deterministic uniform [0,1) from integer keys (arrays broadcast)."""
    with np.errstate(over="ignore"):
        h = np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over="ignore"):
        for k in keys:
            h = _mix64(h + np.asarray(k).astype(np.uint64) * np.uint64(0xD6E8FEB86659FD93) + np.uint64(0x2545F4914F6CDD1D))
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def value_noise3(p, seed):
    """Trilinear value noise on the integer lattice, p: (...,3) float64."""
    pf = np.floor(p)
    t = p - pf
    t = t * t * (3.0 - 2.0 * t)
    i = pf.astype(np.int64) + (1 << 20)
    out = 0.0
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                w = (t[..., 0] if dx else 1 - t[..., 0]) * (t[..., 1] if dy else 1 - t[..., 1]) * (t[..., 2] if dz else 1 - t[..., 2])
                out = out + w * hash_u01(seed, i[..., 0] + dx, i[..., 1] + dy, i[..., 2] + dz)
    return out


def fractal_noise(p, seed, octaves=3):
    amp, freq, tot, norm = 1.0, 1.0, 0.0, 0.0
    for o in range(octaves):
        tot = tot + amp * (value_noise3(p * freq, seed + o) * 2.0 - 1.0)
        norm += amp
        amp *= 0.5
        freq *= 2.0
    return tot / norm


# --------------------------------------------------------------------------- scene container
@dataclass
class SceneData:
    points: np.ndarray  # (nv,3) f32 world space
    indices: np.ndarray  # (nt,3) u32
    tri_mesh: np.ndarray  # (nt,) u32
    tri_material: np.ndarray  # (nt,) i32
    tri_area_light: np.ndarray  # (nt,) i32
    meshes: list  # [(has_normals, has_uvs, swaps_handedness)]
    materials: list  # [dict(kind=, a=, b=, c=, remap=)]
    lights: list  # [dict(kind='point'|'spot'|'distant'|'rect', ...)]
    normals: np.ndarray = None
    uvs: np.ndarray = None
    spheres: list = field(default_factory=list)  # [dict(o2w=4x4, w2o=4x4, radius=, material=)]
    background: tuple = (0.0, 0.0, 0.0)
    split_method: int = abi.SPLIT_SAH
    max_shapes_in_node: int = 1
    camera: dict = None  # position, target, up, fov_axis, fov_degrees
    name: str = ""
    shape_order: np.ndarray = None  # Scene.shapes order (ids: triangles, then spheres); None = natural
    film_res: tuple = None  # FilmSettings.res when the scene came from a loader
    light_structs: list = None  # ready abi.LightDesc values (loaders); overrides `lights`
    textures: list = field(default_factory=list)  # [(h, w, 3) f32 arrays]; a material's `tex` key indexes it

    @property
    def n_triangles(self):
        return int(self.indices.shape[0])

    def light_descs(self, factory):
        """Build LightDesc structs through `factory` — an object exposing
        make_rect_light / make_spot_light / make_point_light (the HIP library's
        host helpers; the tests pass the checker.s own)."""
        if self.light_structs is not None:
            arr = (abi.LightDesc * max(1, len(self.light_structs)))()
            for k, l in enumerate(self.light_structs):
                C.memmove(C.byref(arr[k]), C.byref(l), C.sizeof(abi.LightDesc))
            return arr
        arr = (abi.LightDesc * max(1, len(self.lights)))()
        for k, l in enumerate(self.lights):
            if l["kind"] == "rect":
                factory.make_rect_light(l["l2w"], l["l2w_inv"], l["L"], l["size"], arr[k])
            elif l["kind"] == "spot":
                factory.make_spot_light(l["l2w"], l["l2w_inv"], l["I"], l["total_width"], l["falloff_start"], arr[k])
            elif l["kind"] == "point":
                factory.make_point_light(l["l2w"], l["I"], arr[k])
            elif l["kind"] == "distant":
                arr[k].kind = abi.LIGHT_DISTANT
                arr[k].p = abi.f3(l["w"])
                arr[k].i = abi.f3(l["L"])
            else:
                raise ValueError(l["kind"])
        return arr

    def desc(self, factory):
        """(SceneDesc, keepalive) for yk_scene_create."""
        keep = {}
        d = abi.SceneDesc()
        keep["points"] = np.ascontiguousarray(self.points, dtype=F)
        keep["indices"] = np.ascontiguousarray(self.indices, dtype=np.uint32)
        keep["tri_mesh"] = np.ascontiguousarray(self.tri_mesh, dtype=np.uint32)
        keep["tri_material"] = np.ascontiguousarray(self.tri_material, dtype=np.int32)
        keep["tri_area_light"] = np.ascontiguousarray(self.tri_area_light, dtype=np.int32)
        keep["normals"] = None if self.normals is None else np.ascontiguousarray(self.normals, dtype=F)
        keep["uvs"] = None if self.uvs is None else np.ascontiguousarray(self.uvs, dtype=F)
        d.n_vertices = keep["points"].shape[0]
        d.points = abi.ptr(keep["points"], abi.f32p)
        d.normals = abi.ptr(keep["normals"], abi.f32p)
        d.uvs = abi.ptr(keep["uvs"], abi.f32p)
        d.n_triangles = keep["indices"].shape[0]
        d.indices = abi.ptr(keep["indices"], abi.u32p)
        d.tri_mesh = abi.ptr(keep["tri_mesh"], abi.u32p)
        d.tri_material = abi.ptr(keep["tri_material"], abi.i32p)
        d.tri_area_light = abi.ptr(keep["tri_area_light"], abi.i32p)
        meshes = (abi.MeshDesc * max(1, len(self.meshes)))()
        for k, (hn, hu, sw) in enumerate(self.meshes):
            meshes[k].has_normals, meshes[k].has_uvs, meshes[k].swaps_handedness = int(hn), int(hu), int(sw)
        keep["meshes"] = meshes
        d.n_meshes = len(self.meshes)
        d.meshes = C.cast(meshes, C.POINTER(abi.MeshDesc))
        sph = (abi.SphereDesc * max(1, len(self.spheres)))()
        for k, s in enumerate(self.spheres):
            sph[k].object_to_world = abi.f16(s["o2w"])
            sph[k].world_to_object = abi.f16(s["w2o"])
            sph[k].radius = float(s["radius"])
            sph[k].material = int(s["material"])
        keep["spheres"] = sph
        d.n_spheres = len(self.spheres)
        d.spheres = C.cast(sph, C.POINTER(abi.SphereDesc))
        mats = (abi.MaterialDesc * max(1, len(self.materials)))()
        for k, m in enumerate(self.materials):
            mats[k].kind = m["kind"]
            mats[k].a = abi.f3(m.get("a", (0, 0, 0)))
            mats[k].b = abi.f3(m.get("b", (0, 0, 0)))
            mats[k].c = float(m.get("c", 0.0))
            mats[k].flags = (1 if m.get("remap", False) else 0) | (2 if m.get("tex") is not None else 0)
            mats[k].a_texture = int(m["tex"]) if m.get("tex") is not None else 0
        keep["materials"] = mats
        d.n_materials = len(self.materials)
        d.materials = C.cast(mats, C.POINTER(abi.MaterialDesc))
        lights = self.light_descs(factory)
        keep["lights"] = lights
        d.n_lights = len(self.lights) if self.light_structs is None else len(self.light_structs)
        d.lights = C.cast(lights, C.POINTER(abi.LightDesc))
        d.background = abi.f3(self.background)
        d.split_method = self.split_method
        d.max_shapes_in_node = self.max_shapes_in_node
        texs = (abi.TextureDesc * max(1, len(self.textures)))()
        keep["texture_arrays"] = [np.ascontiguousarray(t, dtype=F) for t in self.textures]
        for k, t in enumerate(keep["texture_arrays"]):
            texs[k].height, texs[k].width = t.shape[0], t.shape[1]
            texs[k].rgb = abi.ptr(t, abi.f32p)
        keep["textures"] = texs
        d.n_textures = len(self.textures)
        d.textures = C.cast(texs, C.POINTER(abi.TextureDesc))
        keep["shape_order"] = None if self.shape_order is None else np.ascontiguousarray(self.shape_order, dtype=np.uint32)
        d.shape_order = abi.ptr(keep["shape_order"], abi.u32p)
        return d, keep


def _mat4_mul(a, b):
    """Matrix4x4 * Matrix4x4 (math/matrix.rs:289-301): four products added left to right in f32."""
    r = np.zeros((4, 4), dtype=F)
    for i in range(4):
        for j in range(4):
            r[i, j] = F(F(F(a[i, 0] * b[0, j]) + F(a[i, 1] * b[1, j])) + F(a[i, 2] * b[2, j])) + F(a[i, 3] * b[3, j])
    return r


def _transform_points(m, p):
    """&Transform * Point3 (math/transform.rs:128-142): m[r][0]*x + m[r][1]*y + m[r][2]*z + m[r][3] in that
    order, in f32, divided by w unless w == 1.  A zero term added to -0.0 gives +0.0, as in the reference."""
    p = np.asarray(p, dtype=F)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    rows = [((m[r, 0] * x + m[r, 1] * y).astype(F) + m[r, 2] * z).astype(F) + m[r, 3] for r in range(4)]
    out = np.stack(rows[:3], axis=1).astype(F)
    w = rows[3].astype(F)
    div = w != F(1)
    out[div] = (out[div] / w[div, None]).astype(F)
    return out


def _translation(v):
    m = np.eye(4, dtype=F)
    m[:3, 3] = np.asarray(v, dtype=F)
    mi = np.eye(4, dtype=F)
    mi[:3, 3] = -np.asarray(v, dtype=F)
    return m, mi


# --------------------------------------------------------------------------- cfg 1: Cornell
def cornell():
    """scene/mod.rs:154-530, constants verbatim; f32 arithmetic throughout."""
    LEFT, RIGHT, BOTTOM, TOP, FRONT, BACK = F(555), F(0), F(0), F(550), F(0), F(560)
    X_CENTER = (LEFT + RIGHT) / F(2)
    Z_CENTER = (FRONT + BACK) / F(2)
    HEIGHT = TOP - BOTTOM
    LIGHT_WH = F(100)
    LIGHT_HALF_WH = LIGHT_WH / F(2)
    LIGHT_FRONT, LIGHT_BACK = Z_CENTER - LIGHT_HALF_WH, Z_CENTER + LIGHT_HALF_WH
    LIGHT_LEFT, LIGHT_RIGHT = X_CENTER + LIGHT_HALF_WH, X_CENTER - LIGHT_HALF_WH
    HOLE_TOP = TOP + HEIGHT * F(0.025)

    WHITE, IMAGE, RED, GREEN, BLACKBODY, COPPER, GLASS = range(7)
    c180 = F(1) * F(180) / F(255)
    materials = [
        dict(kind=abi.MAT_MATTE, a=(c180, c180, c180), c=0.0),
        dict(kind=abi.MAT_MATTE, a=(c180, c180, c180), c=0.0),  # image texture -> white matte
        dict(kind=abi.MAT_MATTE, a=(F(180) / F(255), F(0) / F(255), F(0) / F(255)), c=0.0),
        dict(kind=abi.MAT_MATTE, a=(F(0) / F(255), F(180) / F(255), F(0) / F(255)), c=0.0),
        dict(kind=abi.MAT_MATTE, a=(0, 0, 0), c=0.0),
        dict(kind=abi.MAT_METAL, a=(0.27105, 0.67693, 1.31640), b=(3.60920, 2.62480, 2.29210), c=0.01, remap=True),
        dict(kind=abi.MAT_GLASS, a=(1, 1, 1), b=(1, 1, 1), c=1.5),
    ]

    size = F(100) / F(1000)
    area = size * size
    radiance = F(2.0) / (area * F(np.pi))
    l2w, l2w_inv = _translation((X_CENTER / F(1000), HOLE_TOP / F(1000), (-Z_CENTER) / F(1000)))
    lights = [dict(kind="rect", l2w=l2w, l2w_inv=l2w_inv, L=(radiance,) * 3, size=(size, size))]

    quad = [0, 1, 2, 0, 2, 3]
    mesh_defs = []  # (indices, points, uvs, material, area_light)
    mesh_defs.append((quad, [(LIGHT_RIGHT, HOLE_TOP, LIGHT_FRONT), (LIGHT_LEFT, HOLE_TOP, LIGHT_FRONT), (LIGHT_LEFT, HOLE_TOP, LIGHT_BACK), (LIGHT_RIGHT, HOLE_TOP, LIGHT_BACK)], None, BLACKBODY, 0))
    walls = [
        (quad, [(RIGHT, BOTTOM, BACK), (LEFT, BOTTOM, BACK), (LEFT, BOTTOM, FRONT), (RIGHT, BOTTOM, FRONT)], None, WHITE),
        (quad, [(RIGHT, TOP, FRONT), (LEFT, TOP, FRONT), (LEFT, TOP, LIGHT_FRONT), (RIGHT, TOP, LIGHT_FRONT)], None, WHITE),
        (quad, [(RIGHT, TOP, LIGHT_BACK), (LEFT, TOP, LIGHT_BACK), (LEFT, TOP, BACK), (RIGHT, TOP, BACK)], None, WHITE),
        (quad, [(LIGHT_LEFT, TOP, FRONT), (LEFT, TOP, FRONT), (LEFT, TOP, BACK), (LIGHT_LEFT, TOP, BACK)], None, WHITE),
        (quad, [(RIGHT, TOP, FRONT), (LIGHT_RIGHT, TOP, FRONT), (LIGHT_RIGHT, TOP, BACK), (RIGHT, TOP, BACK)], None, WHITE),
        ([0, 2, 1, 0, 3, 2], [(LIGHT_RIGHT, HOLE_TOP, LIGHT_FRONT), (LIGHT_LEFT, HOLE_TOP, LIGHT_FRONT), (LIGHT_LEFT, TOP, LIGHT_FRONT), (LIGHT_RIGHT, TOP, LIGHT_FRONT)], None, WHITE),
        (quad, [(LIGHT_RIGHT, HOLE_TOP, LIGHT_BACK), (LIGHT_LEFT, HOLE_TOP, LIGHT_BACK), (LIGHT_LEFT, TOP, LIGHT_BACK), (LIGHT_RIGHT, TOP, LIGHT_BACK)], None, WHITE),
        (quad, [(LIGHT_LEFT, TOP, LIGHT_FRONT), (LIGHT_LEFT, TOP, LIGHT_BACK), (LIGHT_LEFT, HOLE_TOP, LIGHT_BACK), (LIGHT_LEFT, HOLE_TOP, LIGHT_FRONT)], None, WHITE),
        (quad, [(LIGHT_RIGHT, HOLE_TOP, LIGHT_FRONT), (LIGHT_RIGHT, HOLE_TOP, LIGHT_BACK), (LIGHT_RIGHT, TOP, LIGHT_BACK), (LIGHT_RIGHT, TOP, LIGHT_FRONT)], None, WHITE),
        (quad, [(RIGHT, TOP, BACK), (LEFT, TOP, BACK), (LEFT, BOTTOM, BACK), (RIGHT, BOTTOM, BACK)], [(0, 0), (0, 1), (1, 1), (1, 0)], IMAGE),
        (quad, [(RIGHT, TOP, FRONT), (RIGHT, TOP, BACK), (RIGHT, BOTTOM, BACK), (RIGHT, BOTTOM, FRONT)], None, GREEN),
        (quad, [(LEFT, BOTTOM, FRONT), (LEFT, BOTTOM, BACK), (LEFT, TOP, BACK), (LEFT, TOP, FRONT)], None, RED),
    ]
    for w in walls:
        mesh_defs.append((w[0], w[1], w[2], w[3], -1))
    tall_idx = [0, 1, 2, 0, 2, 3, 4, 0, 3, 4, 3, 5, 5, 3, 2, 5, 2, 6, 6, 2, 1, 6, 1, 7, 7, 1, 0, 7, 0, 4]
    tall_pts = [(423, 330, 247), (265, 330, 296), (314, 330, 456), (472, 330, 406), (423, 0, 247), (472, 0, 406), (314, 0, 456), (265, 0, 296)]
    mesh_defs.append((tall_idx, tall_pts, None, GLASS, -1))

    pts, uvs, idx, tmesh, tmat, tal, meshes = [], [], [], [], [], [], []
    base = 0
    # scene/mod.rs:177-185: into_meters * handedness_swap, applied by Mesh::new (shapes/mesh.rs:27-29)
    handedness_swap = np.diag(np.asarray([1, 1, -1, 1], dtype=F))
    into_meters = np.diag(np.asarray([0.001, 0.001, 0.001, 1], dtype=F))
    to_world = _mat4_mul(into_meters, handedness_swap)
    for mi, (ind, p, uv, mat, al) in enumerate(mesh_defs):
        w = _transform_points(to_world, p)
        pts.append(w)
        uvs.append(np.asarray(uv, dtype=F) if uv is not None else np.zeros((len(p), 2), dtype=F))
        ii = np.asarray(ind, dtype=np.uint32).reshape(-1, 3) + np.uint32(base)
        idx.append(ii)
        tmesh += [mi] * len(ii)
        tmat += [mat] * len(ii)
        tal += [al] * len(ii)
        meshes.append((False, uv is not None, True))
        base += len(p)
    o2w, w2o = _translation((0.186, 0.082, -0.168))
    return SceneData(
        points=np.concatenate(pts),
        uvs=np.concatenate(uvs),
        indices=np.concatenate(idx),
        tri_mesh=np.asarray(tmesh, dtype=np.uint32),
        tri_material=np.asarray(tmat, dtype=np.int32),
        tri_area_light=np.asarray(tal, dtype=np.int32),
        meshes=meshes,
        materials=materials,
        lights=lights,
        spheres=[dict(o2w=o2w, w2o=w2o, radius=0.082, material=COPPER)],
        background=(0, 0, 0),
        split_method=abi.SPLIT_MIDDLE,
        max_shapes_in_node=1,
        camera=dict(position=(0.278, 0.273, 0.800), target=(0.278, 0.273, -0.260), up=(0, 1, 0), fov_axis=abi.FOV_X, fov_degrees=40.0),
        name="cornell",
    )


def cornell_triangles_only():
    """Cornell without the copper sphere — the HIP path renders triangle meshes."""
    s = cornell()
    s.spheres = []
    s.name = "cornell-tris"
    return s


def glass_balls():
    """Whitted test scene: the Cornell box with the copper sphere turned to glass, two more glass
    spheres (one tinted, one with a different index, overlapping the glass box's shadow), a
    point light beside the area light and a non-black background seen through the light hole —
    nested reflection / transmission subtrees, total internal reflection inside the box, the
    emitter seen through glass."""
    s = cornell()
    glass = len(s.materials) - 1
    s.materials = list(s.materials) + [dict(kind=abi.MAT_GLASS, a=(0.9, 0.95, 1.0), b=(0.6, 0.9, 0.7), c=1.33), dict(kind=abi.MAT_GLASS, a=(1, 0.8, 0.7), b=(1, 1, 1), c=2.4)]
    spheres = []
    for (x, y, z), r, m in (((0.186, 0.082, -0.168), 0.082, glass), ((0.30, 0.40, -0.25), 0.06, glass + 1), ((0.40, 0.10, -0.12), 0.05, glass + 2)):
        o2w, w2o = _translation((x, y, z))
        spheres.append(dict(o2w=o2w, w2o=w2o, radius=r, material=m))
    s.spheres = spheres
    l2w, _ = _translation((0.45, 0.45, -0.05))
    s.lights = list(s.lights) + [dict(kind="point", l2w=l2w, I=(0.05, 0.045, 0.04))]
    s.background = (0.05, 0.07, 0.1)
    s.name = "glass-balls"
    return s


def coplanar_slabs(seed=7):
    """Regression scene for tie hits that RAISE t_max: three slabs of heavily
    overlapping random triangles, each slab in one plane (y = -0, x = 0.5, z = -1), so that most
    rays hit many triangles at distances that differ by rounding only; per-vertex normals make
    the winner of a tie visible in the shading-normals integrator."""
    r = np.random.default_rng(seed)
    pts, nrm, idx, tmesh = [], [], [], []
    base = 0
    for mi, (axis, value) in enumerate(((1, -0.0), (0, 0.5), (2, -1.0))):
        nv, nt = 60, 160
        p = r.uniform(-1.5, 1.5, (nv, 3)).astype(F)
        p[:, axis] = F(value)
        n = r.normal(size=(nv, 3)).astype(F)
        n /= np.linalg.norm(n, axis=1, keepdims=True).astype(F)
        pts.append(p)
        nrm.append(n.astype(F))
        idx.append(r.integers(0, nv, (nt, 3)).astype(np.uint32) + np.uint32(base))
        tmesh += [mi] * nt
        base += nv
    nt = len(tmesh)
    l2w, _ = _translation((2.0, 2.5, 1.5))
    return SceneData(
        points=np.concatenate(pts), normals=np.concatenate(nrm), indices=np.concatenate(idx), tri_mesh=np.asarray(tmesh, dtype=np.uint32),
        tri_material=(np.arange(nt) % 3).astype(np.int32), tri_area_light=np.full(nt, -1, dtype=np.int32), meshes=[(True, False, False)] * 3,
        materials=[dict(kind=abi.MAT_MATTE, a=(0.8, 0.3, 0.2), c=0.0), dict(kind=abi.MAT_MATTE, a=(0.2, 0.7, 0.3), c=0.4), dict(kind=abi.MAT_GLOSSY, a=(0.5, 0.5, 0.9), c=0.3)],
        lights=[dict(kind="point", l2w=l2w, I=(30.0, 30.0, 30.0))], background=(0.1, 0.1, 0.1), split_method=abi.SPLIT_MIDDLE, max_shapes_in_node=4,
        camera=dict(position=(2.3, -0.1, -1.3), target=(0.0, 0.0, 0.0), up=(0, 1, 0), fov_axis=abi.FOV_Y, fov_degrees=70.0), name="coplanar-slabs")


def deep_chain(n=70):
    """A BVH as deep as it has shapes: n large triangles in the planes x = 3^-k.  The Middle
    split (bvh.rs:341-372) cuts the centroid range in half, which separates the one triangle
    with the largest x from all the others, level after level.  A ray travelling towards +x
    enters the nested child first and defers the single-triangle child every time, so its
    traversal stack holds n - 1 entries before the first leaf is visited: n > 65 exceeds the
    reference's 64-entry stack (assert at bvh.rs:174)."""
    xs = (F(3.0) ** -np.arange(n, dtype=np.float64)).astype(F)
    pts = np.zeros((3 * n, 3), dtype=F)
    for k in range(n):
        pts[3 * k] = (xs[k], -2.0, -2.0)
        pts[3 * k + 1] = (xs[k], 2.0, -2.0)
        pts[3 * k + 2] = (xs[k], 0.0, 2.0)
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    l2w, _ = _translation((-1.0, 0.5, 0.0))
    return SceneData(
        points=pts, indices=idx, tri_mesh=np.zeros(n, dtype=np.uint32), tri_material=(np.arange(n) % 2).astype(np.int32),
        tri_area_light=np.full(n, -1, dtype=np.int32), meshes=[(False, False, False)],
        materials=[dict(kind=abi.MAT_MATTE, a=(0.7, 0.6, 0.5), c=0.0), dict(kind=abi.MAT_GLASS, a=(1.0, 1.0, 1.0), b=(1.0, 1.0, 1.0), c=1.5)],
        lights=[dict(kind="point", l2w=l2w, I=(4.0, 4.0, 4.0))], background=(0.2, 0.2, 0.2), split_method=abi.SPLIT_MIDDLE, max_shapes_in_node=1,
        camera=dict(position=(-1.0, 0.0, 0.0), target=(1.0, 0.0, 0.0), up=(0, 1, 0), fov_axis=abi.FOV_X, fov_degrees=20.0), name=f"deep-chain-{n}")


# --------------------------------------------------------------------------- cfg 2: bunny-class mesh
def _cube_sphere(n):
    """6 faces x n x n quads; returns unit-sphere points (nv,3) f64 and triangles (nt,3)."""
    g = np.linspace(-1.0, 1.0, n + 1)
    u, v = np.meshgrid(g, g, indexing="xy")
    one = np.ones_like(u)
    faces = [(one, v, -u), (-one, v, u), (u, one, -v), (u, -one, v), (u, v, one), (-u, v, -one)]
    pts, tris = [], []
    base = 0
    jj, ii = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    a = (jj * (n + 1) + ii).ravel()
    b, c, d = a + 1, a + n + 2, a + n + 1
    for fx, fy, fz in faces:
        p = np.stack([fx.ravel(), fy.ravel(), fz.ravel()], axis=1)
        p /= np.linalg.norm(p, axis=1, keepdims=True)
        pts.append(p)
        tris.append(np.stack([a, b, c], axis=1) + base)
        tris.append(np.stack([a, c, d], axis=1) + base)
        base += p.shape[0]
    return np.concatenate(pts), np.concatenate(tris)


def _fit_unit(points64):
    """scene/ply.rs:99-108: scale(1/max_extent) * translation(-center), f32."""
    p = points64.astype(F)
    bmin, bmax = p.min(axis=0), p.max(axis=0)
    diag = bmax - bmin
    center = bmin + diag / F(2)
    s = F(1) / diag.max()
    return (s * p + s * (-center)).astype(F)


def bunny_class_raw(n=76, seed=0xB077E):
    """The mesh as a PLY file holds it: float32 vertices BEFORE Scene::ply's fit-to-unit-cube transform, and the faces
    (tests/scene_files.py::write_cfg2_ply writes exactly this)."""
    sph, tris = _cube_sphere(n)
    disp = 1.0 + 0.15 * fractal_noise(sph * 2.5, seed, 3)
    raw = (sph * disp[:, None]).astype(F)
    pts = _fit_unit(raw)
    # orient triangles outward (counter-clockwise seen from outside)
    p0, p1, p2 = pts[tris[:, 0]], pts[tris[:, 1]], pts[tris[:, 2]]
    nrm = np.cross(p1 - p0, p2 - p0)
    flip = (nrm * (p0 + p1 + p2)).sum(axis=1) < 0
    tris[flip] = tris[flip][:, [0, 2, 1]]
    return raw, tris


def bunny_class(n=76, seed=0xB077E):
    raw, tris = bunny_class_raw(n, seed)
    pts = _fit_unit(raw)
    nt = tris.shape[0]
    l2w, _ = _translation((5.0, 5.0, 0.0))
    return SceneData(
        points=pts,
        indices=tris.astype(np.uint32),
        tri_mesh=np.zeros(nt, dtype=np.uint32),
        tri_material=np.zeros(nt, dtype=np.int32),
        tri_area_light=np.full(nt, -1, dtype=np.int32),
        meshes=[(False, False, False)],
        materials=[dict(kind=abi.MAT_MATTE, a=(1, 1, 1), c=0.0)],
        lights=[dict(kind="point", l2w=l2w, I=(600.0, 600.0, 600.0))],
        background=(0, 0, 0),
        split_method=abi.SPLIT_SAH,
        max_shapes_in_node=1,
        camera=dict(position=(2, 2, 2), target=(0, 0, 0), up=(0, 1, 0), fov_axis=abi.FOV_X, fov_degrees=40.0),
        name=f"bunny-class-{nt}",
    )


# --------------------------------------------------------------------------- cfg 3-5: instanced city
def _icosphere(level):
    t = (1.0 + 5.0**0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.asarray(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(level):
        cache, nf = {}, []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[k] = len(v) - 1
            return cache[k]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.asarray(v), np.asarray(f, dtype=np.int64)


def _vertex_normals(p, tris):
    fn = np.cross(p[tris[:, 1]] - p[tris[:, 0]], p[tris[:, 2]] - p[tris[:, 0]])
    n = np.zeros_like(p)
    for k in range(3):
        np.add.at(n, tris[:, k], fn)
    return n / np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)


def _rotations(axis, angle):
    a = axis / np.linalg.norm(axis, axis=1, keepdims=True)
    c, s = np.cos(angle), np.sin(angle)
    x, y, z = a[:, 0], a[:, 1], a[:, 2]
    R = np.empty((len(angle), 3, 3))
    R[:, 0, 0] = c + x * x * (1 - c); R[:, 0, 1] = x * y * (1 - c) - z * s; R[:, 0, 2] = x * z * (1 - c) + y * s
    R[:, 1, 0] = y * x * (1 - c) + z * s; R[:, 1, 1] = c + y * y * (1 - c); R[:, 1, 2] = y * z * (1 - c) - x * s
    R[:, 2, 0] = z * x * (1 - c) - y * s; R[:, 2, 1] = z * y * (1 - c) + x * s; R[:, 2, 2] = c + z * z * (1 - c)
    return R


def city(grid=(40, 20), level=3, seed=1, mix="mixed", max_shapes_in_node=1):
    """grid[0]*grid[1] displaced icospheres (20*4^level triangles each) in an open
    box with a rectangular area light and two point lights.

    mix="mixed": 50 % Lambert, 15 % Oren-Nayar, 15 % copper-like metal,
                 10 % glossy, 10 % glass (cfg 3/4)
    mix="metal_glass": 50 % GGX metal / 50 % glass (cfg 5)
    """
    gx, gz = grid
    ninst = gx * gz
    base_v, base_f = _icosphere(level)
    disp = 1.0 + 0.18 * fractal_noise(base_v * 2.0, 77 + seed, 3)
    base_p = base_v * disp[:, None]
    base_n = _vertex_normals(base_p, base_f)
    nv, nf = base_p.shape[0], base_f.shape[0]
    inst = np.arange(ninst)
    ix, iz = inst % gx, inst // gx
    scale = 0.6 + 0.4 * hash_u01(seed, inst, 1)
    radius = 0.36 * scale
    jx = (hash_u01(seed, inst, 2) - 0.5) * 0.25
    jz = (hash_u01(seed, inst, 3) - 0.5) * 0.25
    cx, cz = ix + 0.5 + jx, iz + 0.5 + jz
    cy = radius * (0.82 + 0.6 * hash_u01(seed, inst, 4))
    axis = np.stack([hash_u01(seed, inst, 5) - 0.5, hash_u01(seed, inst, 6) - 0.5, hash_u01(seed, inst, 7) - 0.5], axis=1) + 1e-3
    R = _rotations(axis, hash_u01(seed, inst, 8) * 2 * np.pi)
    P = np.einsum("nij,vj->nvi", R, base_p) * radius[:, None, None] + np.stack([cx, cy, cz], axis=1)[:, None, :]
    N = np.einsum("nij,vj->nvi", R, base_n)
    pts = [P.reshape(-1, 3)]
    nrm = [N.reshape(-1, 3)]
    idx = [(base_f[None, :, :] + (inst * nv)[:, None, None]).reshape(-1, 3)]
    tri_mesh = [np.repeat(inst, nf)]

    # materials: one per instance
    h = hash_u01(seed, inst, 9)
    c0, c1, c2 = hash_u01(seed, inst, 10), hash_u01(seed, inst, 11), hash_u01(seed, inst, 12)
    r1 = hash_u01(seed, inst, 13)
    materials = []
    for k in range(ninst):
        col = (0.2 + 0.7 * c0[k], 0.2 + 0.7 * c1[k], 0.2 + 0.7 * c2[k])
        if mix == "mixed":
            if h[k] < 0.50:
                m = dict(kind=abi.MAT_MATTE, a=col, c=0.0)
            elif h[k] < 0.65:
                m = dict(kind=abi.MAT_MATTE, a=col, c=float(np.float32(np.deg2rad(20.0))))
            elif h[k] < 0.80:
                m = dict(kind=abi.MAT_METAL, a=(0.27105, 0.67693, 1.31640), b=(3.60920, 2.62480, 2.29210), c=0.01 + 0.29 * r1[k], remap=True)
            elif h[k] < 0.90:
                m = dict(kind=abi.MAT_GLOSSY, a=col, c=0.1 + 0.5 * r1[k], remap=False)
            else:
                m = dict(kind=abi.MAT_GLASS, a=(1, 1, 1), b=(1, 1, 1), c=1.5)
        else:
            if h[k] < 0.5:
                m = dict(kind=abi.MAT_METAL, a=(0.27105, 0.67693, 1.31640), b=(3.60920, 2.62480, 2.29210), c=0.01 + 0.29 * r1[k], remap=True)
            else:
                m = dict(kind=abi.MAT_GLASS, a=(1, 1, 1), b=(1, 1, 1), c=1.5)
        materials.append(m)
    tri_mat = [np.repeat(inst, nf)]
    # every second instance carries shading normals
    meshes = [(bool(k % 2 == 0), False, False) for k in range(ninst)]

    GROUND, LIGHTMAT = ninst, ninst + 1
    materials.append(dict(kind=abi.MAT_MATTE, a=(0.55, 0.55, 0.55), c=0.0))
    materials.append(dict(kind=abi.MAT_MATTE, a=(0, 0, 0), c=0.0))
    W, D, H = float(gx), float(gz), 3.0
    box_p = np.array([(0, 0, 0), (W, 0, 0), (W, 0, D), (0, 0, D), (0, H, 0), (W, H, 0), (W, H, D), (0, H, D)], dtype=np.float64)
    box_f = np.array([(0, 2, 1), (0, 3, 2), (0, 1, 5), (0, 5, 4), (1, 2, 6), (1, 6, 5), (2, 3, 7), (2, 7, 6), (3, 0, 4), (3, 4, 7)], dtype=np.int64)
    off = ninst * nv
    pts.append(box_p)
    nrm.append(np.zeros_like(box_p))
    idx.append(box_f + off)
    tri_mesh.append(np.full(len(box_f), ninst))
    tri_mat.append(np.full(len(box_f), GROUND))
    meshes.append((False, True, False))
    box_uv = np.array([(0, 0), (1, 0), (1, 1), (0, 1), (0, 1), (1, 1), (1, 0), (0, 0)], dtype=np.float64)
    off += len(box_p)

    # rectangular area light above the scene, facing -y (identity orientation)
    LY = 5.0
    lsx, lsz = 0.6 * W, 0.6 * D
    lc = (W / 2, LY, D / 2)
    lp = np.array([(lc[0] - lsx / 2, LY, lc[2] - lsz / 2), (lc[0] + lsx / 2, LY, lc[2] - lsz / 2), (lc[0] + lsx / 2, LY, lc[2] + lsz / 2), (lc[0] - lsx / 2, LY, lc[2] + lsz / 2)])
    lf = np.array([(0, 1, 2), (0, 2, 3)], dtype=np.int64)  # geometric normal points down (-y)
    pts.append(lp)
    nrm.append(np.zeros_like(lp))
    idx.append(lf + off)
    tri_mesh.append(np.full(2, ninst + 1))
    tri_mat.append(np.full(2, LIGHTMAT))
    meshes.append((False, False, False))
    ntri = ninst * nf + len(box_f) + 2
    tal = np.full(ntri, -1, dtype=np.int32)
    tal[-2:] = 0
    l2w, l2w_inv = _translation(lc)
    pl1, _ = _translation((0.25 * W, 2.2, 0.3 * D))
    pl2, _ = _translation((0.75 * W, 2.6, 0.7 * D))
    lights = [
        dict(kind="rect", l2w=l2w, l2w_inv=l2w_inv, L=(1.6, 1.5, 1.4), size=(lsx, lsz)),
        dict(kind="point", l2w=pl1, I=(60.0, 50.0, 40.0)),
        dict(kind="point", l2w=pl2, I=(35.0, 45.0, 60.0)),
    ]
    uvs = np.zeros((ninst * nv + len(box_p) + len(lp), 2))
    uvs[ninst * nv : ninst * nv + len(box_p)] = box_uv
    return SceneData(
        points=np.concatenate(pts).astype(F),
        normals=np.concatenate(nrm).astype(F),
        uvs=uvs.astype(F),
        indices=np.concatenate(idx).astype(np.uint32),
        tri_mesh=np.concatenate(tri_mesh).astype(np.uint32),
        tri_material=np.concatenate(tri_mat).astype(np.int32),
        tri_area_light=tal,
        meshes=meshes,
        materials=materials,
        lights=lights,
        background=(0.04, 0.05, 0.07),
        split_method=abi.SPLIT_SAH,
        max_shapes_in_node=max_shapes_in_node,
        camera=dict(position=(0.06 * W, 2.4, 0.08 * D), target=(0.7 * W, 0.2, 0.65 * D), up=(0, 1, 0), fov_axis=abi.FOV_X, fov_degrees=55.0),
        name=f"city-{gx}x{gz}-L{level}-{mix}-{ntri}",
    )


def by_name(name):
    """Named workloads of BASELINE.json `configs`."""
    if name == "cornell":
        return cornell()
    if name == "cornell-tris":
        return cornell_triangles_only()
    if name == "glass-balls":
        return glass_balls()
    if name == "coplanar-slabs":
        return coplanar_slabs()
    if name.startswith("deep-chain-"):
        return deep_chain(int(name.split("-")[-1]))
    if name == "cfg2":
        return bunny_class()
    if name == "cfg3":
        return city((40, 20), 3, 1, "mixed")
    if name == "cfg5":
        return city((100, 80), 3, 1, "metal_glass")
    if name == "city-small":
        return city((6, 4), 2, 1, "mixed")
    if name == "city-tiny":
        return city((3, 2), 1, 1, "mixed")
    raise KeyError(name)
