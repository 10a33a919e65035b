"""Deterministic synthetic stand-ins for the reference's scenes, written in its own file formats.

The reference's Objects/*.off|*.txt are Git-LFS pointer stubs (no geometry available), so every
benchmark/test scene is generated here with a matching triangle budget and material mix:
  lightbox      ~ LightBox*.txt      closed room + emissive panel (+ optional props)
  teapot_like   ~ Teapot.txt         smooth closed metallic/specular body, ~1.4e5 triangles
  dragon_like   ~ DragonGlass.off    bumpy closed dielectric body, ~8.7e5 triangles
  sponza_like   ~ Sponza.off         colonnaded atrium with drapes, several materials, ~2.6e5
Each generator returns a list of (file name, header/material text, triangles) objects; write_objects
turns them into `.txt` (raw triangle list) or `.off` files that ObjectLoader/OFFLoader parse
(formats: Include/ObjectLoader.h:68-229, object_settings.md).
"""
import os

import numpy as np

f32 = np.float32


# ---------------------------------------------------------------------------------------------
# mesh helpers: every function returns float32 triangles of shape (n, 3, 3)
# ---------------------------------------------------------------------------------------------
def _grid_to_tris(P):
    """P: (nu, nv, 3) vertex grid -> two triangles per cell."""
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[:-1, 1:], P[1:, 1:]
    t1 = np.stack([a, b, c], axis=2).reshape(-1, 3, 3)
    t2 = np.stack([b, d, c], axis=2).reshape(-1, 3, 3)
    return np.concatenate([t1, t2]).astype(f32)


def surface(fn, nu, nv, u0=0.0, u1=1.0, v0=0.0, v1=1.0):
    u = np.linspace(u0, u1, nu + 1)
    v = np.linspace(v0, v1, nv + 1)
    U, V = np.meshgrid(u, v, indexing="ij")
    return _grid_to_tris(np.stack(fn(U, V), axis=-1))


def quad(p0, eu, ev, nu=1, nv=1):
    p0, eu, ev = (np.asarray(x, dtype=np.float64) for x in (p0, eu, ev))
    return surface(lambda U, V: tuple(p0[k] + U * eu[k] + V * ev[k] for k in range(3)), nu, nv)


def box(lo, hi):
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    d = hi - lo
    ex, ey, ez = np.array([d[0], 0, 0]), np.array([0, d[1], 0]), np.array([0, 0, d[2]])
    return np.concatenate([quad(lo, ex, ey), quad(lo + ez, ex, ey), quad(lo, ex, ez), quad(lo + ey, ex, ez),
                           quad(lo, ey, ez), quad(lo + ex, ey, ez)])


def sphere(center, radius, nu, nv, bump=None):
    cx, cy, cz = center

    def fn(U, V):
        th, ph = U * 2 * np.pi, V * np.pi
        r = radius * (1.0 if bump is None else bump(th, ph))
        return cx + r * np.sin(ph) * np.cos(th), cy + r * np.sin(ph) * np.sin(th), cz + r * np.cos(ph)
    return surface(fn, nu, nv)


def cylinder(base, radius, height, nseg, nring, flute=0.0, nflutes=16):
    bx, by, bz = base

    def fn(U, V):
        th = U * 2 * np.pi
        r = radius * (1.0 - flute * (0.5 + 0.5 * np.cos(nflutes * th))) * (1.0 - 0.12 * V)
        return bx + r * np.cos(th), by + r * np.sin(th), bz + V * height
    return surface(fn, nseg, nring)


def arch(p_from, p_to, rise, thickness, nseg, nsec):
    """Half-torus-like arch from p_from to p_to (same z), bulging upward by `rise`."""
    a, b = np.asarray(p_from, float), np.asarray(p_to, float)
    mid, half = (a + b) / 2, (b - a) / 2
    span = np.linalg.norm(half)
    ax = half / span
    side = np.cross(ax, [0, 0, 1.0])

    def fn(U, V):
        t = U * np.pi
        ring = V * 2 * np.pi
        cxn = -np.cos(t) * span
        czn = np.sin(t) * rise
        # local frame: radial direction in the arch plane, and `side`
        rx, rz = -np.cos(t), np.sin(t) * rise / span
        n = np.sqrt(rx * rx + rz * rz)
        rx, rz = rx / n, rz / n
        off_r = thickness * np.cos(ring)
        off_s = thickness * np.sin(ring)
        x = mid[0] + ax[0] * (cxn + rx * off_r) + side[0] * off_s
        y = mid[1] + ax[1] * (cxn + rx * off_r) + side[1] * off_s
        z = mid[2] + czn + rz * off_r
        return x, y, z
    return surface(fn, nseg, nsec)


def drape(p0, eu, drop, nu, nv, amp, waves, phase):
    """Hanging cloth: spans eu horizontally, hangs `drop` down, folds along the normal of eu."""
    p0, eu = np.asarray(p0, float), np.asarray(eu, float)
    nrm = np.cross(eu / np.linalg.norm(eu), [0, 0, 1.0])

    def fn(U, V):
        fold = amp * np.sin(waves * 2 * np.pi * U + phase) * (0.3 + 0.7 * V) + 0.3 * amp * np.sin(5 * np.pi * V + U * 9)
        return (p0[0] + U * eu[0] + nrm[0] * fold, p0[1] + U * eu[1] + nrm[1] * fold, p0[2] - V * drop + 0 * U)
    return surface(fn, nu, nv)


# ---------------------------------------------------------------------------------------------
# scenes: list of objects {name, material (header text), tris}
# ---------------------------------------------------------------------------------------------
def _obj(name, material, tris):
    return {"name": name, "material": material, "tris": np.ascontiguousarray(tris, dtype=f32)}


def lightbox(props=True):
    """Closed 12-triangle room around the default camera with an emissive ceiling panel."""
    objs = [_obj("LightBoxRoom.txt", "default-material", box((-3, -4, -2), (9, 4, 3.5))),
            _obj("LightBoxLamp.txt", "lightsource 1 1 1 5", quad((2.5, -1.5, 3.45), (3, 0, 0), (0, 3, 0)))]
    if props:
        objs.append(_obj("LightBoxMirror.txt", "specular 0.9 0.9 0.95 0.05", box((5.5, -3, -2), (7, -1.2, 0.8))))
        objs.append(_obj("LightBoxGlass.off", "transparent 1 1 1 0.9 1.5", sphere((4.5, 1.6, -0.9), 1.1, 24, 16)))
        objs.append(_obj("LightBoxRed.txt", "lambertian 0.8 0.15 0.1", box((3.5, -0.6, -2), (4.6, 0.5, -0.7))))
        objs.append(_obj("LightBoxGold.off", "metallic 0.9 0.7 0.2 0.25 0.9", sphere((6.0, 2.6, -1.2), 0.8, 20, 14)))
    return objs


def teapot_like(target_tris=140_000):
    """Smooth closed body of revolution with a handle-like torus, on a ground slab."""
    n = max(8, int(np.sqrt(target_tris * 0.42)))

    def body(th, ph):
        return 1.0 + 0.25 * np.sin(ph * 2.0) ** 2 + 0.05 * np.cos(6 * th) * np.sin(ph) ** 4
    pot = sphere((5.0, 0.0, -0.2), 1.6, n, n, bump=body)
    m = max(8, int(np.sqrt(target_tris * 0.08)))

    def torus(U, V):
        a, b = U * 2 * np.pi, V * 2 * np.pi
        R, r = 0.9, 0.17
        return 5.0 + 0 * a + r * np.sin(b) * 0.6, 1.9 + (R + r * np.cos(b)) * np.cos(a) * 0.8, -0.1 + (R + r * np.cos(b)) * np.sin(a)
    handle = surface(torus, m, m)
    return [_obj("TeapotBody.txt", "metallic 0.85 0.55 0.35 0.2 0.8", pot),
            _obj("TeapotHandle.txt", "specular 0.9 0.9 0.9 0.1", handle),
            _obj("TeapotGround.txt", "specular 0.6 0.6 0.65 0.6", quad((-6, -12, -2.2), (30, 0, 0), (0, 24, 0), 24, 24))]


def dragon_like(target_tris=870_000):
    """Bumpy closed dielectric body (many thin features -> long refraction chains) over a floor."""
    n = max(8, int(np.sqrt(target_tris * 0.5)))

    def bump(th, ph):
        return (1.0 + 0.22 * np.sin(3 * th + 2 * ph) * np.sin(ph) ** 2 + 0.10 * np.sin(9 * th) * np.sin(7 * ph)
                + 0.04 * np.sin(31 * th + 5) * np.sin(29 * ph))
    body = sphere((5.5, 0.0, 0.0), 1.9, n, n, bump=bump)
    return [_obj("DragonGlass.off", "transparent 1 1 1 0.9 1.5", body),
            _obj("DragonFloor.txt", "lambertian 0.7 0.7 0.72", quad((-6, -12, -2.4), (30, 0, 0), (0, 24, 0), 16, 16)),
            _obj("DragonLamp.txt", "lightsource 1 0.95 0.85 6", quad((3.5, -2, 5.0), (4, 0, 0), (0, 4, 0)))]


def sponza_like(target_tris=260_000):
    """Colonnaded atrium: floor, walls, two rows of fluted columns with arches, hanging drapes,
    statues, an emissive panel and an open roof strip (sky visible)."""
    k = np.sqrt(target_tris / 260_000.0)
    objs = []
    x0, x1, yw, zf, zc = -6.0, 26.0, 8.0, -1.6, 9.0
    # floor (glossy stone) and walls
    objs.append(_obj("SponzaFloor.off", "specular 0.75 0.72 0.68 0.35",
                     quad((x0, -yw, zf), (x1 - x0, 0, 0), (0, 2 * yw, 0), int(96 * k), int(48 * k))))
    walls = [quad((x0, -yw, zf), (x1 - x0, 0, 0), (0, 0, zc - zf), int(64 * k), int(24 * k)),
             quad((x0, yw, zf), (x1 - x0, 0, 0), (0, 0, zc - zf), int(64 * k), int(24 * k)),
             quad((x1, -yw, zf), (0, 2 * yw, 0), (0, 0, zc - zf), int(32 * k), int(24 * k)),
             quad((x0, -yw, zf), (0, 2 * yw, 0), (0, 0, zc - zf), int(32 * k), int(24 * k)),
             # roof in two slabs leaving a strip of open sky along the middle
             quad((x0, -yw, zc), (x1 - x0, 0, 0), (0, yw - 1.5, 0), int(32 * k), int(8 * k)),
             quad((x0, 1.5, zc), (x1 - x0, 0, 0), (0, yw - 1.5, 0), int(32 * k), int(8 * k)),
             # upper gallery floors behind the colonnades
             quad((x0, -yw, 3.6), (x1 - x0, 0, 0), (0, 3.2, 0), int(48 * k), int(6 * k)),
             quad((x0, yw - 3.2, 3.6), (x1 - x0, 0, 0), (0, 3.2, 0), int(48 * k), int(6 * k))]
    objs.append(_obj("SponzaWalls.off", "lambertian 0.78 0.74 0.66", np.concatenate(walls)))
    # columns + capitals + arches
    cols, arches = [], []
    xs = np.linspace(-3.0, 23.0, 10)
    for side in (-4.6, 4.6):
        for x in xs:
            cols.append(cylinder((x, side, zf), 0.48, 5.0, int(56 * k), int(28 * k), flute=0.10))
            cols.append(box((x - 0.65, side - 0.65, zf + 5.0), (x + 0.65, side + 0.65, zf + 5.35)))
            cols.append(box((x - 0.7, side - 0.7, zf), (x + 0.7, side + 0.7, zf + 0.3)))
        for xa, xb in zip(xs[:-1], xs[1:]):
            arches.append(arch((xa, side, zf + 5.35), (xb, side, zf + 5.35), 1.3, 0.22, int(40 * k), int(14 * k)))
    objs.append(_obj("SponzaColumns.off", "lambertian 0.82 0.8 0.75", np.concatenate(cols)))
    objs.append(_obj("SponzaArches.off", "lambertian 0.7 0.66 0.6", np.concatenate(arches)))
    # drapes in three colours
    nd = int(120 * k)
    for name, mat, y, ph in (("SponzaDrapeRed.off", "lambertian 0.75 0.1 0.08", -4.2, 0.3),
                             ("SponzaDrapeGreen.off", "lambertian 0.12 0.55 0.15", 4.2, 1.7),
                             ("SponzaDrapeBlue.off", "lambertian 0.1 0.2 0.7", -4.2, 2.9)):
        xa = {"SponzaDrapeRed.off": 2.0, "SponzaDrapeGreen.off": 8.0, "SponzaDrapeBlue.off": 14.0}[name]
        objs.append(_obj(name, mat, drape((xa, y, 8.2), (4.4, 0, 0), 4.4, nd, nd, 0.22, 5, ph)))
    # statues and a glass orb
    ns = int(110 * k)
    objs.append(_obj("SponzaStatueGold.off", "metallic 0.9 0.72 0.3 0.3 0.9",
                     sphere((9.0, 0.0, -0.2), 1.3, ns, ns, bump=lambda th, ph: 1 + 0.18 * np.sin(4 * th) * np.sin(3 * ph) + 0.05 * np.sin(17 * th + ph * 13))))
    objs.append(_obj("SponzaOrbGlass.off", "transparent 1 1 1 0.9 1.5", sphere((4.0, 1.8, -0.7), 0.9, int(64 * k), int(48 * k))))
    objs.append(_obj("SponzaVase.off", "specular 0.3 0.5 0.6 0.15",
                     sphere((5.0, -2.2, -0.6), 0.95, int(72 * k), int(64 * k), bump=lambda th, ph: 1 + 0.3 * np.sin(2 * ph) ** 2)))
    objs.append(_obj("SponzaLamp.txt", "lightsource 1 0.92 0.8 8", quad((15.0, -1.0, 8.6), (3, 0, 0), (0, 2, 0))))
    return objs


SCENES = {"lightbox": lightbox, "teapot_like": teapot_like, "dragon_like": dragon_like, "sponza_like": sponza_like}


# ---------------------------------------------------------------------------------------------
# writers for the two file formats
# ---------------------------------------------------------------------------------------------
def _fmt(a):
    # repr-exact float32 text so that parsing gives back the same bits
    return np.char.mod("%.9g", a)


def write_raw(path, material, tris, header=""):
    tris = np.asarray(tris, dtype=f32).reshape(-1, 9)
    with open(path, "w") as f:
        f.write(header + material + "\n" + str(len(tris)) + "\n")
        f.write("\n".join(" ".join(row) for row in _fmt(tris)))
        f.write("\n")


def write_off(path, material, tris, header=""):
    tris = np.asarray(tris, dtype=f32).reshape(-1, 3)
    verts, inv = np.unique(tris.view(np.dtype((np.void, 12))).ravel(), return_inverse=True)
    verts = verts.view(f32).reshape(-1, 3)
    faces = inv.reshape(-1, 3)
    with open(path, "w") as f:
        f.write(header + material + "\nOFF\n%d %d 0\n" % (len(verts), len(faces)))
        f.write("\n".join(" ".join(row) for row in _fmt(verts)))
        f.write("\n")
        f.write("\n".join("3 %d %d %d" % tuple(r) for r in faces))
        f.write("\n")


def write_objects(objs, directory):
    """Writes every object into `directory`; returns the file paths in order."""
    os.makedirs(directory, exist_ok=True)
    paths = []
    for o in objs:
        p = os.path.join(directory, o["name"])
        (write_off if ".off" in o["name"] else write_raw)(p, o["material"], o["tris"])
        paths.append(p)
    return paths


def write_ini(path, object_dir, filenames, skybox_path="./Textures/DaylightBox"):
    """A RayTracer.ini with the reference's keys (RayTracer.ini:1-13)."""
    with open(path, "w") as f:
        f.write("[ShaderPaths]\nTracerVert = ./src/shaders/Vertex.glsl\nTracerFrag = ./src/shaders/Fragment.glsl\n\n")
        f.write("[SkyBox]\nPath = %s\n\n[Objects]\nDirectoryPath = %s\nFilenames = %s\n" % (skybox_path, object_dir, ", ".join(filenames)))


# ---------------------------------------------------------------------------------------------
# synthetic textures (the reference's PNGs are not shipped with this repository)
# ---------------------------------------------------------------------------------------------
def noise_texture(width=600, height=400, seed=1234):
    """Uniform RGB8 noise of the size of the reference's Textures/Noise/rgbNoise.png."""
    return np.random.default_rng(seed).integers(0, 256, size=(height, width, 3), dtype=np.uint8)


def sky_cubemap(size=511, seed=99):
    """Procedural daylight cube map, six RGBA8 faces in GL order +X,-X,+Y,-Y,+Z,-Z.
    Note the tracer rotates directions by -pi/2 about x before the lookup, so world 'up' (z)
    ends up on the +Y face."""
    faces = np.zeros((6, size, size, 4), dtype=np.uint8)
    t = (np.arange(size) + 0.5) / size * 2 - 1
    S, T = np.meshgrid(t, t)          # S along x (columns), T along rows
    one = np.ones_like(S)
    dirs = [(one, -T, -S), (-one, -T, S), (S, one, T), (S, -one, -T), (S, -T, one), (-S, -T, -one)]
    sun = np.array([0.45, 0.8, 0.4])
    sun = sun / np.linalg.norm(sun)
    rng = np.random.default_rng(seed)
    for i, (x, y, z) in enumerate(dirs):
        n = np.sqrt(x * x + y * y + z * z)
        x, y, z = x / n, y / n, z / n
        up = np.clip(y, -1, 1)
        horizon = np.exp(-np.abs(up) * 4.0)
        sky = np.stack([0.35 + 0.45 * horizon, 0.55 + 0.35 * horizon, 0.95 - 0.1 * horizon], -1)
        ground = np.stack([0.32 + 0 * up, 0.30 + 0 * up, 0.27 + 0 * up], -1)
        col = np.where((up > 0)[..., None], sky, ground)
        d = np.clip(x * sun[0] + y * sun[1] + z * sun[2], 0, 1)
        col = col + (d ** 200)[..., None] * np.array([1.0, 0.95, 0.8]) + (d ** 8)[..., None] * 0.15
        col = col + rng.normal(0, 0.01, col.shape)
        faces[i, :, :, :3] = np.clip(col * 255.0, 0, 255).astype(np.uint8)
        faces[i, :, :, 3] = 255
    return faces


# glibc rand() with its default seed (srand is never called by the reference), as consumed by
# Scene::Tick (src/Scene.cpp:31): g++ evaluates the three arguments right to left, so the first
# value drawn lands in u_RandSeed.z.
def glibc_rand_stream(count, seed=1):
    r = [0] * (34 + 310 + count)
    r[0] = seed
    for i in range(1, 31):
        r[i] = (16807 * r[i - 1]) % 2147483647
    for i in range(31, 34):
        r[i] = r[i - 31]
    for i in range(34, len(r)):
        r[i] = (r[i - 31] + r[i - 3]) & 0xFFFFFFFF
    return [(v >> 1) for v in r[344:344 + count]]


def frame_seeds(n_frames, first_frame=0):
    """u_RandSeed triples of frames first_frame .. first_frame+n-1 of a fresh reference process."""
    vals = glibc_rand_stream(3 * (first_frame + n_frames))
    out = np.zeros((n_frames, 3), dtype=f32)
    for k in range(n_frames):
        a, b, c = vals[3 * (first_frame + k): 3 * (first_frame + k) + 3]
        # float(rand()) / RAND_MAX : int -> float (rounded), RAND_MAX promoted to float
        z, y, x = (f32(a) / f32(2147483647), f32(b) / f32(2147483647), f32(c) / f32(2147483647))
        out[k] = (x, y, z)
    return out
