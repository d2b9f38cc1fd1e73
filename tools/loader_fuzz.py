"""Randomised parity of the pbrt-v3 / PLY loaders: random VALID scene files in the dialect the
reference reads (scene/pbrt/mod.rs:94-936) — odd number formats, comments, brackets or bare
single values, nested attribute / transform blocks, named materials, every material / light /
shape kind with random parameters, plymesh files (ascii, little and big endian) — loaded by the
product (yk_load_pbrt) and by the independent restatement (oracle/loaders.py); every array,
material constant, light record, camera field must agree bit for bit, or both must reject.
CPU only.    loader_fuzz.py [first_seed] [count]"""
import os
import sys
import tempfile

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import scene_files as sf
from test_loaders import assert_same_camera, assert_same_scene
from yuki_amd import loaders
from yuki_amd._ffi import YukiError


def num(r, lo=-3.0, hi=3.0):
    v = r.uniform(lo, hi)
    k = int(r.integers(0, 8))
    if k == 0:
        return str(int(round(v)))
    if k == 1:
        return f"{v:.3e}"
    if k == 2:
        s = f"{v:.4f}"
        return s.replace("0.", ".", 1) if s.startswith("0.") or s.startswith("-0.") else s
    if k == 3:
        return f"{int(round(v))}."
    if k == 4:
        return f"{v:.9g}"
    if k == 5:
        return "-0" if r.random() < 0.5 else "0"
    if k == 6:
        return f"+{abs(v):.3f}" if r.random() < 0.01 else f"{v:.2f}"  # a leading '+' is not a number to the reference's lexer
    return repr(float(np.float32(v)))


def vals(r, n, lo=-3.0, hi=3.0, bare_ok=False):
    items = " ".join(num(r, lo, hi) for _ in range(n))
    if n == 1 and bare_ok and r.random() < 0.5:
        return items
    pad = r.choice(["", " ", "\n  "])
    return "[" + pad + items + pad + "]"


def rgb(r, lo=0.0, hi=1.0):
    return vals(r, 3, lo, hi)


def material(r, texnames):
    k = r.choice(["matte", "matte", "glass", "metal", "glossy", "plastic"])
    if k == "matte":
        p = []
        if r.random() < 0.7:
            if texnames and r.random() < 0.4:
                p.append(f'"texture Kd" "{r.choice(texnames)}"')
            else:
                p.append(f'"rgb Kd" {rgb(r)}')
        if r.random() < 0.5:
            p.append(f'"float sigma" {vals(r, 1, 0, 60, True)}')
        return f'"matte" ' + " ".join(p)
    if k == "glass":
        p = []
        if r.random() < 0.6:
            p.append(f'"rgb Kr" {rgb(r)}')
        if r.random() < 0.6:
            p.append(f'"rgb Kt" {rgb(r)}')
        if r.random() < 0.6:
            p.append(f'"float {r.choice(["eta", "index"])}" {vals(r, 1, 1, 2.5, True)}')
        return '"glass" ' + " ".join(p)
    if k == "metal":
        p = []
        if r.random() < 0.5:
            p.append(f'"rgb eta" {rgb(r, 0.1, 3)}')
        if r.random() < 0.5:
            p.append(f'"rgb k" {rgb(r, 1, 5)}')
        if r.random() < 0.6:
            p.append(f'"float roughness" {vals(r, 1, 0, 1, True)}')
        if r.random() < 0.4:
            p.append(f'"bool remaproughness" "{r.choice(["true", "false"])}"')
        return '"metal" ' + " ".join(p)
    if k == "glossy":
        p = []
        if r.random() < 0.7:
            p.append(f'"rgb Rs" {rgb(r)}')
        if r.random() < 0.7:
            p.append(f'"float roughness" {vals(r, 1, 0, 1, True)}')
        return '"glossy" ' + " ".join(p)
    return '"plastic"'  # unknown to the reference: what it does with it is part of the contract


def transform(r):
    k = int(r.integers(0, 4))
    if k == 0:
        return f"Translate {num(r)} {num(r)} {num(r)}"
    if k == 1:
        return f"Scale {num(r, 0.2, 2)} {num(r, 0.2, 2)} {num(r, -2, 2)}"
    if k == 2:
        return f"Rotate {num(r, -180, 180)} {num(r)} {num(r)} {num(r, 0.1, 1)}"
    return "# " + r.choice(["a comment", 'Shape "sphere"', "[ 1 2 3"])


def shape(r, plys):
    k = r.choice(["sphere", "trianglemesh", "plymesh", "trianglemesh"])
    if k == "sphere":
        return f'Shape "sphere"' + (f' "float radius" {vals(r, 1, 0.1, 1.5, True)}' if r.random() < 0.8 else "")
    if k == "plymesh":
        return f'Shape "plymesh" "string filename" "{r.choice(plys)}"'
    nv = int(r.integers(3, 9))
    nt = int(r.integers(1, 6))
    s = f'Shape "trianglemesh" "integer indices" [{" ".join(str(int(v)) for v in r.integers(0, nv, 3 * nt))}] "point P" {vals(r, 3 * nv)}'
    if r.random() < 0.4:
        s += f' "normal N" {vals(r, 3 * nv, -1, 1)}'
    if r.random() < 0.4:
        s += f' "float {r.choice(["uv", "st"])}" {vals(r, 2 * nv, -2, 2)}'
    return s


def light(r):
    k = r.choice(["point", "distant", "infinite", "spot"])
    if k == "point":
        return f'LightSource "point" "rgb I" {rgb(r, 0, 50)}' + (f' "point from" {vals(r, 3)}' if r.random() < 0.7 else "")
    if k == "distant":
        return f'LightSource "distant" "rgb L" {rgb(r, 0, 5)} "point from" {vals(r, 3)} "point to" {vals(r, 3)}'
    if k == "infinite":
        return f'LightSource "infinite" "rgb L" {rgb(r, 0, 1)}'
    return f'LightSource "spot" "rgb I" {rgb(r, 0, 50)}'  # not handled by the reference's loader


def block(r, depth, plys, texnames, named):
    out = []
    for _ in range(int(r.integers(1, 6))):
        k = int(r.integers(0, 10))
        if k <= 2:
            out.append(transform(r))
        elif k == 3:
            out.append("Material " + material(r, texnames))
        elif k == 4 and named:
            out.append(f'NamedMaterial "{r.choice(named)}"')
        elif k == 5 and depth < 3:
            kind = r.choice(["Attribute", "Transform"])
            out.append(f"{kind}Begin")
            out += ["  " + l for l in block(r, depth + 1, plys, texnames, named)]
            out.append(f"{kind}End")
        elif k == 6 and depth == 0:
            out.append(light(r))
        else:
            out.append(shape(r, plys))
    return out


def write_random_scene(d, seed):
    r = np.random.default_rng(seed)
    os.makedirs(os.path.join(d, "m"), exist_ok=True)
    sf.write_ascii_ply(os.path.join(d, "m", "a.ply"))
    sf.write_binary_ply(os.path.join(d, "m", "b.ply"), "<")
    sf.write_binary_ply(os.path.join(d, "m", "c.ply"), ">", normals=bool(r.integers(0, 2)), uvs=bool(r.integers(0, 2)))
    plys = ["m/a.ply", "m/b.ply", "m/c.ply"]
    texnames = []
    head = [f"LookAt {num(r)} {num(r, 2, 6)} {num(r)}  {num(r, -0.5, 0.5)} 0 0  0 {num(r, -1, 1)} 1",
            f'Camera "perspective" "float fov" {vals(r, 1, 20, 100, True)}',
            f'Film "image" "integer xresolution" [{int(r.integers(8, 200))}] "integer yresolution" [{int(r.integers(8, 200))}]']
    if r.random() < 0.3:
        head.append('Sampler "halton" "integer pixelsamples" 4')
    if r.random() < 0.3:
        head.append('Integrator "path"')
    body = ["WorldBegin"]
    if r.random() < 0.5:
        sf.write_png(os.path.join(d, "m", "t.png"), sf.test_pattern(5, 4))
        body.append('Texture "tx" "spectrum" "imagemap" "string filename" "m/t.png"')
        texnames.append("tx")
    named = []
    for k in range(int(r.integers(0, 3))):
        m = material(r, texnames)
        kind, rest = m.split(" ", 1) if " " in m else (m, "")
        body.append(f'MakeNamedMaterial "nm{k}" "string type" {kind} {rest}')
        named.append(f"nm{k}")
    body += block(r, 0, plys, texnames, named)
    body.append("WorldEnd")
    p = os.path.join(d, f"s{seed}.pbrt")
    with open(p, "w") as f:
        f.write(("\n" if r.random() < 0.8 else " ").join(head + body) + "\n")
    return p


def check_seed(seed, d):
    from oracle import loaders as ol

    p = write_random_scene(d, seed)
    got = want = None
    try:
        got = loaders.load_pbrt(p)
    except YukiError as e:
        got = e
    try:
        want = ol.load_pbrt(p)
    except ol.LoadError as e:
        want = e
    if isinstance(got, Exception) or isinstance(want, Exception):
        if not (isinstance(got, Exception) and isinstance(want, Exception)):
            return f"one side rejected: product {got!r:.120} oracle {want!r:.120}"
        return None
    assert_same_scene(want[0], got[0])
    assert_same_camera(want[1], got[1], got[2], want[2])
    return None


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    bad = rejected = 0
    with tempfile.TemporaryDirectory() as d:
        for seed in range(first, first + count):
            try:
                msg = check_seed(seed, d)
            except AssertionError as e:
                msg = "DIFFERENT: " + str(e)[:200]
            if msg:
                bad += 1
                print(f"seed {seed}: {msg}", flush=True)
    print(f"{count} seeds, {bad} with differences")
