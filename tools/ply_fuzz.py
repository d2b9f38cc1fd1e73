"""Randomised parity of the PLY loader (Scene::ply, scene/ply.rs): random valid files — ascii /
binary little / big endian, vertex properties in random order with extra properties of every
scalar type and extra list properties, x/y/z or normals stored as double (which the reference
does not read: only float32), face lists with uchar / ushort / int counts and char..uint index
types, quads and pentagons, extra elements before / between / after, comments and obj_info —
product loader vs the oracle's, bit for bit, or both reject.  CPU only.
    ply_fuzz.py [first_seed] [count]"""
import os
import struct
import sys
import tempfile

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
from test_loaders import assert_same_scene
from yuki_amd import loaders
from yuki_amd._ffi import YukiError

TYPES = {"char": "b", "uchar": "B", "short": "h", "ushort": "H", "int": "i", "uint": "I", "float": "f", "double": "d"}
ALIAS = {"char": "int8", "uchar": "uint8", "short": "int16", "ushort": "uint16", "int": "int32", "uint": "uint32", "float": "float32", "double": "float64"}


def write_random_ply(path, seed):
    r = np.random.default_rng(seed)
    fmt = r.choice(["ascii", "binary_little_endian", "binary_big_endian"])
    e = "<" if fmt != "binary_big_endian" else ">"
    nv, nf = int(r.integers(3, 30)), int(r.integers(1, 40))
    tname = lambda t: ALIAS[t] if r.random() < 0.3 else t
    vprops = [("x", "float"), ("y", "float"), ("z", "float")]
    if r.random() < 0.5:
        vprops += [("nx", "float"), ("ny", "float"), ("nz", "float")]
    if r.random() < 0.5:
        vprops += [(r.choice(["u", "s"]), "float"), (r.choice(["v", "t"]), "float")]
    if r.random() < 0.15:  # a coordinate stored as double: not a float32 property -> the reference does not read it
        k = int(r.integers(0, len(vprops)))
        vprops[k] = (vprops[k][0], "double")
    for _ in range(int(r.integers(0, 4))):
        vprops.append((r.choice(["red", "quality", "flag", "w", "confidence"]) + str(int(r.integers(0, 9))), r.choice(list(TYPES))))
    if r.random() < 0.25:  # any order (the reference insists on x..z, nx..nz, u v in sequence: mostly a rejection)
        vprops = [vprops[i] for i in r.permutation(len(vprops))]
    else:  # extras sprinkled between the groups, groups in order
        core = [p for p in vprops if p[0] in ("x", "y", "z", "nx", "ny", "nz", "u", "v", "s", "t")]
        extras = [p for p in vprops if p not in core]
        for p in extras:
            core.insert(int(r.integers(0, len(core) + 1)), p)
        vprops = core
    vlist = r.random() < 0.2  # an extra list property on vertices
    ctype, itype = r.choice(["uchar", "ushort", "int"]), r.choice(["char", "uchar", "short", "ushort", "int", "uint", "int", "uint", "int", "uint"])
    iname = r.choice(["vertex_indices", "vertex_index"])
    fextra = [(f"f{k}", r.choice(list(TYPES))) for k in range(int(r.integers(0, 3)))]
    flist_first = r.random() < 0.7
    extra_elem = r.choice(["none", "before", "between", "after"])
    head = ["ply", f"format {fmt} 1.0"]
    if r.random() < 0.5:
        head.append("comment made by tools/ply_fuzz.py")
    if r.random() < 0.3:
        head.append("obj_info something else")

    def elem_extra():
        return ["element edge 2", f"property {tname('int')} vertex1", f"property {tname('int')} vertex2", f"property {tname('uchar')} crease"]

    if extra_elem == "before":
        head += elem_extra()
    head.append(f"element vertex {nv}")
    for n, t in vprops:
        head.append(f"property {tname(t)} {n}")
    if vlist:
        head.append(f"property list {tname('uchar')} {tname('float')} weights")
    if extra_elem == "between":
        head += elem_extra()
    head.append(f"element face {nf}")
    fl = f"property list {tname(ctype)} {tname(itype)} {iname}"
    fprops = ([fl] if flist_first else []) + [f"property {tname(t)} {n}" for n, t in fextra] + ([] if flist_first else [fl])
    head += fprops
    if extra_elem == "after":
        head += elem_extra()
    head.append("end_header")
    ascii_ = fmt == "ascii"
    out = bytearray(("\n".join(head) + "\n").encode())

    def put(t, v):
        if ascii_:
            out.extend((repr(float(v)) if t in ("float", "double") else str(int(v))).encode() + b" ")
        else:
            out.extend(struct.pack(e + TYPES[t], float(v) if t in ("float", "double") else int(v)))

    def endl():
        if ascii_:
            out.extend(b"\n")

    def edges():
        for _ in range(2):
            put("int", r.integers(0, nv)); put("int", r.integers(0, nv)); put("uchar", r.integers(0, 200)); endl()

    if extra_elem == "before":
        edges()
    for i in range(nv):
        for n, t in vprops:
            if t in ("float", "double"):
                put(t, np.float32(r.uniform(-2, 2)))
            else:
                lo, hi = {"char": (-128, 127), "uchar": (0, 255), "short": (-3000, 3000), "ushort": (0, 60000), "int": (-10**6, 10**6), "uint": (0, 10**6)}[t]
                put(t, r.integers(lo, hi))
        if vlist:
            k = int(r.integers(0, 4))
            put("uchar", k)
            for _ in range(k):
                put("float", r.uniform(0, 1))
        endl()
    if extra_elem == "between":
        edges()
    imax = min(nv, 127 if itype == "char" else 255 if itype == "uchar" else nv)
    for i in range(nf):
        k = int(r.choice([3, 3, 3, 4, 5])) if r.random() < 0.995 else int(r.choice([0, 1, 2]))

        def lst():
            put(ctype, k)
            for _ in range(k):
                put(itype, r.integers(0, imax))

        if flist_first:
            lst()
        for n, t in fextra:
            put(t, r.integers(0, 100) if t not in ("float", "double") else r.uniform(0, 1))
        if not flist_first:
            lst()
        endl()
    if extra_elem == "after":
        edges()
    with open(path, "wb") as f:
        f.write(bytes(out))


def check_seed(seed, d):
    from oracle import loaders as ol

    p = os.path.join(d, f"m{seed}.ply")
    write_random_ply(p, seed)
    try:
        got = loaders.load_ply(p)
    except YukiError as e:
        got = e
    try:
        want = ol.load_ply(p)
    except ol.LoadError as e:
        want = e
    if isinstance(got, Exception) or isinstance(want, Exception):
        if not (isinstance(got, Exception) and isinstance(want, Exception)):
            return f"one side rejected: product {got!r:.150} | oracle {want!r:.150}"
        return None
    assert_same_scene(want[0], got[0])
    return None


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    bad = loaded = 0
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
