"""The Rust FFI declarations (integration/rust/yuki_hip_sys/src/lib.rs — source only, no
Rust toolchain here) stay in step with include/yuki_hip.h: same functions with the same
number of parameters, same struct fields in the same order."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _c_header():
    s = open(os.path.join(ROOT, "include", "yuki_hip.h")).read()
    s = re.sub(r"/\*.*?\*/", "", s, flags=re.S)
    funcs = {}
    for m in re.finditer(r"^(?:yk_status|size_t|void\*?|const char\*|uint32_t|yk_context\*)\s+(yk_\w+)\(([^;]*?)\);", s, flags=re.M | re.S):
        args = m.group(2).strip()
        funcs[m.group(1)] = 0 if args in ("void", "") else len(args.split(","))
    structs = {}
    for m in re.finditer(r"typedef struct (yk_\w+) \{(.*?)\} \1;", s, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                name = re.sub(r"\[.*?\]", "", part.strip().split()[-1]).lstrip("*")
                fields.append(name)
        structs[m.group(1)] = fields
    return funcs, structs


def _rust_lib():
    s = open(os.path.join(ROOT, "integration", "rust", "yuki_hip_sys", "src", "lib.rs")).read()
    funcs = {}
    for m in re.finditer(r"pub fn (yk_\w+)\((.*?)\)(?: -> [^;]+)?;", s, flags=re.S):
        args = m.group(2).strip()
        funcs[m.group(1)] = 0 if not args else len([a for a in args.split(",") if a.strip()])
    structs = {}
    for m in re.finditer(r"pub struct (yk_\w+) \{(.*?)\n\}", s, flags=re.S):
        structs[m.group(1)] = re.findall(r"pub (\w+):", m.group(2))
    return funcs, structs


def test_rust_functions_match_the_header():
    cf, _ = _c_header()
    rf, _ = _rust_lib()
    assert len(rf) >= 30
    for name, n in rf.items():
        assert name in cf, f"{name} is not declared in include/yuki_hip.h"
        assert cf[name] == n, f"{name}: {n} parameters in Rust, {cf[name]} in C"
    # everything a renderer needs is bound (the per-stage test hooks may be left out)
    must = {"yk_context_create", "yk_scene_create", "yk_render_tiles", "yk_render_tiles_accumulating", "yk_film_accumulate_tiles", "yk_load_pbrt", "yk_load_ply", "yk_write_exr", "yk_li"}
    assert must <= set(rf)


def test_rust_structs_match_the_header():
    _, cs = _c_header()
    _, rs = _rust_lib()
    assert len(rs) >= 14
    for name, fields in rs.items():
        assert name in cs, name
        assert cs[name] == fields, f"{name}: C {cs[name]} vs Rust {fields}"


def test_exported_symbols_cover_the_rust_bindings(yk):
    rf, _ = _rust_lib()
    L = yk.lib()
    for name in rf:
        assert hasattr(L, name), name
