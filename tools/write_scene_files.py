#!/usr/bin/env python3
"""BASELINE.json's configurations as the FILES its text names ("Stanford-bunny PLY", "~1 M-tri pbrt-v3 scene"): the generated
meshes of yuki_amd/scenes.py written in the formats the reference's loaders read (tests/scene_files.py), for
`bench.py --scene-file` and the loader tests.

    python tools/write_scene_files.py cfg2 out_dir     ->  out_dir/bunny_class.ply
    python tools/write_scene_files.py cfg3 out_dir     ->  out_dir/scene.pbrt + out_dir/meshes/m*.ply (802 files)
    python tools/write_scene_files.py cfg5 out_dir     ->  the 10,240,012-triangle city (8002 PLY files, 0.4 GB)

The pbrt variant of a city has no rectangular area light (the reference's pbrt loader parses AreaLightSource and ignores it,
scene/pbrt/mod.rs:502): its quad stays as black geometry, the two point lights and the background remain."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scene_files as sf  # noqa: E402

from yuki_amd import scenes  # noqa: E402


def main():
    name, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    t0 = time.time()
    if name == "cfg2":
        p = sf.write_cfg2_ply(os.path.join(out, "bunny_class.ply"))
    else:
        res = (3840, 2160) if name == "cfg5" else (1920, 1080)
        p, info = sf.write_scene_as_pbrt(out, scenes.by_name(name), res=res)
        print(info, file=sys.stderr)
    print(f"{p}  ({time.time() - t0:.1f} s)", file=sys.stderr)
    print(p)


if __name__ == "__main__":
    main()
