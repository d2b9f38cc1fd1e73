"""Where a k_shade wave spends its lifetime (diagnostic build: tools/build_variant.sh prof -DYK_SHADE_PROFILE, then
YK_LIB_PATH=yuki_amd/libyuki_hip_prof.so python tools/shade_profile.py).  Per bounce of the cfg3 frame."""
import ctypes as C, sys
sys.path.insert(0, ".")
from yuki_amd import scenes, core as yk, _ffi
L = _ffi.lib()
L.yk_debug_shade_profile.argtypes = [C.c_void_p, C.c_int]
sd = scenes.by_name(sys.argv[1] if len(sys.argv) > 1 else "cfg3")
ctx = yk.Context(0); sc = yk.Scene(ctx, sd)
fs = yk.FilmSettings(res=(1920, 1080)); cam = yk.Camera(sd.camera, fs); tiles = yk.film_tiles(fs)
smp = yk.SamplerType.Stratified((8, 8), True)
names = ["iterations x waves", "material sort", "state load + vertex_setup", "light loop (+ shadow staging)", "vertex_finish", "survivor compaction"]
for depth in (1, 2, 8):
    it = yk.IntegratorType.instantiate(ctx, yk.IntegratorType.Path(yk.PathParams(max_depth=depth)))
    buf = (C.c_ulonglong * 8)()
    it.render_tiles(sc, cam, smp, tiles)
    L.yk_debug_shade_profile(buf, 1)
    out, st = it.render_tiles(sc, cam, smp, tiles)
    L.yk_debug_shade_profile(buf, 1)
    tot = sum(buf[1:6])
    print(f"max_depth {depth}: shade {st.seconds_shade*1e3:.2f} ms, {buf[0]} wave-iterations, {tot / max(1, buf[0]):.0f} cycles per wave-iteration")
    for k in range(1, 6):
        print(f"   {names[k]:32s} {100.0 * buf[k] / tot:5.1f} %   {buf[k] / max(1, buf[0]):8.0f} cycles")
