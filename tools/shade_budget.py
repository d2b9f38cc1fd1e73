"""k_shade's budget on the cfg3 frame, by measurement: the same frame with 0 / 1 / 2 / 3 of the scene's lights
— NEE loops over ALL lights (path.rs:103), so shade(b) = fixed part + n_lights x per-light part — per bounce, from the
YK_DEBUG_BOUNCES breakdown (synchronises after every bounce).  Run on the GPU box: python tools/shade_budget.py"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys; sys.path.insert(0, sys.argv[1])
from yuki_amd import scenes, core as yk
n = int(sys.argv[2])
sd = scenes.by_name("cfg3")
sd.lights = sd.lights[:n]
if n == 0: sd.tri_area_light[:] = -1
ctx = yk.Context(0); sc = yk.Scene(ctx, sd)
fs = yk.FilmSettings(res=(1920, 1080)); cam = yk.Camera(sd.camera, fs); tiles = yk.film_tiles(fs)
it = yk.IntegratorType.instantiate(ctx, yk.IntegratorType.Path(yk.PathParams(max_depth=8)))
for _ in range(2):
    out, st = it.render_tiles(sc, cam, yk.SamplerType.Stratified((8, 8), True), tiles)
print("END", st.rays, st.shadow_rays)
'''
rows = {}
for n in (0, 1, 2, 3):
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, str(n)], capture_output=True, text=True, env=dict(os.environ, YK_DEBUG_BOUNCES="1"))
    lines = [l for l in r.stderr.split("\n") if l.startswith("bounce")]
    last = lines[-8:]  # the second frame
    rows[n] = [(int(re.search(r"rays (\d+)", l).group(1)), float(re.search(r"shade ([\d.]+) ms", l).group(1))) for l in last]
    print(f"# {n} light(s): " + " | ".join(f"b{k}: {v[0] / 1e6:.1f} M vertices, shade {v[1]:.2f} ms" for k, v in enumerate(rows[n][:5])))
print()
print("bounce  vertices(3 lights)  shade ms at 0/1/2/3 lights        fixed ps/vertex   per light ps/vertex (1st, 2nd, 3rd)")
for b in range(5):
    ms = [rows[n][b][1] for n in (0, 1, 2, 3)]
    nv = [rows[n][b][0] for n in (0, 1, 2, 3)]
    ps = [1e9 * ms[k] / max(1, nv[k]) for k in range(4)]  # picoseconds per vertex (the queues differ slightly with the lighting only through roulette: same geometry)
    print(f"  {b}     {nv[3] / 1e6:8.1f} M        {ms[0]:6.2f} {ms[1]:6.2f} {ms[2]:6.2f} {ms[3]:6.2f}        {ps[0]:7.1f}          {ps[1] - ps[0]:6.1f} {ps[2] - ps[1]:6.1f} {ps[3] - ps[2]:6.1f}")
