import sys, time; sys.path.insert(0,'.')
import numpy as np
from yuki_amd import abi, scenes, core as yk
name=sys.argv[1]; spp=int(sys.argv[2]); res=(int(sys.argv[3]),int(sys.argv[4])); batch=int(sys.argv[5]) if len(sys.argv)>5 else 4<<20
sd=scenes.by_name(name)
streams=int(sys.argv[6]) if len(sys.argv)>6 else 2
ctx=yk.Context(0,batch_paths=batch,streams=streams)
t0=time.time(); sc=yk.Scene(ctx,sd); print('scene',time.time()-t0, sc.info().n_nodes, sc.info().build_seconds)
fs=yk.FilmSettings(res=res); cam=yk.Camera(sd.camera,fs); tiles=yk.film_tiles(fs)
smp=yk.SamplerType.Stratified((int(spp**0.5),int(spp**0.5)),True) if int(spp**0.5)**2==spp else yk.SamplerType.Uniform(spp)
it=yk.IntegratorType.instantiate(ctx, yk.IntegratorType.Path(yk.PathParams(max_depth=8)))
for rep in range(3):
    t0=time.time(); out,st=it.render_tiles(sc,cam,smp,tiles); dt=time.time()-t0
    print(f'wall {dt:.3f}s dev {st.seconds_total:.3f}s rays {st.rays} shadow {st.shadow_rays} Mray/s {st.rays/st.seconds_total*1e-6:.1f} trace {st.seconds_trace:.3f} shadow {st.seconds_shadow:.3f} shade {st.seconds_shade:.3f} batches {st.batches}')
print('mean',out.mean(axis=0))
