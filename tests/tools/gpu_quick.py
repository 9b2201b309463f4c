"""Quick GPU-vs-oracle parity probe (development aid; the real checks live in tests/)."""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
from gpu_raytracer_amd import api, scenes

def compare(name, scene, w, h, mode=0, use_bvh=True):
    packed = oracle.PackedScene(scene, use_bvh=use_bvh)
    t0 = time.time(); ref = oracle.render_frame(packed, w, h, mode=mode); t_cpu = time.time() - t0
    with api.Context() as ctx:
        t0 = time.time(); ctx.upload_scene(scene); t_up = time.time() - t0
        st = ctx.render(w, h, scene.camera, mode=mode, counters=True)
        st2 = ctx.render(w, h, scene.camera, mode=mode)
        rgb = ctx.read_rgb32f(); comb = ctx.read_rgba8_combined(); prim, t = ctx.read_hits()
    d = np.abs(rgb - ref["rgb"]); dmax = float(np.nanmax(d)) if d.size else 0.0
    prim_mis = int((prim != ref["prim"]).sum()); npx = w * h
    u8 = np.abs(comb.astype(int) - ref["combined"].astype(int)).max()
    t_mis = int((t != ref["t"]).sum())
    exact = int((rgb.view(np.uint32) != ref["rgb"].view(np.uint32)).any(-1).sum())
    print(f"{name:18s} {w}x{h} m{mode} prim_mismatch={prim_mis}/{npx} t_mismatch={t_mis} rgb_not_bitexact={exact} max|drgb|={dmax:.3g} max|du8|={u8} "
          f"gpu_ms={st2['kernel_ms']:.3f} Mrays/s={npx/st2['kernel_ms']/1e3:.1f} nodes/ray={st['node_visits']/max(1,st['rays']):.1f} tris/ray={st['tri_tests']/max(1,st['rays']):.1f} "
          f"depth={st['bvh_depth']} nodes={st['bvh_nodes']} upload_s={t_up:.2f} cpu_s={t_cpu:.2f} oracle_nodes/ray={ref['counters']['node_visits']/npx:.1f} oracle_tris/ray={ref['counters']['tri_tests']/npx:.1f}", flush=True)

print(api.version())
compare("default", scenes.default_scene(), 256, 256)
compare("default", scenes.default_scene(), 256, 256, mode=1)
compare("empty", scenes.empty_scene(), 64, 64, mode=1)
compare("single_triangle", scenes.single_triangle(), 200, 120)
compare("cornell12", scenes.cornell12(), 256, 256)
compare("cornell12 brute", scenes.cornell12(), 256, 256, use_bvh=False)
compare("soup2000", scenes.random_soup(2000, seed=3, n_spheres=3), 320, 200)
compare("soup2000 brute", scenes.random_soup(2000, seed=3, n_spheres=3), 160, 100, use_bvh=False)
compare("soup50000", scenes.random_soup(50000, seed=5, size=0.15), 320, 200)
sp = scenes.sponza_like()
compare("sponza_like", sp, 480, 270)
compare("sponza_like", sp, 480, 270, mode=1)
with api.Context() as ctx:
    ctx.upload_scene(sp)
    for i in range(3):
        st = ctx.render(1920, 1080, sp.camera)
        print("sponza 1080p kernel_ms", st["kernel_ms"], "Mrays/s", 1920*1080/st["kernel_ms"]/1e3, flush=True)
    st = ctx.render(1920, 1080, sp.camera, counters=True)
    print("counters", st, flush=True)

print("---- extended mode ----", flush=True)
def cmp_ext(name, scene, w, h, spp, b, use_bvh=False, **kw):
    ref = oracle.render_extended(oracle.PackedScene(scene, use_bvh=use_bvh), w, h, spp, b, **kw)
    with api.Context() as ctx:
        ctx.upload_scene(scene)
        st = ctx.render(w, h, scene.camera, mode=2, spp=spp, max_bounces=b, frame_seed=kw.get("frame_seed", 0))
        rgb = ctx.read_rgb32f()
    d = np.abs(rgb - ref["rgb"]).max(-1)
    print(f"{name:14s} {w}x{h} {spp}spp {b}b notexact={(rgb.view(np.uint32)!=ref['rgb'].view(np.uint32)).any(-1).sum()} max|d|={d.max():.3g} out={(d>2e-3).mean():.2e} gpu segs=({st['primary_rays']},{st['continuation_rays']},{st['shadow_rays']}) cpu={ref['segments']} ms={st['kernel_ms']:.3f}", flush=True)
cmp_ext("default", scenes.default_scene(), 96, 64, 8, 4)
cmp_ext("cornell12", scenes.cornell12(), 64, 64, 16, 4)
cmp_ext("soup400", scenes.random_soup(400, seed=21, size=0.7, n_spheres=3, n_lights=3), 72, 48, 6, 5, frame_seed=77)
cmp_ext("sponza", sp, 64, 36, 2, 2, use_bvh=True)
with api.Context() as ctx:
    ctx.upload_scene(sp)
    for spp, b in ((1, 0), (4, 4), (16, 4), (64, 4), (64, 0)):
        st = ctx.render(1920, 1080, sp.camera, mode=2, spp=spp, max_bounces=b)
        print(f"sponza 1080p ext {spp}spp {b}b: kernel_ms={st['kernel_ms']:.2f} rays={st['rays']/1e6:.1f}M (cam {st['primary_rays']/1e6:.1f} cont {st['continuation_rays']/1e6:.1f} shadow {st['shadow_rays']/1e6:.1f}) Mrays/s={st['rays']/st['kernel_ms']/1e3:.0f}", flush=True)
    st = ctx.render(1920, 1080, sp.camera, mode=2, spp=4, max_bounces=4, counters=True)
    print("counters 4spp", st, flush=True)
