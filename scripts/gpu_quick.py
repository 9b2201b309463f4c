"""Quick GPU-vs-oracle parity probe (development aid; the real checks live in tests/)."""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
