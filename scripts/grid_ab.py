"""Light grids (csrc/shadow_grid.h) against the BVH for the shadow segments: same frame, time, list statistics (development aid).
usage: grid_ab.py [scene] [width height spp bounces]"""
import sys, os, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_raytracer_amd import api, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "sponza_like"
w, h, spp, bounces = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (1920, 1080, 16, 4)
sc = scenes.SCENES[name]()
with api.Context() as ctx:
    t0 = time.time(); ctx.upload_scene(sc); print(f"{name}: upload {1e3 * (time.time() - t0):.0f} ms", flush=True)
    t0 = time.time(); ctx.upload_scene(sc); print(f"{name}: second upload {1e3 * (time.time() - t0):.0f} ms", flush=True)
    t0 = time.time(); ctx.prepare(); print(f"{name}: light grids {1e3 * (time.time() - t0):.0f} ms", flush=True)
    print("grids:", ctx.debug_shadow_grid())
    for li in range(len(sc.lights)): print("  light", li, ctx.debug_shadow_grid(li))
    crc = {}
    for label, kw in (("bvh", {"no_shadow_grid": True}), ("grid", {})):
        for rep in range(3):
            st = ctx.render(w, h, sc.camera, mode=2, spp=spp, max_bounces=bounces, **kw)
        crc[label] = zlib.crc32(ctx.read_rgb32f().tobytes())
        print(f"{label}: {st['kernel_ms']:.2f} ms, {st['rays'] / st['kernel_ms'] / 1e3:.0f} Mrays/s all segments, crc {crc[label]}", flush=True)
        st = ctx.render(w, h, sc.camera, mode=2, spp=min(spp, 4), max_bounces=bounces, counters=True, **kw)
        g = ctx.debug_shadow_grid()
        print(f"   counted: nodes/seg {st['node_visits'] / st['rays']:.2f} tris/seg {st['tri_tests'] / st['rays']:.2f} shadow segs {st['shadow_rays'] / 1e6:.1f} M, "
              f"answered by grids {g['segments_answered'] / 1e6:.1f} M, entries read per answered {g['entries_read'] / max(1, g['segments_answered']):.2f}")
    print("SAME FRAME" if crc["bvh"] == crc["grid"] else "FRAMES DIFFER")
