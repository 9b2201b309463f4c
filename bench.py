#!/usr/bin/env python3
"""bench.py — Mrays/s of the ray-casting hot path on N MI355X (one process per GPU).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one frame of synthetic input: the sponza-like scene (262,144 triangles,
seed 0x53504F4E) at 1920x1080, 64 spp (BASELINE.json's headline workload), scene already resident in HBM.  With N
ranks the frame's 32x32 tiles are interleaved over the ranks (tile i -> rank i mod N, scene replicated, no data-path
collective); torch.distributed carries only the barrier and the max / sum of a few scalars.  After the timed region
every rank copies its tiles into one shared host framebuffer and rank 0 prints the CRC of the assembled float image:
it is the same number for every N (a pixel does not depend on the partition).  Rank 0 prints ONE JSON line.

`value` follows the written metric: a ray is one traced segment, primary (camera) or
continuation.  The extended mode also traces shadow segments toward every light with a non-zero contribution; they
are real any-hit BVH walks and are reported beside it (`all_segments_mrays_per_s`), not inside it.

There is no CPU fallback: without librt_hip.so or a HIP device this script fails.
"""
import argparse
import hashlib
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HEADLINE_FRAME_CRC = 4012668657  # CRC-32 of the float image of the default workload (sponza-like 1920x1080, 64 spp, 4 bounces): asserted against the
                                 # oracle's crops in tests/test_gpu_baseline_configs.py and printed beside the measured one (config.frame_crc_expected)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
S_OUT_BYTES = 16 + 12 + 8  # per pixel: RGBA32F + three unorm8 texels + (prim id, t) hit record

# Path-state bytes the queue pipeline moves per segment KIND (gpu_raytracer_amd/csrc/wavefront.hip; 16-byte records):
#   extension segment (camera or continuation): written by generate/finish: ray_o, ray_d, thr, rad (64) + queue id (4);
#     read by the closest-hit walk: id (4) + ray_o, ray_d (32), writes the hit record (16); the shade stage reads
#     id + hit (20) and writes the vertex record (32); the finish stage reads id + the vertex record, thr, rad, ray_d (84)
#   shadow segment walked through the BVH (no light grids, or handed on by them): queue entry written and read (8) + the vertex record
#     read (32) + the visibility word's update (atomic, 4)
#   the light-grid stage goes by vertex, not by segment: per extension segment the queue id (4) + the vertex record (32) read once for
#     all its lights + the visibility word's update (4)
#   path: the sample's radiance written once and read once by the resolve (32) + pxy (4)
S_STATE_EXTENSION = 64 + 4 + 4 + 32 + 16 + 20 + 32 + 84
S_STATE_SHADOW = 8 + 32 + 4
S_STATE_SHADOW_VERTEX = 4 + 32 + 4
S_STATE_PATH = 32 + 4


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="sponza_like", choices=["sponza_like", "bistro_like", "cornell12"])
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--mode", default="auto", choices=["auto", "reference", "extended"])
    ap.add_argument("--kernel", default="wavefront", choices=["wavefront", "state_machine", "nested"],
                    help="extended-mode implementation (all three produce identical images)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--device-tree", action="store_true", help="render on the tree rt_upload_scene builds on the device (10 ms) instead of the host builder's (rt_prepare)")
    ap.add_argument("--cpu-sample", default=None, help="resolution of the bounded CPU-baseline sample (default 960x540 "
                    "for the reference mode, 160x90 at <= 4 spp for the extended mode)")
    ap.add_argument("--share-of", type=int, default=0, metavar="N",
                    help="development aid (profiles of the 8-GPU configurations on one GPU): render only rank 0's share of an N-way "
                         "tile partition in this single process; the line is labelled as such and is not a scaling result")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="exercise only the multi-process glue (tile partition, barrier, max-reduce, frame assembly + CRC) on "
                         "CPU/gloo with a synthetic frame; renders nothing")
    return ap.parse_args()


def init_dist(selftest):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # RT_BENCH_BACKEND=gloo + RT_BENCH_DEVICE=0: rehearsal of the multi-rank path on a box with fewer GPUs than ranks
        backend = "gloo" if selftest else os.environ.get("RT_BENCH_BACKEND", "nccl")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist, world, rank, local_rank


def owned_tiles(width, height, tile, rank, world):
    """Row-major tile indices this rank renders: tile i -> rank i mod N (src/compute.rs:199-200 ordering)."""
    tx, ty = (width + tile - 1) // tile, (height + tile - 1) // tile
    return list(range(rank, tx * ty, world)), tx, ty


def _reduce(dist, value, device, op):
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=getattr(dist.ReduceOp, op))
    return float(t.item())


def max_over_ranks(dist, value, device):
    return _reduce(dist, value, device, "MAX")


def sum_over_ranks(dist, value, device):
    return _reduce(dist, value, device, "SUM")


def barrier_sync(dist, torch_cuda):
    if dist is not None:
        dist.barrier()
    if torch_cuda is not None:
        torch_cuda.synchronize()


def assemble_frame(dist, rank, world, local_frame, width, height, tile):
    """SURVEY §8e "gather": every rank copies its disjoint tiles (D2H already done: local_frame is this rank's read-back,
    zero outside its tiles) into ONE host framebuffer shared by the ranks of the node (a /dev/shm mapping named after
    the rendezvous port); rank 0 returns (crc32 of the float image, the image), the others (None, None).  No collective
    moves pixels."""
    frame_bytes = height * width * 3 * 4
    if dist is None:
        img = np.ascontiguousarray(local_frame, dtype=np.float32)
        return zlib.crc32(img.tobytes()) & 0xFFFFFFFF, img
    path = f"/dev/shm/rt_bench_fb_{os.environ.get('MASTER_PORT', '0')}_{os.getuid()}"
    if rank == 0:
        with open(path, "wb") as f:
            f.truncate(frame_bytes)
    dist.barrier()
    fb = np.memmap(path, dtype=np.float32, mode="r+", shape=(height, width, 3))
    tiles, tx, _ = owned_tiles(width, height, tile, rank, world)
    for t in tiles:
        ox, oy = (t % tx) * tile, (t // tx) * tile
        fb[oy:oy + tile, ox:ox + tile] = local_frame[oy:oy + tile, ox:ox + tile]
    fb.flush()
    dist.barrier()
    out = (None, None)
    if rank == 0:
        img = np.array(fb)
        out = (zlib.crc32(img.tobytes()) & 0xFFFFFFFF, img)
    del fb
    dist.barrier()
    if rank == 0:
        os.unlink(path)
    return out


def synthetic_frame(width, height):
    """A frame that is a pure function of the pixel coordinates (the --selftest-cpu stand-in for a rendered one)."""
    y, x = np.mgrid[0:height, 0:width].astype(np.float32)
    return np.stack([x * 0.25 + y, x - y * 0.5, (x * 7.0 + y * 13.0) % 31.0], axis=-1).astype(np.float32)


def available_cpus():
    """CPUs this process may actually use: cgroup quota if set, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
        return max(1, n)
    except Exception:
        pass
    # no quota visible: a GPU box exposes every host CPU but grants a share of 16 per GPU
    return max(1, min(n, int(os.environ.get("RT_BENCH_CPU_THREADS", "16"))))


def cpu_baseline(scene, camera, sample, mode_name, spp, bounces):
    """The oracle timed on this host's cores on a BOUNDED sample of the same workload.
    reference mode: the CPU restatement of the reference kernel walking the reference-format BVH (chunked
    mesh-order leaves, no t-culling: what the reference's kernel executes per ray).
    extended mode: the CPU statement of the extended mode (same reference-format BVH traversal per segment).
    `value` counts camera + continuation segments, like the headline value."""
    import oracle
    w, h = (int(v) for v in sample.split("x"))
    cores = available_cpus()
    packed = oracle.PackedScene(scene)

    def run(p, ww, hh):
        t0 = time.perf_counter()
        if mode_name == "extended":
            r = oracle.render_extended(p, ww, hh, min(spp, 4), bounces, camera=camera, threads=cores)
            seg = r["segments"]
            metric, allseg = seg["camera"] + seg["continuation"], seg["camera"] + seg["continuation"] + seg["shadow"]
        else:
            # 32x32 work items so every thread stays busy; the image does not depend on the tile size
            r = oracle.render_frame(p, ww, hh, camera=camera, mode=1, threads=cores, want_rgba8=False, tile_size=32)
            metric = allseg = r["counters"]["rays"]
        return r, metric, allseg, time.perf_counter() - t0

    r, metric, allseg, dt = run(packed, w, h)
    what = f"{min(spp, 4)} spp {bounces} bounces extended mode" if mode_name == "extended" else "1 spp primary rays"
    out = {"value": metric / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
           "all_segments_mrays_per_s": allseg / dt / 1e6,
           "sample": f"{scene.name} {w}x{h} {what}, same camera, reference-format BVH ({len(packed.nodes)} nodes, no t-culling), "
                     f"{metric} camera+continuation of {allseg} segments in {dt:.2f} s",
           "nodes_per_ray": r["counters"]["node_visits"] / max(r["counters"]["rays"], 1),
           "tris_per_ray": r["counters"]["tri_tests"] / max(r["counters"]["rays"], 1)}
    # second flavour (SURVEY 8d ii): what the same cores do with a decent traversal - one triangle per leaf, children
    # slab-tested against the closest hit, near child first, any-hit shadow segments.  Not the reference's algorithm
    # (tests/test_oracle_extended.py shows it renders the same frames); a larger sample because it is ~100x faster.
    try:
        fast = oracle.PackedScene(scene, bvh=oracle.build_bvh(scene.triangles, scene.vertices, per_triangle=True))
        oracle.set_fast_traversal(True)
        r2, metric2, allseg2, dt2 = run(fast, w * 4, h * 4)
        out["culled_traversal"] = {"value": metric2 / dt2 / 1e6, "unit": "Mrays/s", "cores": cores,
                                   "all_segments_mrays_per_s": allseg2 / dt2 / 1e6,
                                   "sample": f"{w * 4}x{h * 4}, {len(fast.nodes)}-node one-triangle-per-leaf BVH, ordered + distance-culled "
                                             f"traversal, {metric2} camera+continuation of {allseg2} segments in {dt2:.2f} s",
                                   "nodes_per_ray": r2["counters"]["node_visits"] / max(r2["counters"]["rays"], 1),
                                   "tris_per_ray": r2["counters"]["tri_tests"] / max(r2["counters"]["rays"], 1)}
    finally:
        oracle.set_fast_traversal(False)
    return out


def kernel_source_sha16():
    """Identity of the device code a committed counter profile belongs to: sha256 over the sources librt_hip.so is built from."""
    csrc = os.path.join(ROOT, "gpu_raytracer_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(csrc)):
        p = os.path.join(csrc, name)
        if os.path.isfile(p) and name.endswith((".hip", ".h", ".cpp")):
            h.update(name.encode())
            h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def load_profile(workload):
    """The committed rocprofv3 --pmc summary for this workload (profiles/traffic.json, written by scripts/summarize_profile.py)
    if it was taken from THIS build of the kernels, else None."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        e = json.load(open(p)).get(workload)
    except Exception:
        return None
    if not isinstance(e, dict) or e.get("source_sha16") != kernel_source_sha16():
        return None
    return e


def main():
    args = parse_args()
    dist, world, rank, local_rank = init_dist(args.selftest_cpu)
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    n_gpus = max(world, 1)
    # 32x32 tiles (finer tiles so that every rank gets hundreds of them; measured per-rank imbalance at 8 ranks
    # 2.0 % -> 0.6 %).  On one GPU the tile size only orders the work; 16..64-pixel tiles measured 2 % faster than the
    # reference's 128 at 1080p (scripts/tile_size_probe.py).  The image does not depend on it (bit-identical for 8..256).
    tile = 32

    if args.selftest_cpu:
        tiles, tx, ty = owned_tiles(args.width, args.height, tile, rank, n_gpus)
        barrier_sync(dist, None)
        t0 = time.perf_counter()
        time.sleep(0.01 * (rank + 1))
        barrier_sync(dist, None)
        dt = max_over_ranks(dist, time.perf_counter() - t0, "cpu")
        total = sum_over_ranks(dist, float(len(tiles)), "cpu")
        # frame assembly on CPU buffers: this rank "rendered" the synthetic frame on its own tiles only
        full = synthetic_frame(args.width, args.height)
        mine = np.zeros_like(full)
        for t in tiles:
            ox, oy = (t % tx) * tile, (t // tx) * tile
            mine[oy:oy + tile, ox:ox + tile] = full[oy:oy + tile, ox:ox + tile]
        crc, img = assemble_frame(dist, rank, n_gpus, mine, args.width, args.height, tile)
        if rank == 0:
            print(json.dumps({"selftest": True, "n_ranks": n_gpus, "tiles_total": int(total), "tiles_expected": tx * ty,
                              "max_dt": dt, "frame_crc": crc, "frame_crc_expected": zlib.crc32(full.tobytes()) & 0xFFFFFFFF,
                              "frame_equal": bool(np.array_equal(img, full))}))
        if dist is not None:
            dist.destroy_process_group()
        return

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    local_rank = int(os.environ.get("RT_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if dist is None or dist.get_backend() != "gloo" else "cpu"

    from gpu_raytracer_amd import api, scenes
    scene = scenes.SCENES[args.scene]()
    mode_name = args.mode
    if mode_name == "auto":
        mode_name = "extended" if hasattr(api, "EXTENDED_AVAILABLE") and api.EXTENDED_AVAILABLE else "reference"
    mode = api.MODE_EXTENDED if mode_name == "extended" else api.MODE_WAVEFRONT
    spp = args.spp if mode_name == "extended" else 1
    bounces = args.bounces

    ctx = api.Context((local_rank,))
    ctx.upload_scene(scene)  # scene resident in HBM before the timed region (tree built on the device in milliseconds)
    # the scene stays for every step: the host builder's tree (rt_prepare RT_PREPARE_QUALITY_TREE: binned SAH + insertion-based optimisation,
    # 0.35 s for this scene, frames 2 % faster than on the device-built tree, same image) and, for the extended mode, the light grids of its
    # shadow stage (built lazily by the first frame otherwise).  --device-tree keeps the tree rt_upload_scene built.
    ctx.prepare((0 if args.device_tree else api.PREPARE_QUALITY_TREE) | (api.PREPARE_SHADOW_GRIDS if mode_name == "extended" else 0))

    part_world = args.share_of if (args.share_of > 1 and n_gpus == 1) else n_gpus

    def step(counters=False, stage_times=False):
        return ctx.render(args.width, args.height, scene.camera, mode=mode, spp=spp, max_bounces=bounces,
                          tile_size=tile, tile_rank=rank, tile_world=part_world, counters=counters, stage_times=stage_times,
                          kernel_sm=args.kernel == "state_machine", kernel_v1=args.kernel == "nested")

    for _ in range(args.warmup):
        step()
    barrier_sync(dist, torch.cuda)
    t0 = time.perf_counter()
    kernel_ms, seg = [], np.zeros(3)
    for _ in range(args.steps):
        st = step()
        kernel_ms.append(st["kernel_ms"])
        seg += (st["primary_rays"], st["continuation_rays"], st["shadow_rays"])
    barrier_sync(dist, torch.cuda)
    dt = max_over_ranks(dist, time.perf_counter() - t0, dev)
    total_seg = [sum_over_ranks(dist, float(v), dev) for v in seg]  # camera, continuation, shadow over all ranks and steps
    metric_rays, all_rays = total_seg[0] + total_seg[1], sum(total_seg)

    # SURVEY §8e gather, outside the timed region: D2H of this rank's tiles, then one shared host framebuffer
    barrier_sync(dist, torch.cuda)
    g0 = time.perf_counter()
    local_frame = ctx.read_rgb32f()  # device epilogue (pack to rgb32f) + D2H through pinned staging + the copy into the caller's array
    d2h_ms = max_over_ranks(dist, (time.perf_counter() - g0) * 1e3, dev)
    g1 = time.perf_counter()
    crc, _img = assemble_frame(dist, rank, n_gpus, local_frame, args.width, args.height, tile)
    if part_world != n_gpus:
        crc = None  # a share of the frame, not the frame
    crc_ms = max_over_ranks(dist, (time.perf_counter() - g1) * 1e3, dev)  # shared-framebuffer copies (N > 1) + zlib.crc32 over the float image on rank 0
    gather_ms = d2h_ms + crc_ms

    # exact node / triangle fetch counts and wave-level step statistics from the counting variant of the same kernels
    # the same frame on one lane (RT_WF_LANES=1: every stage kernel alone on the chip) - what the rocprofv3 summaries under profiles/ time
    one_lane_ms, grid_ms, grid_launches = None, 0.0, 0
    if mode_name == "extended" and os.environ.get("RT_WF_LANES") is None:
        os.environ["RT_WF_LANES"] = "1"
        try:
            step()
            one_lane, grid_ms, grid_launches = [], 0.0, 0
            for _ in range(2):
                one_lane.append(step(stage_times=args.kernel == "wavefront")["kernel_ms"])
                ms, n = ctx.debug_stage_times()  # every k_wf_shadow_grid launch of the frame, HIP events on the stream it was launched on
                grid_ms, grid_launches = grid_ms + ms, grid_launches + n
            one_lane_ms = float(np.mean(one_lane))
        finally:
            del os.environ["RT_WF_LANES"]
        step()
    stc = step(counters=True)
    diag = list(ctx.debug_counters().values())
    grid = ctx.debug_shadow_grid()  # the per-light triangle lists of the shadow stage (csrc/shadow_grid.h): what they hold and answered
    avg_kernel_ms = float(np.mean(kernel_ms))
    ext_segments = stc["primary_rays"] + stc["continuation_rays"]
    wavefront = mode_name == "extended" and args.kernel == "wavefront"
    # (a triangle tested from a light's list is a 48-byte entry instead of a 48-byte record: the same figure; its cell costs 8 bytes more)
    fetch_bytes = stc["node_visits"] * stc["node_bytes"] + stc["tri_tests"] * stc["tri_bytes"] + grid["segments_answered"] * 8
    grids_on = wavefront and grid["lights_with_grid"] > 0 and grid["segments_answered"] > 0
    shadow_state = (ext_segments * S_STATE_SHADOW_VERTEX + (stc["shadow_rays"] - grid["segments_answered"]) * S_STATE_SHADOW) if grids_on else stc["shadow_rays"] * S_STATE_SHADOW
    state_bytes = (ext_segments * S_STATE_EXTENSION + shadow_state + stc["primary_rays"] * S_STATE_PATH) if wavefront else 0
    # SURVEY §8d's algorithmic bytes: every record fetch + the path state + the pixels.  The scene (23 MB) is cache
    # resident, so most of this never reaches HBM: it is reported as a rate, not as a fraction of the HBM peak.
    alg_bytes = fetch_bytes + state_bytes + stc["pixels"] * S_OUT_BYTES
    # bytes that have to cross the HBM interface whatever the caches do: the path state, the pixels, the scene once
    compulsory_bytes = state_bytes + stc["pixels"] * S_OUT_BYTES + stc["scene_bytes"]

    headline = (args.scene, args.width, args.height, spp, bounces, mode_name) == ("sponza_like", 1920, 1080, 64, 4, "extended") and part_world == n_gpus
    workload_key = f"{scene.name}_{args.width}x{args.height}_{mode_name}" + (f"_{spp}spp_{bounces}b" if mode_name == "extended" else "") + \
                   (f"_share{part_world}" if part_world != n_gpus else "")
    prof = load_profile(workload_key)
    traffic = prof["traffic"] if prof else None
    if traffic is not None and n_gpus > 1 and all_rays > 0:  # measured for the whole frame on one GPU: rank 0's share by segments
        traffic = traffic * stc["rays"] / (all_rays / args.steps)
    if rank == 0:
        workload = f"{scene.name} {scene.n_triangles} tris {args.width}x{args.height} {spp} spp " + \
                   (f"{bounces} bounces (extended mode)" if mode_name == "extended" else
                    "primary rays, reference semantics (mode 1: one pixel-centre ray per pixel; the reference has no spp/bounces)")
        frame = {  # the whole pipeline, one frame (all stage kernels, HIP events on the launch stream)
            "kernel_avg_ms": avg_kernel_ms, "kernel_avg_ms_one_lane": one_lane_ms,
            "traffic": traffic, "traffic_source": f"profiles/traffic.json@{prof['source_sha16']} ({prof.get('profile', '')}): per-kernel FETCH_SIZE x read factor + WRITE_SIZE" if prof else None,
            "hbm_GBps": (traffic / (avg_kernel_ms * 1e-3) / 1e9) if traffic is not None else None,
            "hbm_frac": (traffic / (avg_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic is not None else None,
            "compulsory_hbm_bytes": compulsory_bytes,
            "algorithmic_bytes": alg_bytes, "algorithmic_GBps": alg_bytes / (avg_kernel_ms * 1e-3) / 1e9,
            "state_bytes_per_segment": {"extension": S_STATE_EXTENSION, "shadow_through_bvh": S_STATE_SHADOW, "light_grid_stage_per_extension_segment": S_STATE_SHADOW_VERTEX,
                                        "path": S_STATE_PATH} if wavefront else None,
            "nodes_per_ray": stc["node_visits"] / max(stc["rays"], 1), "tris_per_ray": stc["tri_tests"] / max(stc["rays"], 1),
            "note": "algorithmic_*: SURVEY 8d's per-segment figure over every kernel (node fetches x 80 B + triangle fetches x 48 B + per-kind path state + "
                    "pixels x 36 B); ~90 % of it is served by L1/L2 (the scene is 28 MB), so it is a rate, not an HBM fraction",
        }
        dom = prof.get("dominant_kernel") if prof else None
        if wavefront and grid_launches and grid["lights_with_grid"]:
            # SURVEY 8d / DESIGN 4 for the dominant kernel k_wf_shadow_grid, per launch: every extension segment's queue id and vertex record are
            # read once and its visibility word updated (40 B), every segment the lists answer reads its cell's header (8 B) and 48 B per list
            # entry it looks at, every segment handed on is written to the next queue (4 B)
            launches_per_frame = grid_launches / 2.0
            alg_grid = (grid["entries_read"] * 48 + grid["segments_answered"] * 8 + ext_segments * S_STATE_SHADOW_VERTEX +
                        (stc["shadow_rays"] - grid["segments_answered"]) * 4) / launches_per_frame
            launch_ms = grid_ms / grid_launches
            achieved = alg_grid / (launch_ms * 1e-3) / 1e9
            dom_traffic = dom.get("hbm_bytes_per_launch") if dom and dom.get("name") == "k_wf_shadow_grid" else None
            roofline = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": dom_traffic,
                "kernel": "k_wf_shadow_grid", "launches_per_frame": launches_per_frame, "avg_launch_ms": launch_ms,
                "algorithmic_bytes_per_launch": alg_grid, "share_of_one_lane_frame": grid_ms / 2.0 / one_lane_ms if one_lane_ms else None,
                "traffic_over_algorithmic": (dom_traffic / alg_grid) if dom_traffic else None,
                "achieved_source": "algorithmic bytes per launch (counting variant of the same kernels, this run) / average duration of the kernel's launches, HIP events on "
                                   "their stream, frames on one lane (this run); traffic: FETCH_SIZE x 1 + WRITE_SIZE of the committed PMC passes of this build - its reads are "
                                   "scattered 16-byte quads, for which FETCH_SIZE counts requests of 64 bytes (profiles/r03_fetch_calibration.json)",
                "practical_bound": "the rate of scattered memory requests (37 G/s of 64 bytes against 24-43 G/s that a bare gather loop sustains, profiles/r03_fetch_calibration.json): "
                                   "every segment costs its cell's block and 0.5 further entries, none of it shared between neighbouring lanes beyond depth 0; fewer requests "
                                   "per segment (packed cell blocks), more requests in flight (the next light's block fetched while this one's is looked at) and more waves "
                                   "all measured no faster (profiles/ab_r03.json)",
                "read_requests_per_s": dom.get("read_requests_per_s") if dom else None,
                "shadow_grids": {"lights_with_grid": grid["lights_with_grid"], "bytes": grid["bytes"], "entries": grid["entries"],
                                 "shadow_segments_answered_share": grid["segments_answered"] / max(stc["shadow_rays"], 1),
                                 "entries_read_per_answered_segment": grid["entries_read"] / max(grid["segments_answered"], 1)},
                "second_kernel": {"name": "k_wf_trace<closest> (+ k_wf_trace_camera)", "bound": "VALU issue (branchy scalar f32, no MFMA; the tree is cache resident): see `issue`"},
                "frame": frame,
            }
        else:  # reference mode, or the megakernels: the frame is one kernel
            achieved_bytes = traffic if traffic is not None else compulsory_bytes
            achieved = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9
            roofline = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": "k_render_reference" if mode_name == "reference" else args.kernel, "avg_launch_ms": avg_kernel_ms,
                "algorithmic_bytes_per_launch": alg_bytes,
                "achieved_source": "algorithmic bytes (node fetches x 80 B + triangle fetches x 48 B + pixels x 36 B, counting variant of the same kernel) / kernel time (HIP events); "
                                   "the scene is cache resident, so this rate is served by L1 / L2, not by HBM: see hbm_bytes",
                "hbm_bytes": achieved_bytes, "hbm_frac": achieved_bytes / (avg_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "practical_bound": "VALU issue and memory latency of the per-lane tree walk (branchy scalar f32, no MFMA)",
                "frame": frame,
            }
        if wavefront and diag[3] > 0:
            roofline["issue"] = {
                "node_step_lane_utilisation": stc["node_visits"] / (diag[3] * 64.0),
                "triangle_step_lane_utilisation": diag[5] / max(diag[4] * 64.0, 1.0),
                "triangle_trips_per_step": diag[6] / max(diag[4], 1),
                "segments_per_refill_per_wave": stc["rays"] / max(diag[7], 1),
                "stack_high_water": diag[0],
                "valu_busy_from_profile": prof.get("valu_busy") if prof else None,
                "source": "RT_FLAG_COUNTERS variant of the same kernels (this run); valu_busy = 4 x SQ_ACTIVE_INST_VALU / (kernel time x "
                          "2.4 GHz x 1024 SIMDs) from the committed SQ pass",
            }
        if dom:
            roofline["profile_dominant_kernel"] = dom
        out = {
            "metric": "Mrays/s", "value": metric_rays / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "scene": scene.name, "triangles": scene.n_triangles,
                       "resolution": [args.width, args.height], "spp": spp, "mode": mode_name,
                       "implementation": args.kernel if mode_name == "extended" else "k_render_reference",
                       "tree": {0: "host builder: binned SAH + insertion-based optimisation (rt_prepare RT_PREPARE_QUALITY_TREE)", 1: "host PLOC",
                                2: "device build (rt_upload_scene)"}.get(stc.get("tree_build"), "?") + f", {stc['bvh_nodes']} 8-wide nodes, depth {stc['bvh_depth']}",
                       "partition": f"tiles {tile}x{tile} interleaved over {n_gpus} rank(s), scene replicated, no collective" if part_world == n_gpus else
                                    f"REHEARSAL: rank 0's share of a {part_world}-way partition in {tile}x{tile} tiles on one GPU (not a scaling result)",
                       "bounces": bounces if mode_name == "extended" else 0,
                       "ray_definition": "value counts camera + continuation segments (BASELINE.md 3); shadow segments are traced too and reported in all_segments_mrays_per_s",
                       "all_segments_mrays_per_s": all_rays / dt / 1e6,
                       "paths_per_s": total_seg[0] / dt,
                       "segments_per_step": {"camera": total_seg[0] / args.steps, "continuation": total_seg[1] / args.steps, "shadow": total_seg[2] / args.steps},
                       "kernel_mrays_per_s_rank0": ext_segments / avg_kernel_ms / 1e3,
                       "frame_crc": crc, "frame_crc_expected": HEADLINE_FRAME_CRC if headline else None,
                       "gather_ms": gather_ms, "d2h_ms": d2h_ms, "crc_ms": crc_ms,
                       "megakernel_fallback": bool(stc.get("flags", 0) & 1)},
            "roofline": roofline,
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            sample = args.cpu_sample or ("160x90" if mode_name == "extended" else "960x540")
            out["cpu_baseline"] = cpu_baseline(scene, scene.camera, sample, mode_name, spp, bounces)
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
