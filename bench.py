#!/usr/bin/env python3
"""bench.py — Mrays/s of the ray-casting hot path on N MI355X (one process per GPU).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one frame of synthetic input: the sponza-like scene
(262,144 triangles, seed 0x53504F4E) at 1920x1080 (BASELINE.json's headline workload), scene already
resident in HBM.  With N ranks the frame's 128x128 tiles are interleaved over the ranks (tile i ->
rank i mod N, scene replicated, no data-path collective); torch.distributed is used only for the
barrier and the max-over-ranks of the timed region.  Rank 0 prints ONE JSON line.

There is no CPU fallback: without librt_hip.so or a HIP device this script fails.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
S_OUT_BYTES = 16 + 12 + 8  # per pixel: RGBA32F + three unorm8 texels + (prim id, t) hit record
S_STATE_BYTES = 2 * 76  # per segment through the wavefront queues: one WavefrontRay-sized record written and read


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
    ap.add_argument("--cpu-sample", default=None, help="resolution of the bounded CPU-baseline sample (default 960x540 "
                    "for the reference mode, 160x90 at <= 4 spp for the extended mode)")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="exercise only the multi-process glue (tile partition, barrier, max-reduce) on CPU/gloo; renders nothing")
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


def max_over_ranks(dist, value, device):
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, value, device):
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def barrier_sync(dist, torch_cuda):
    if dist is not None:
        dist.barrier()
    if torch_cuda is not None:
        torch_cuda.synchronize()


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
    extended mode: the CPU statement of the extended mode (same reference-format BVH traversal per segment)."""
    import oracle
    w, h = (int(v) for v in sample.split("x"))
    cores = available_cpus()
    packed = oracle.PackedScene(scene)
    t0 = time.perf_counter()
    if mode_name == "extended":
        sspp = min(spp, 4)
        r = oracle.render_extended(packed, w, h, sspp, bounces, camera=camera, threads=cores)
        rays = sum(r["segments"][k] for k in ("camera", "continuation", "shadow"))
        what = f"{sspp} spp {bounces} bounces extended mode"
    else:
        # 32x32 work items so every thread stays busy; the image does not depend on the tile size
        r = oracle.render_frame(packed, w, h, camera=camera, mode=1, threads=cores, want_rgba8=False, tile_size=32)
        rays = r["counters"]["rays"]
        what = "1 spp primary rays"
    dt = time.perf_counter() - t0
    out = {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
           "sample": f"{scene.name} {w}x{h} {what}, same camera, reference-format BVH ({len(packed.nodes)} nodes, no t-culling), "
                     f"{rays} segments in {dt:.2f} s",
           "nodes_per_ray": r["counters"]["node_visits"] / max(r["counters"]["rays"], 1),
           "tris_per_ray": r["counters"]["tri_tests"] / max(r["counters"]["rays"], 1)}
    # second flavour (SURVEY 8d ii): what the same cores do with a decent traversal - one triangle per leaf, children
    # slab-tested against the closest hit, near child first, any-hit shadow segments.  Not the reference's algorithm
    # (tests/test_oracle_extended.py shows it renders the same frames); a larger sample because it is ~100x faster.
    try:
        fw, fh = w * 4, h * 4
        fast = oracle.PackedScene(scene, bvh=oracle.build_bvh(scene.triangles, scene.vertices, per_triangle=True))
        oracle.set_fast_traversal(True)
        t0 = time.perf_counter()
        if mode_name == "extended":
            r2 = oracle.render_extended(fast, fw, fh, sspp, bounces, camera=camera, threads=cores)
            rays2 = sum(r2["segments"][k] for k in ("camera", "continuation", "shadow"))
        else:
            r2 = oracle.render_frame(fast, fw, fh, camera=camera, mode=1, threads=cores, want_rgba8=False, tile_size=32)
            rays2 = r2["counters"]["rays"]
        dt2 = time.perf_counter() - t0
        out["culled_traversal"] = {"value": rays2 / dt2 / 1e6, "unit": "Mrays/s", "cores": cores,
                                   "sample": f"{fw}x{fh}, {len(fast.nodes)}-node one-triangle-per-leaf BVH, ordered + distance-culled traversal, "
                                             f"{rays2} segments in {dt2:.2f} s",
                                   "nodes_per_ray": r2["counters"]["node_visits"] / max(r2["counters"]["rays"], 1),
                                   "tris_per_ray": r2["counters"]["tri_tests"] / max(r2["counters"]["rays"], 1)}
    finally:
        oracle.set_fast_traversal(False)
    return out


def load_traffic(workload):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/traffic.json), or None."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get(workload)
        except Exception:
            return None
    return None


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
        if rank == 0:
            print(json.dumps({"selftest": True, "n_ranks": n_gpus, "tiles_total": int(total), "tiles_expected": tx * ty,
                              "max_dt": dt}))
        if dist is not None:
            dist.destroy_process_group()
        return

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    local_rank = int(os.environ.get("RT_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from gpu_raytracer_amd import api, scenes
    scene = scenes.SCENES[args.scene]()
    mode_name = args.mode
    if mode_name == "auto":
        mode_name = "extended" if hasattr(api, "EXTENDED_AVAILABLE") and api.EXTENDED_AVAILABLE else "reference"
    mode = api.MODE_EXTENDED if mode_name == "extended" else api.MODE_WAVEFRONT
    spp = args.spp if mode_name == "extended" else 1
    bounces = args.bounces

    ctx = api.Context((local_rank,))
    ctx.upload_scene(scene)  # scene resident in HBM before the timed region

    def step(counters=False):
        return ctx.render(args.width, args.height, scene.camera, mode=mode, spp=spp, max_bounces=bounces,
                          tile_size=tile, tile_rank=rank, tile_world=n_gpus, counters=counters,
                          kernel_sm=args.kernel == "state_machine", kernel_v1=args.kernel == "nested")

    for _ in range(args.warmup):
        step()
    barrier_sync(dist, torch.cuda)
    t0 = time.perf_counter()
    kernel_ms, rays = [], 0
    for _ in range(args.steps):
        st = step()
        kernel_ms.append(st["kernel_ms"])
        rays += st["rays"]
    barrier_sync(dist, torch.cuda)
    dt = max_over_ranks(dist, time.perf_counter() - t0, dev)
    total_rays = sum_over_ranks(dist, float(rays), dev)

    # algorithmic bytes of one launch on this rank: exact node / triangle fetch counts from the counting variant
    stc = step(counters=True)
    alg_bytes = stc["node_visits"] * stc["node_bytes"] + stc["tri_tests"] * stc["tri_bytes"] + stc["pixels"] * S_OUT_BYTES
    if mode_name == "extended" and args.kernel == "wavefront":
        alg_bytes += stc["rays"] * S_STATE_BYTES
    avg_kernel_ms = float(np.mean(kernel_ms))
    achieved = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9

    # PMC HBM bytes per launch were measured for the whole frame on one GPU (profiles/traffic.json); rank 0's share of a
    # partitioned frame is scaled by its share of the segments
    traffic = load_traffic(f"{scene.name}_{args.width}x{args.height}_{mode_name}")
    if traffic is not None and n_gpus > 1 and total_rays > 0:
        traffic = traffic * stc["rays"] / (total_rays / args.steps)
    if rank == 0:
        workload = f"{scene.name} {scene.n_triangles} tris {args.width}x{args.height} {spp} spp " + \
                   (f"{bounces} bounces (extended mode)" if mode_name == "extended" else
                    "primary rays, reference semantics (mode 1: one pixel-centre ray per pixel; the reference has no spp/bounces)")
        out = {
            "metric": "Mrays/s", "value": total_rays / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "scene": scene.name, "triangles": scene.n_triangles,
                       "resolution": [args.width, args.height], "spp": spp, "mode": mode_name,
                       "implementation": args.kernel if mode_name == "extended" else "k_render_reference",
                       "partition": f"tiles {tile}x{tile} interleaved over {n_gpus} rank(s), scene replicated, no collective",
                       "bounces": bounces if mode_name == "extended" else 0,
                       "rays_per_step": total_rays / args.steps,
                       "mrays_per_s_camera_and_continuation_only": total_rays / dt / 1e6 * (stc["primary_rays"] + stc["continuation_rays"]) / max(stc["rays"], 1),
                       "segments_rank0": {"camera": stc["primary_rays"], "continuation": stc["continuation_rays"], "shadow": stc["shadow_rays"]},
                       "kernel_mrays_per_s_rank0": stc["rays"] / avg_kernel_ms / 1e3},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_render_reference" if mode_name == "reference" else
                                   "wavefront pipeline (k_wf_trace<shadow> + k_wf_trace<closest> + k_wf_shade + k_wf_finish + k_wf_generate)",
                         "kernel_avg_ms": avg_kernel_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "nodes_per_ray": stc["node_visits"] / max(stc["rays"], 1), "tris_per_ray": stc["tri_tests"] / max(stc["rays"], 1),
                         "note": "algorithmic bytes = node fetches x 48 B + triangle fetches x 48 B + pixels x 36 B + (wavefront) segments x 152 B of "
                                 "queue state (rank 0's share); kernel_avg_ms is the HIP-event time of all stage kernels of one frame; "
                                 "the 23 MB scene is cache resident, so this exceeds what HBM itself moves (see traffic)"},
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
