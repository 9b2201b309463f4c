"""N > 1 path on CPU: world_size-2 and -3 gloo runs of bench.py's multi-process glue (tile partition, barrier,
max-over-ranks, assembly of the ranks' tiles into one frame + its CRC) and the tile-ownership rule shared by bench.py and rt_render's tile_rank/tile_world."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_bench_glue_gloo(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--selftest-cpu"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout  # rank 0 prints ONE JSON line
    j = json.loads(lines[0])
    assert j["n_ranks"] == world and j["tiles_total"] == j["tiles_expected"] == 60 * 34  # 1080p in the 32x32 tiles bench.py deals over several ranks
    assert j["max_dt"] >= 0.01 * world  # the MAX over ranks, not rank 0's own time
    # SURVEY 8e gather: the ranks' disjoint tiles assembled in one shared host framebuffer give the frame a single
    # process would have produced (same CRC as bench.py's N = 1 line prints for the same frame)
    assert j["frame_equal"] and j["frame_crc"] == j["frame_crc_expected"]


def test_frame_assembly_single_process_crc_matches():
    import zlib
    import bench
    full = bench.synthetic_frame(200, 90)
    crc, img = bench.assemble_frame(None, 0, 1, full, 200, 90, 32)
    assert crc == zlib.crc32(full.tobytes()) & 0xFFFFFFFF and img is not None


def test_tile_ownership_is_a_partition():
    import bench
    for (w, h) in [(1920, 1080), (3840, 2160), (200, 100), (128, 128)]:
        for world in (1, 2, 4, 8):
            seen = []
            for r in range(world):
                tiles, tx, ty = bench.owned_tiles(w, h, 128, r, world)
                seen += tiles
            assert sorted(seen) == list(range(tx * ty))
    tiles, tx, ty = bench.owned_tiles(1920, 1080, 128, 3, 8)
    assert (tx, ty) == (15, 9) and tiles[:3] == [3, 11, 19]
