"""The multi-rank path on CPU: world_size 2 and 3 over gloo.  Every rank packs its share of a known frame exactly as
rt_render_tiles lays it out, the tiles are gathered to rank 0 through the same TileGather bench.py uses over RCCL,
and rank 0 reassembles the frame.  Also checks that the library's tile arithmetic agrees with the host restatement."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import capi, tiles


def known_frame(w, h):
    y, x = np.mgrid[0:h, 0:w]
    return np.stack([(x * 7 + y * 3) % 251, (x ^ y) % 256, (x * y + 11) % 253], axis=-1).astype(np.uint8)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def worker(rank, world, port, w, h, tile, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frame = known_frame(w, h)
        tg = tiles.TileGather(torch, dist, w, h, tile, world, rank, "cpu")
        for step in range(2):                       # buffers are reused frame after frame
            tg.local.copy_(torch.from_numpy(tiles.pack_rank_tiles((frame + step).astype(np.uint8), tile, world, rank)))
            dist.barrier()
            gathered = tg.gather()
            if rank == 0:
                got = tiles.assemble(gathered.numpy(), w, h, tile, world)
                assert np.array_equal(got, (frame + step).astype(np.uint8))
        # the timing reduction of bench.py: MAX over ranks
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == world
        if rank == 0:
            open(out_path, "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h,tile", [(2, 200, 120, 32), (3, 97, 65, 16), (2, 64, 64, 64)])
def test_tiles_gather_over_gloo(tmp_path, world, w, h, tile):
    out = str(tmp_path / "ok.txt")
    mp.spawn(worker, args=(world, free_port(), w, h, tile, out), nprocs=world, join=True)
    assert open(out).read() == "ok"


@pytest.mark.parametrize("w,h,tile,n", [(3840, 2160, 64, 8), (200, 120, 32, 2), (97, 65, 16, 3), (64, 64, 64, 2), (17, 5, 16, 4)])
def test_library_tile_arithmetic_matches(w, h, tile, n):
    t = capi.Tiling(w, h, tile, n)
    assert capi.tile_count(t) == tiles.tile_count(w, h, tile)
    assert capi.tiles_per_rank(t) == tiles.tiles_per_rank(w, h, tile, n)
    owned = [tiles.rank_tiles(w, h, tile, n, r) for r in range(n)]
    assert sorted(sum(owned, [])) == list(range(tiles.tile_count(w, h, tile)))       # a partition: every tile exactly once
    assert max(map(len, owned)) - min(map(len, owned)) <= 1                          # balanced to within one tile


def test_pack_then_assemble_is_identity():
    for w, h, tile, n in [(50, 30, 16, 3), (64, 48, 16, 1), (33, 33, 32, 5)]:
        frame = known_frame(w, h)
        gathered = np.concatenate([tiles.pack_rank_tiles(frame, tile, n, r) for r in range(n)])
        assert np.array_equal(tiles.assemble(gathered, w, h, tile, n), frame)


def gpu_worker(rank, world, port, out_path):
    """One process per rank, all on GPU 0: render this rank's tiles with the real kernels, gather, assemble."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracer_amd import scenes as S
        prims, lights, camd = S.random_scene(90, 6, lights=2)
        W, H, tile = 232, 120, 32
        cam = capi.make_camera(**dict(camd, res=(W, H)))
        ctx = capi.Context(0)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.upload(capi.Scene.create(prims, lights))
        tiling = capi.Tiling(W, H, tile, world)
        tg = tiles.TileGather(torch, dist, W, H, tile, world, rank, "cuda")
        for ms in (0, 2):
            ctx.render_tiles_dev(cam, tiling, rank, tg.local.data_ptr(), ms, 0.0)
            torch.cuda.synchronize()
            gathered = tg.gather()
            if rank == 0:
                frame = torch.empty(H * W * 3, dtype=torch.uint8, device="cuda")
                ctx.assemble_tiles_dev(tiling, gathered.data_ptr(), frame.data_ptr())
                torch.cuda.synchronize()
                ctx.use_own_stream()
                want = ctx.render(cam, max_scatters=ms)
                ctx.set_stream(torch.cuda.current_stream().cuda_stream)
                assert np.array_equal(frame.cpu().numpy().reshape(H, W, 3), want), ms
            dist.barrier()
        ctx.close()
        if rank == 0:
            open(out_path, "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_ranks_render_gather_assemble_on_the_gpu(tmp_path, world):
    out = str(tmp_path / "ok.txt")
    mp.spawn(gpu_worker, args=(world, free_port(), out), nprocs=world, join=True)
    assert open(out).read() == "ok"


def test_missing_rccl_is_an_error_not_a_crash():
    """Without a loadable RCCL the exchange entry points return RT_ERR_COMM (-9) with a message; everything else keeps
    working.  The library binds RCCL once per process, so the miss is forced in a child process."""
    import subprocess
    import sys
    code = ("import ctypes as C, sys\n"
            "from raytracer_amd import capi\n"
            "L = capi.lib()\n"
            "buf = (C.c_uint8 * 128)()\n"
            "rc = L.rt_comm_unique_id(buf)\n"
            "msg = L.rt_last_error().decode()\n"
            "assert rc == -9, rc\n"
            "assert 'RCCL not found' in msg, msg\n"
            "rc2 = L.rt_comm_unique_id(buf)\n"          # asked again: the same answer, still no crash
            "assert rc2 == -9, rc2\n"
            "lut = (C.c_uint8 * 256)()\n"
            "assert L.rt_gamma_lut(lut) == 0 and lut[255] == 255\n"
            "print('ok')\n")
    env = dict(os.environ, RT_RCCL_LIBRARY="/nonexistent/librccl_missing.so")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.returncode, r.stdout, r.stderr)
