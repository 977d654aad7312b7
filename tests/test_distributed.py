"""CPU suite, part 3: the N>1 path (batch sharding + image gather) over gloo, world_size 2.

The HIP kernels need a GPU, so the per-rank "renderer" here is the CPU oracle: what is
under test is the host logic of pytorch_mesh_renderer_amd.distributed -- shard bounds,
uneven shards, the all-gather hand-over and the shared-mesh gradient all-reduce.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from pytorch_mesh_renderer_amd import distributed
from pytorch_mesh_renderer_amd.common import synthetic


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_render(shard, width, height):
    ids, bary, z = oracle.forward(shard["clip"].numpy(), shard["triangles"].numpy(), width, height)
    return torch.from_numpy(np.concatenate([bary, z[..., None]], -1))  # [B_local,H,W,4]


def _root_worker(rank, world, port, n_total, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    distributed.init_from_env(backend="gloo")
    job = synthetic.sphere_job(n_total, 24, 20, 6)
    shard = distributed.shard_batch(job, rank, world)
    handle = distributed.ImageGather(n_total, mode="root", dst=0)
    for _ in range(2):                       # two steps, waiting one step late
        handle.wait()                        # (nothing in flight the first time: None)
        handle.start(_oracle_render(shard, 24, 20))
    full = handle.wait()
    assert (full is not None) == (rank == 0)
    # the transform hook (bench.py hands over 8-bit frames): applied to the shard before the gather
    to_u8 = lambda t: (t.clamp(0, 1) * 255).to(torch.uint8)
    handle.start(_oracle_render(shard, 24, 20), transform=to_u8)
    frames = handle.wait()
    if rank == 0:
        assert frames.dtype == torch.uint8 and torch.equal(frames, to_u8(full))
        torch.save(full, os.path.join(out_dir, "root.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _depth2_worker(rank, world, port, n_total, out_dir):
    """bench.py's hand-over loop at depth 2 (round 4): step k starts its gather after waiting for step k - 2's;
    every step hands over DIFFERENT frames (scaled by the step number), so a receive buffer that is reused too
    early -- or a wait() that returns the wrong step -- shows."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    distributed.init_from_env(backend="gloo")
    job = synthetic.sphere_job(n_total, 24, 20, 6)
    shard = distributed.shard_batch(job, rank, world)
    base = _oracle_render(shard, 24, 20)
    handle = distributed.ImageGather(n_total, mode="root", dst=0, depth=2)
    received, held = [], []
    for k in range(5):
        if handle.in_flight() >= handle.depth:
            out = handle.wait()              # step k - 2's frames
            received.append(None if out is None else out.clone())
            held.append(out)                 # still valid while the NEXT start() runs (depth + 1 buffers)
        handle.start(base * float(k + 1))
        if held and held[-1] is not None:    # the tensor wait() just returned was not touched by that start()
            assert torch.equal(held[-1], received[-1])
    assert handle.in_flight() == 2
    try:
        handle.start(base)                   # a third one in flight: refused
        raise AssertionError("depth 2 accepted three hand-overs in flight")
    except RuntimeError:
        pass
    for out in handle.drain():
        received.append(None if out is None else out.clone())
    assert handle.in_flight() == 0 and handle.wait() is None
    if rank == 0:
        torch.save(received, os.path.join(out_dir, "depth2.pt"))
    else:
        assert all(r is None for r in received)
    dist.barrier()
    dist.destroy_process_group()


def _rotating_worker(rank, world, port, n_total, n_steps, out_dir):
    """bench.py's hand-over loop at N > 1 since round 5: RotatingImageGather -- step s's global batch lands on rank
    s mod N, one exchange per block of N steps, depth 2.  Every step hands over different frames."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    distributed.init_from_env(backend="gloo")
    job = synthetic.sphere_job(n_total, 24, 20, 6)
    shard = distributed.shard_batch(job, rank, world)
    base = _oracle_render(shard, 24, 20)
    handle = distributed.RotatingImageGather(n_total, depth=2)
    to_u8 = lambda t: (t.clamp(0, 8) * 31).to(torch.uint8)
    got = []
    for k in range(n_steps):
        if handle.in_flight() >= handle.depth:
            out = handle.wait()
            got.append(None if out is None else (out[0], out[1].clone()))
        handle.start(base * float(k + 1), transform=to_u8)
        assert handle.in_flight() <= handle.depth
    for out in handle.drain():
        got.append(None if out is None else (out[0], out[1].clone()))
    assert handle.in_flight() == 0 and handle.wait() is None
    torch.save([g for g in got if g is not None], os.path.join(out_dir, "rotating%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total,n_steps", [(2, 5, 7), (3, 5, 7), (3, 6, 9)])   # uneven / even shards, with and without a tail
def test_rotating_gather(tmp_path, world, n_total, n_steps):
    mp.spawn(_rotating_worker, args=(world, _free_port(), n_total, n_steps, str(tmp_path)), nprocs=world, join=True)
    full = _oracle_render(synthetic.sphere_job(n_total, 24, 20, 6), 24, 20)
    seen = {}
    for rank in range(world):
        for step, images in torch.load(os.path.join(str(tmp_path), "rotating%d.pt" % rank)):
            assert step % world == rank and step not in seen
            seen[step] = images
    assert sorted(seen) == list(range(n_steps))          # every step's global batch landed on exactly one rank
    for step, images in seen.items():
        want = ((full * float(step + 1)).clamp(0, 8) * 31).to(torch.uint8)
        assert torch.equal(images, want), "step %d" % step


def test_rotating_gather_single_process_is_identity():
    handle = distributed.RotatingImageGather(3, depth=1)
    for k in range(3):
        if handle.in_flight() >= handle.depth:
            step, images = handle.wait()
            assert step == k - 1 and torch.equal(images, torch.full((3, 2, 2, 4), float(k - 1)))
        handle.start(torch.full((3, 2, 2, 4), float(k)))
    (last,) = handle.drain()
    assert last[0] == 2 and torch.equal(last[1], torch.full((3, 2, 2, 4), 2.0))


def test_gather_depth_two_world2(tmp_path):
    mp.spawn(_depth2_worker, args=(2, _free_port(), 5, str(tmp_path)), nprocs=2, join=True)
    full = _oracle_render(synthetic.sphere_job(5, 24, 20, 6), 24, 20)
    received = torch.load(os.path.join(str(tmp_path), "depth2.pt"))
    assert len(received) == 5                # steps 0..4, in order
    for k in range(5):
        assert torch.equal(received[k], full * float(k + 1)), "hand-over %d" % k


def test_gather_to_root_world2(tmp_path):
    mp.spawn(_root_worker, args=(2, _free_port(), 5, str(tmp_path)), nprocs=2, join=True)
    job = synthetic.sphere_job(5, 24, 20, 6)
    assert torch.equal(torch.load(os.path.join(str(tmp_path), "root.pt")), _oracle_render(job, 24, 20))


def _worker(rank, world, port, n_total, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = distributed.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    job = synthetic.sphere_job(n_total, 48, 40, 6)
    local, handle = distributed.render_sharded(lambda s: _oracle_render(s, 48, 40), job, n_total)
    begin, end = distributed.shard_bounds(n_total, rank, world)
    assert local.shape[0] == end - begin
    full = handle.wait()
    grad = torch.full((job["vertices"].shape[1], 3), float(rank + 1))
    distributed.allreduce_shared_mesh_grad(grad)
    torch.save({"full": full, "grad": grad}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [4, 5])   # even and uneven shards
def test_sharded_render_and_gather_world2(tmp_path, n_total):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True)
    job = synthetic.sphere_job(n_total, 48, 40, 6)
    want = _oracle_render(job, 48, 40)
    for rank in range(world):
        got = torch.load(os.path.join(str(tmp_path), "rank%d.pt" % rank))
        assert got["full"].shape == want.shape
        assert torch.equal(got["full"], want), "gathered images differ from the single-process render"
        assert torch.all(got["grad"] == 3.0)   # 1 + 2: summed over ranks


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 32, 33, 64):
        for world in (1, 2, 3, 8):
            spans = [distributed.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_gather_is_identity():
    x = torch.arange(24.0).reshape(2, 3, 4, 1)
    assert torch.equal(distributed.gather_images(x, 2), x)


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE must start the two ranks itself (fresh children
    under torch.distributed.run) and propagate their status.  Here there is no GPU: both ranks come up,
    find no MI355X, refuse to fall back, and the parent exits non-zero with their message."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("the no-GPU behaviour is what this test pins")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1",
                           "--warmup", "0", "--cpu-sample", "0"], env=env, capture_output=True, text=True,
                          timeout=300)
    assert proc.returncode != 0
    assert "needs an MI355X" in proc.stderr
    assert '"metric"' not in proc.stdout
