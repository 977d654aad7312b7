"""The N > 1 hand-over's peer-copy transport, rehearsed with several PROCESSES on the box's one GPU (round 6).

RotatingImageGather(transport="peer"): every rank maps every root's receive buffers through CUDA IPC handles and copies its
frames of step j straight into root j's buffer between two barriers.  One GPU cannot show what the copy engines do across
xGMI; it can show that the handle exchange, the buffer turns, the barriers and the bookkeeping deliver every step's global
batch to exactly one rank, the right one, with the right frames -- the same assertions tests/test_distributed.py makes for
the RCCL / gloo exchange.  gloo is the control plane (RCCL refuses two ranks on one device)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pytorch_mesh_renderer_amd import distributed

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _frames(rank, count, step, dtype):
    base = torch.arange(count * 6 * 5 * 4, dtype=torch.float32).reshape(count, 6, 5, 4) % 97.0
    return (base + 100.0 * step + 1000.0 * rank).to(dtype)


def _worker(rank, world, port, n_total, n_steps, dtype_name, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")           # every rank on the one GPU
    torch.cuda.set_device(dev)
    dtype = getattr(torch, dtype_name)
    handle = distributed.RotatingImageGather(n_total, depth=2, transport="peer")
    count = handle.counts[rank]
    got = []
    for k in range(n_steps):
        if handle.in_flight() >= handle.depth:
            out = handle.wait()
            got.append(None if out is None else (out[0], out[1].cpu()))
        handle.start(_frames(rank, count, k, dtype).to(dev))
    for out in handle.drain():
        got.append(None if out is None else (out[0], out[1].cpu()))
    assert handle.in_flight() == 0 and handle.wait() is None
    torch.cuda.synchronize()
    handle.close()
    torch.save([g for g in got if g is not None], os.path.join(out_dir, "peer%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total,n_steps,dtype_name", [(2, 5, 7, "float32"), (3, 6, 9, "uint8"), (3, 5, 8, "float32")])
def test_peer_transport_delivers_every_step_to_its_root(device, tmp_path, world, n_total, n_steps, dtype_name):
    mp.spawn(_worker, args=(world, _free_port(), n_total, n_steps, dtype_name, str(tmp_path)), nprocs=world, join=True)
    counts = [distributed.shard_bounds(n_total, r, world)[1] - distributed.shard_bounds(n_total, r, world)[0] for r in range(world)]
    dtype = getattr(torch, dtype_name)
    seen = {}
    for rank in range(world):
        for step, images in torch.load(os.path.join(str(tmp_path), "peer%d.pt" % rank)):
            assert step % world == rank and step not in seen
            seen[step] = images
    assert sorted(seen) == list(range(n_steps))
    for step, images in seen.items():
        want = torch.cat([_frames(r, counts[r], step, dtype) for r in range(world)], 0)
        assert torch.equal(images, want), "step %d" % step


def test_peer_transport_single_rank_hands_the_frames_back_without_a_copy(device):
    handle = distributed.RotatingImageGather(3, depth=1, transport="peer")
    frames = torch.rand(3, 4, 4, 4, device=device)
    handle.start(frames)
    ((step, images),) = handle.drain()
    assert step == 0 and images.data_ptr() == frames.data_ptr()
    with pytest.raises(ValueError):
        distributed.RotatingImageGather(3, transport="carrier pigeon")
