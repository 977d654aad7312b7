"""Multi-GPU execution of render jobs: one process per GPU, batch-sharded.

The reference is single-process and loops over the batch serially
(src/mesh_renderer/rasterize.py:112-121); every image is an independent (camera,
mesh) job, so the batch shards across ranks with NO data-path collective.  The only
exchange is the hand-over of the finished images: one all-gather (or gather to a
root; round 5: to a ROTATING root, RotatingImageGather -- bench.py's default at N > 1) of
[B_local, H, W, 4] images over RCCL / xGMI, which is independent of the backward pass and
therefore issued on a side stream so that it overlaps it.  When
all ranks optimise one shared mesh, its [V,3] gradient is summed with a (latency
bound, 30 KB - 300 KB) all-reduce.

torch.distributed's "nccl" backend is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import collections
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment.

    Returns (rank, world_size, local_rank).  A single-process run (no WORLD_SIZE or
    WORLD_SIZE=1) does not create a group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # MR_DIST_BACKEND=gloo rehearses the multi-rank code path on a box with fewer GPUs
            # than ranks (RCCL refuses two ranks on one device)
            backend = os.environ.get("MR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank


def shard_bounds(n_items, rank, world_size):
    """Contiguous, balanced [begin, end) of `n_items` independent jobs for `rank`:
    the first n_items % world_size ranks get one extra job."""
    base, extra = divmod(n_items, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_batch(tensors, rank, world_size):
    """Slice every [B, ...] tensor of a dict to this rank's jobs (non-tensors pass through)."""
    out = {}
    for key, value in tensors.items():
        if torch.is_tensor(value) and value.dim() > 0 and key != "triangles":
            begin, end = shard_bounds(value.shape[0], rank, world_size)
            out[key] = value[begin:end]
        else:
            out[key] = value
    return out


class ImageGather:
    """Hand-over of per-rank image shards, overlappable with the backward pass.

    start(local) enqueues the collective on a side stream (GPU) and returns at once;
    wait() makes the current stream wait for it and returns the [B_total, H, W, C]
    tensor (mode "all": on every rank; mode "root": on rank `dst`, None elsewhere).
    Shards may be uneven (they are padded to the largest one).

    depth (round 4): how many hand-overs may be in flight at once.  wait() returns the OLDEST one; with
    depth 2 a step waits for the hand-over of the step BEFORE the previous one right before it starts
    its own, so a hand-over has two steps to finish in: bound by the root's inbound links it then costs
    their bandwidth, not their latency on top.  The returned tensor is one of depth + 1 receive buffers
    used in turn: it stays valid until the (depth + 1)-th start() after the start() that filled it.

    mode "all"  = all_gather: every rank ends up with every image.
    mode "root" = gather to one rank (RCCL point-to-point under the hood): on a fully
                  connected xGMI node the root receives its N-1 shards over N-1 different
                  links at once, the other ranks only send -- 1/N of the all-gather's traffic."""

    def __init__(self, n_total, group=None, force_collective=False, mode="all", dst=0, depth=1):
        """force_collective: run the collective even in a 1-rank group (smoke tests of the
        RCCL / side-stream path on a single GPU)."""
        if mode not in ("all", "root"):
            raise ValueError("mode must be 'all' or 'root'")
        if depth < 1:
            raise ValueError("depth must be at least 1")
        self.mode, self.dst, self.depth = mode, dst, int(depth)
        self._pending = collections.deque()   # (event or None, out or None, max_count, send buffer) per hand-over in flight
        self.group = group
        self.force = bool(force_collective) and dist.is_initialized()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_total = n_total
        self.counts = [shard_bounds(n_total, r, self.world)[1] - shard_bounds(n_total, r, self.world)[0]
                       for r in range(self.world)]
        self._side = None
        # receive buffers (depth + 1, used in turn) and the padded send buffer are allocated once per
        # (shape, dtype, device): a step's hand-over allocates nothing (round 2 allocated `gathered`
        # and its chunk list every step).
        self._recv = {}
        self._turn = 0
        self._send_pad = {}

    def start(self, local, transform=None):
        """transform: optional callable applied to `local` ON THE SIDE STREAM before the
        collective (e.g. mesh_renderer.to_uint8: the 8-bit conversion then overlaps the caller's
        compute too instead of sitting on its stream)."""
        if len(self._pending) >= self.depth:
            raise RuntimeError("%d hand-over(s) already in flight (depth %d): call wait() first"
                               % (len(self._pending), self.depth))
        if self.world == 1 and not self.force:
            self._pending.append((None, transform(local) if transform is not None else local, 0, None))
            return
        max_count = max(self.counts)
        if local.shape[0] != self.counts[self.rank]:
            raise ValueError("rank %d holds %d images, expected %d"
                             % (self.rank, local.shape[0], self.counts[self.rank]))
        on_gpu = local.is_cuda
        if on_gpu:
            if self._side is None:
                self._side = torch.cuda.Stream(device=local.device)
            self._side.wait_stream(torch.cuda.current_stream(local.device))
        ctx = torch.cuda.stream(self._side) if on_gpu else _NullContext()
        with ctx:
            if transform is not None:
                if on_gpu:
                    local.record_stream(self._side)   # read by the side stream: keep it alive for it
                send = transform(local).detach()     # (the tensor itself: it may carry ready-made frames)
            else:
                send = local.detach()
            key = (tuple(send.shape[1:]), send.dtype, send.device)
            if send.shape[0] != max_count:   # an uneven shard: padded to the largest one, in a buffer kept for that
                padded = self._send_pad.get(key)
                if padded is None:
                    padded = self._send_pad[key] = torch.zeros((max_count,) + key[0], dtype=send.dtype,
                                                               device=send.device)
                padded[:send.shape[0]].copy_(send)
                send = padded
            send = send.contiguous()
            if on_gpu:
                send.record_stream(self._side)
            gathered, chunks = None, None
            if self.mode == "all" or self.rank == self.dst:
                ring = self._recv.get(key)
                if ring is None:
                    ring = self._recv[key] = []
                    for _ in range(self.depth + 1):
                        buf = torch.empty((self.world * max_count,) + key[0], dtype=send.dtype, device=send.device)
                        ring.append((buf, list(buf.chunk(self.world, 0))))
                gathered, chunks = ring[self._turn % len(ring)]
                self._turn += 1
            if self.mode == "root":
                dist.gather(send, chunks, dst=self.dst, group=self.group)
            elif dist.get_backend(self.group) == "gloo":
                dist.all_gather(chunks, send, group=self.group)
            else:
                dist.all_gather_into_tensor(gathered, send, group=self.group)
            done = None
            if on_gpu:   # wait() orders the caller's stream behind THIS hand-over, not behind the whole side stream
                done = torch.cuda.Event()
                done.record(self._side)
            # (the send buffer must outlive the collective: it rides along until the hand-over is waited for)
            self._pending.append((done, gathered, max_count, send))

    def in_flight(self):
        return len(self._pending)

    def wait(self):
        """The OLDEST hand-over in flight (None when there is none, and on a non-root rank of mode "root")."""
        if not self._pending:
            return None
        done, out, max_count, _send = self._pending.popleft()
        if self.world == 1 and not self.force:
            return out
        if done is not None:
            torch.cuda.current_stream().wait_event(done)
        if out is None:                    # mode "root" on a non-root rank
            return None
        if out.is_cuda:
            # allocated on the side stream, consumed on the caller's: tell the caching allocator
            out.record_stream(torch.cuda.current_stream(out.device))
        if all(c == max_count for c in self.counts):
            return out
        pieces = [out[r * max_count: r * max_count + c] for r, c in enumerate(self.counts)]
        return torch.cat(pieces, 0)

    def drain(self):
        """wait() for everything in flight; returns the results, oldest first."""
        out = []
        while self._pending:
            out.append(self.wait())
        return out


class RotatingImageGather:
    """Hand-over with a ROTATING root (round 5): the frames of step s -- the whole global batch of that step -- are
    assembled on rank s mod N (the rank that encodes / writes that step's frames), not all on rank 0.

    Why: xGMI is point-to-point.  Gathering every step's frames on ONE rank uses that rank's N - 1 inbound links and
    nothing else: at N = 8 with 134 MB of 8-bit frames per rank and step that is >= 1.75 ms per step at ~77 GB/s per
    link and direction, against a 0.65 ms render step -- the node would run at a third of N x one GPU (DESIGN.md section
    6).  With the root rotating, every link carries one shard per N steps.  And instead of N concurrent gathers on N
    communicators, the exchange is ONE balanced collective per block of N steps on the default communicator: every
    rank keeps the frames of its last N steps (the tensors themselves: no staging copy) and one all_to_all sends the
    frames of the block's step j to rank j, which receives that step's shard from every rank straight into its
    [B_total, H, W, C] buffer.  Per step and link that is 1/N of a shard (17 MB of 8-bit frames at N = 8: 0.22 ms),
    all N (N - 1) links busy at once, one collective per N steps, on a side stream under the next block's compute.

    start(local, transform)  keeps this step's frames and, when the block is full, issues the exchange; at most
                             `depth` exchanges may be in flight (wait() first, as with ImageGather).
    wait()                   the OLDEST exchange in flight -> (step, images [B_total, H, W, C]) for the step this rank
                             roots in that block, or None if the block ended before that step (a drained tail).
                             The tensor is one of depth + 1 receive buffers used in turn.
    drain()                  flushes a partly filled block, then wait()s for everything; results oldest first.
    in_flight()              exchanges in flight.
    Latency: a step's frames reach their root up to N steps later than with a per-step gather; a rank holds the
    frames of up to N steps (plus those of the exchanges in flight)."""

    def __init__(self, n_total, group=None, depth=2, force_collective=False, transport="rccl"):
        """force_collective: run the collective even in a 1-rank group (smoke tests of the RCCL / side-stream path on a
        single GPU).

        transport (round 6): "rccl" -- one all_to_all per block, RCCL's copy kernels on the compute units (the default, the
        path that has met RCCL) -- or "peer": every rank maps every root's receive buffers once (CUDA IPC handles exchanged
        over the group) and then COPIES its frames of the block's step j straight into root j's buffer on its side stream
        (hipMemcpy between devices: the copy engines, no kernel of ours or RCCL's), between two tiny barriers on the same
        stream -- "every root's slot is free" in front, "every copy has landed" behind.  Device tensors only.  Rehearsed
        with two processes on ONE GPU (tests/test_distributed_gpu.py); never run across two GPUs: DESIGN.md section 6."""
        if depth < 1:
            raise ValueError("depth must be at least 1")
        if transport not in ("rccl", "peer"):
            raise ValueError("transport must be 'rccl' or 'peer'")
        self.transport = transport
        self._peer = {}                       # per (shape, dtype, device): [root][slot] -> that root's receive buffer as seen from here
        self._flag = None
        self.group, self.depth = group, int(depth)
        self.force = bool(force_collective) and dist.is_initialized()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_total = n_total
        self.counts = [shard_bounds(n_total, r, self.world)[1] - shard_bounds(n_total, r, self.world)[0]
                       for r in range(self.world)]
        self._pending = collections.deque()   # (event or None, receive buffer, first step of the block, steps in it, sent tensors)
        self._recv = {}                       # per (shape, dtype, device): depth + 1 receive buffers, used in turn
        self._held = []                       # this block's frames so far, one tensor per step
        self._block = 0
        self._side = None

    def in_flight(self):
        return len(self._pending)

    def start(self, local, transform=None):
        if local.shape[0] != self.counts[self.rank]:
            raise ValueError("rank %d holds %d images, expected %d" % (self.rank, local.shape[0], self.counts[self.rank]))
        closes_block = len(self._held) + 1 == self.world
        if closes_block and len(self._pending) >= self.depth:
            raise RuntimeError("%d exchange(s) already in flight (depth %d): call wait() first"
                               % (len(self._pending), self.depth))
        on_gpu = local.is_cuda
        if on_gpu:
            if self._side is None:
                self._side = torch.cuda.Stream(device=local.device)
            self._side.wait_stream(torch.cuda.current_stream(local.device))
        with (torch.cuda.stream(self._side) if on_gpu else _NullContext()):
            if transform is not None:
                if on_gpu:
                    local.record_stream(self._side)   # read by the side stream: keep it alive for it
                send = transform(local).detach()     # (the tensor itself: it may carry ready-made frames)
            else:
                send = local.detach()
            send = send.contiguous()
            if on_gpu:
                # read by the exchange on the side stream, possibly long after the caller dropped its own reference
                send.record_stream(self._side)
            if self._held and (send.shape != self._held[0].shape or send.dtype != self._held[0].dtype):
                raise ValueError("the frames of one block must have one shape and dtype")
            self._held.append(send)
            if closes_block:
                self._exchange()

    def _exchange(self):
        """(on the side stream) this block's step-j frames -> rank j; this rank receives its step's shard from everyone."""
        held, n_steps = self._held, len(self._held)
        key = (tuple(held[0].shape[1:]), held[0].dtype, held[0].device)
        ring = self._recv.get(key)
        if ring is None:
            ring = self._recv[key] = [torch.empty((self.n_total,) + key[0], dtype=key[1], device=key[2])
                                      for _ in range(self.depth + 1)]
        out = ring[self._block % len(ring)]
        outs = list(out.split(self.counts, 0))                       # from rank r: its counts[r] images
        # a tail block has no frames for steps >= n_steps: their roots are sent the last step's again and ignore them
        inputs = [held[j] if j < n_steps else held[n_steps - 1] for j in range(self.world)]
        if self.transport == "peer" and self.world == 1:
            out = held[0]     # one rank: its frames ARE the global batch -- nothing is copied, the tensor itself is handed back
        elif self.world == 1 and not self.force:
            outs[0].copy_(inputs[0])
        elif self.transport == "peer":
            self._exchange_by_peer_copies(key, ring, held, n_steps)
        elif dist.get_backend(self.group) == "gloo":
            # (the CPU tests' backend has no all_to_all: the same exchange as non-blocking sends and receives; device
            #  tensors -- a rehearsal of N ranks on fewer GPUs, MR_DIST_BACKEND=gloo -- travel through the host)
            on_gpu = inputs[0].is_cuda
            src = [t.cpu() for t in inputs] if on_gpu else inputs
            dst = [torch.empty(o.shape, dtype=o.dtype) for o in outs] if on_gpu else outs
            dst[self.rank].copy_(src[self.rank])
            requests = []
            for r in range(self.world):
                if r != self.rank:
                    requests.append(dist.isend(src[r], r, group=self.group, tag=self._block))
                    requests.append(dist.irecv(dst[r], r, group=self.group, tag=self._block))
            for q in requests:
                q.wait()
            if on_gpu:
                for o, d in zip(outs, dst):
                    o.copy_(d)
        else:
            dist.all_to_all(outs, inputs, group=self.group)
        done = None
        if out.is_cuda:
            done = torch.cuda.Event()
            done.record(self._side)
        # (the sent tensors ride along until the exchange is waited for)
        self._pending.append((done, out, self._block * self.world, n_steps, inputs))
        self._block += 1
        self._held = []

    def _group_barrier(self, device):
        """(on the side stream) a barrier every rank's side stream passes in order: with RCCL a one-element all-reduce --
        stream-ordered: it completes on a rank only after every rank's earlier work on that stream has --, with gloo the
        side stream is drained on the host first."""
        if dist.get_backend(self.group) == "gloo":
            if device.type == "cuda":
                torch.cuda.current_stream(device).synchronize()
            dist.barrier(group=self.group)
            return
        if self._flag is None:
            self._flag = torch.zeros(1, dtype=torch.float32, device=device)
        dist.all_reduce(self._flag, group=self.group)

    def _exchange_by_peer_copies(self, key, ring, held, n_steps):
        device = key[2]
        if device.type != "cuda":
            raise ValueError("the peer transport copies between device buffers: frames must live on a GPU")
        views = self._peer.get(key)
        if views is None:   # once per frame shape: every rank learns how to reach every root's receive buffers
            from torch.multiprocessing.reductions import reduce_tensor
            mine = [reduce_tensor(buf) for buf in ring]
            everyone = [None] * self.world
            dist.all_gather_object(everyone, mine, group=self.group)
            views = [ring if r == self.rank else [rebuild(*args) for rebuild, args in everyone[r]] for r in range(self.world)]
            self._peer[key] = views
        slot = self._block % len(ring)
        first_row = sum(self.counts[:self.rank])
        self._group_barrier(device)          # every root has finished with what this slot held (its consumer ran before its start())
        for j in range(min(n_steps, self.world)):   # a tail block has no frames for the later roots: nothing is sent to them
            views[j][slot][first_row:first_row + self.counts[self.rank]].copy_(held[j], non_blocking=True)
        self._group_barrier(device)          # every rank's copies have landed in every root's buffer

    def close(self):
        """Drops the mapped views of the other ranks' buffers (peer transport) and lets every rank do so before any of them
        frees its buffers.  Call after drain(); a no-op for the RCCL transport."""
        if self._peer:
            self._peer = {}
            if torch.cuda.is_available():
                torch.cuda.synchronize()
                torch.cuda.ipc_collect()     # the consumer side of the IPC handles is released NOW, not at interpreter exit
            if dist.is_initialized() and self.world > 1:
                dist.barrier(group=self.group)

    def wait(self):
        if not self._pending:
            return None
        done, out, first, n_steps, _sent = self._pending.popleft()
        if done is not None:
            torch.cuda.current_stream().wait_event(done)
        if self.rank >= n_steps:       # (a drained tail: the block ended before this rank's step)
            return None
        if out.is_cuda:
            out.record_stream(torch.cuda.current_stream(out.device))
        return first + self.rank, out

    def drain(self):
        first = []
        if self._held:
            if len(self._pending) >= self.depth:
                first.append(self.wait())
            with (torch.cuda.stream(self._side) if self._side is not None else _NullContext()):
                self._exchange()
        while self._pending:
            first.append(self.wait())
        return first


class _NullContext:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def gather_images(local, n_total, group=None):
    """Blocking convenience wrapper around ImageGather."""
    g = ImageGather(n_total, group)
    g.start(local)
    return g.wait()


def allreduce_shared_mesh_grad(grad, group=None):
    """Sum the gradient of a mesh shared by all ranks (e.g. several views of one vertex
    set): one small all-reduce, in place."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group)
    return grad


def render_sharded(render_fn, batch, n_total, group=None):
    """Render this rank's shard of `batch` (a dict of full-batch tensors, see
    common.synthetic.sphere_job) with `render_fn(shard) -> [B_local,H,W,4]` and return
    (local_images, gather_handle).  Call gather_handle.wait() after the backward."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    shard = shard_batch(batch, rank, world)
    local = render_fn(shard)
    handle = ImageGather(n_total, group)
    handle.start(local)
    return local, handle
