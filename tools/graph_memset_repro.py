"""Minimal repro attempt for the ROCm finding of DESIGN.md section 4.7: a hipMemsetAsync NODE of a captured
graph wrote 0xC6 instead of 0 on the second replay once eager work had run in between (round 5, ROCm 7.2; every
zero-fill of the library became a kernel because of it).  No library code here: torch for the capture, ctypes for the
memset.  Prints, per buffer size, what the node left behind after replay 1, after eager allocator traffic, and after
replay 2, with the node's captured pointer next to what the allocator hands out afterwards.

    python tools/graph_memset_repro.py
"""
import ctypes

import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int


def trial(nbytes, in_pool):
    dev = torch.device("cuda:0")
    outside = torch.full((nbytes,), 7, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        torch.zeros(16, device=dev).sum()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        buf = torch.empty(nbytes, dtype=torch.uint8, device=dev) if in_pool else outside
        rc = hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, nbytes, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
        out = buf.to(torch.int32).sum().reshape(1).clone()
    ptr = buf.data_ptr()
    buf.fill_(9)
    g.replay()
    torch.cuda.synchronize()
    first = int(out.item())
    # eager traffic: allocations of the same size class, kernels, an empty_cache
    junk = [torch.full((nbytes,), 0xC6, dtype=torch.uint8, device=dev) for _ in range(64)]
    aliases = sum(1 for j in junk if j.data_ptr() == ptr)
    big = torch.rand(1 << 24, device=dev).sum()
    del junk
    torch.cuda.empty_cache()
    junk = [torch.full((nbytes,), 0xC6, dtype=torch.uint8, device=dev) for _ in range(64)]
    aliases += sum(1 for j in junk if j.data_ptr() == ptr)
    buf.fill_(9)
    g.replay()
    torch.cuda.synchronize()
    second, raw = int(out.item()), buf[:8].tolist()
    print("%6d bytes, %-22s replay 1 sum %d, replay 2 sum %d (first bytes %s), eager allocations at the node's pointer: %d  %s" % (
        nbytes, "buffer from graph pool" if in_pool else "buffer allocated before", first, second, raw, aliases,
        "OK" if first == 0 and second == 0 else "<<< WRONG"), flush=True)
    del big
    return first == 0 and second == 0


if __name__ == "__main__":
    ok = True
    for in_pool in (True, False):
        for n in (8, 64, 512, 4096, 1 << 20):
            ok = trial(n, in_pool) and ok
    print("memset nodes behaved" if ok else "memset node misbehaved: see the lines marked WRONG")
