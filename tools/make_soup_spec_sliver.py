"""Regenerates tests/golden/soup_spec_sliver_seed17_trial43.npz: the inputs of the one recorded failure of the round-4
lane-variant fuzzer (`tests/fuzz_lane_variants_gpu.py --seed 17`, trial 43: B=2 V=185 T=991 207x181 L=1, "d clip
deviates by 3.2e-02 of 6.7e-04" between the specular lane kernel and the rows kernel), by replaying that script's
random stream on the host.  Inputs only: the test (tests/test_backward_truth_gpu.py) runs the forward on the device and
adjudicates every kernel against oracle/truth64.py."""
import os
import numpy as np

rng = np.random.default_rng(17)
for trial in range(44):
    B = int(rng.integers(1, 4)); V = int(rng.integers(4, 250))
    T = int(rng.integers(1, 2000 if trial % 4 == 3 else 350))
    W, H = int(rng.integers(8, 400)), int(rng.integers(8, 280))
    pos = (rng.normal(size=(B, V, 3)) * [1.0, 1.0, 0.3] * (0.2 if trial % 4 == 1 else 1.0)).astype(np.float32)
    tris = rng.integers(0, V, size=(T, 3)).astype(np.int32)
    A = int(rng.integers(1, 13))
    rng.normal(size=(B, V, A)); rng.normal(size=(A,)); rng.normal(size=(B, H, W, A))
    L = int(rng.integers(1, 5))
    nrm, kd, ks = rng.normal(size=(B, V, 3)).astype(np.float32), rng.random(size=(B, V, 3)).astype(np.float32), rng.random(size=(B, V, 3)).astype(np.float32)
    lp = (rng.normal(size=(B, L, 3)) * 3.0 + [0.0, 0.0, 4.0]).astype(np.float32)
    li = (rng.random(size=(B, L, 3)) + 0.1).astype(np.float32)
    amb = (rng.random(size=(B, 3)) * 0.3).astype(np.float32) if trial % 2 else None
    cam = (rng.normal(size=(B, 3)) + [0.0, 0.0, 5.0]).astype(np.float32)
    shin = (1.2 + 2.0 * rng.random(size=(B, V))).astype(np.float32) if trial % 3 == 0 else (1.2 + 3.0 * rng.random(size=(B,))).astype(np.float32)
    g = rng.normal(size=(B, H, W, 4)).astype(np.float32) / np.float32(H * W)
assert (B, V, T, W, H, L) == (2, 185, 991, 207, 181, 1), (B, V, T, W, H, L)
xf = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
xf[:, 3, 2] = 0.5
xf[:, 3, 3] = 1.2
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "soup_spec_sliver_seed17_trial43.npz")
np.savez_compressed(out, positions=pos, transforms=xf, triangles=tris, normals=nrm, diffuse=kd, specular=ks, light_positions=lp,
                    light_intensities=li, ambient=amb, camera=cam, shininess=shin, upstream=g, W=W, H=H)
print(out, os.path.getsize(out))
