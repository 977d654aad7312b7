"""Fuzz of the fused shading backward on random triangle soups (GPU): the rows kernel and every lane-accumulating
variant (k_accumulate_lanes: folded / difference-basis, dense and sign-coded upstream), EACH against the float64
truth (oracle/truth64.py: the reference's gradient formulas in binary64 on the same stored barycentrics) within the
rounding bound of tests/backward_fuzz.py -- not against one another: on sliver triangles two binary32 evaluations
differ by more than either differs from the truth.  Finds races and corner cases of the lane kernels' run
bookkeeping (one-pixel runs, merge-table overflow, strips that end mid-run, ragged image edges): those are off by
orders of magnitude more than the bound.  A fixed-seed slice runs inside `pytest -m gpu`
(tests/test_backward_truth_gpu.py).

    python tests/fuzz_shade_backward_gpu.py [--trials N] [--seed S] [--small]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import backward_fuzz

ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=120)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--small", action="store_true")
args = ap.parse_args()
t0 = time.time()
report = backward_fuzz.run(backward_fuzz.shade_trial, args.trials, args.seed, small=args.small, progress=20)
for line in report.failures[:40]:
    print("MISMATCH", line)
bad = bool(report.failures) or report.with_gradients < args.trials // 2
print("FUZZ", "FAILED" if bad else "OK", "%d trials (%d with gradients), %d values beyond the rounding bound, %.0f s; worst excess "
      "(error / bound) per kernel: %s" % (args.trials, report.with_gradients, len(report.failures), time.time() - t0, report.summary()))
sys.exit(1 if bad else 0)
