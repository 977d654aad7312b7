"""Fuzz of the other two lane-accumulating backwards on random soups (GPU):
  * rasterize()'s one-pass backward, mr_interpolate_raster_backward: rows kernel and AttrFoldLaneFn (normalised G-buffer);
  * the specular shading backward: rows kernel (everything), SpecFoldLaneFn (vertex gradients only: 18 sums; folded: 9).
EACH kernel against the float64 truth (oracle/truth64.py) within the rounding bound of tests/backward_fuzz.py -- see
tests/fuzz_shade_backward_gpu.py.  A fixed-seed slice runs inside `pytest -m gpu` (tests/test_backward_truth_gpu.py).

    python tests/fuzz_lane_variants_gpu.py [--trials N] [--seed S] [--small]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import backward_fuzz

ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=200)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--small", action="store_true")
args = ap.parse_args()
t0 = time.time()
failures, summaries, short = [], [], False
for name, fn in (("rasterize", backward_fuzz.attr_trial), ("specular", backward_fuzz.specular_trial)):
    report = backward_fuzz.run(fn, args.trials, args.seed, small=args.small, progress=40)
    failures += report.failures
    short = short or report.with_gradients < args.trials // 2
    summaries.append("%s (%d with gradients): %s" % (name, report.with_gradients, report.summary()))
for line in failures[:40]:
    print("MISMATCH", line)
bad = bool(failures) or short
print("FUZZ", "FAILED" if bad else "OK", "%d trials each, %d values beyond the rounding bound, %.0f s; worst excess (error / bound) "
      "per kernel -- %s" % (args.trials, len(failures), time.time() - t0, " | ".join(summaries)))
sys.exit(1 if bad else 0)
