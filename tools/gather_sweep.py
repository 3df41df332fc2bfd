"""Seeded sweep of the tap-major gather paths (forward / input gradient / weight gradient vs torch) over 72 more shapes
than the test suite carries: python tools/gather_sweep.py  (GPU)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import test_ops_gpu as T
bad = 0
for seed in (11, 23, 37):
    for c in T._random_gather_cases(24, seed):
        try:
            T.test_gather_paths_on_plan_boundaries(c)
        except AssertionError as e:
            bad += 1; print("FAIL", c, e)
print("done, failures:", bad)
