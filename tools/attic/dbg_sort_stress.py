#!/usr/bin/env python3
"""Run ON THE GPU BOX: tests/test_gpu_sort_stress.py's random clouds for many more seeds (argv: first, last)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_sort_stress as t
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(a, b):
    try:
        t.test_random_clouds_every_level_depth(seed)
    except AssertionError as e:
        bad += 1
        print("FAILED seed", seed, str(e)[:200])
print("seeds", a, "..", b, "failures", bad)
