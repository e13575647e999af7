#!/usr/bin/env python3
"""Run ON THE GPU BOX: tests/test_gpu_traverse_stress.py's random traversals for many more seeds (argv: first, last)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_traverse_stress as t
a, b = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3 and sys.argv[3] == "tight":  # BFS queues that start too small and grow by the bare minimum: many resumes
    from implicitbvh_amd import api
    api.BFS_INITIAL_FACTOR, api.BFS_GROWTH = 1, 1
bad = 0
for seed in range(a, b):
    try:
        t.check_one(seed)
    except AssertionError as e:
        bad += 1
        print("FAILED seed", seed, str(e)[:200])
print("seeds", a, "..", b, "failures", bad)
