"""Soak: the randomised parity tests of tests/test_gpu_parity.py over many more seeds than the suite runs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest
from rufus_amd import capi
import tests.test_gpu_parity as T


class MP:
    def __init__(self): self.saved = {}
    def setenv(self, k, v): self.saved.setdefault(k, os.environ.get(k)); os.environ[k] = v
    def delenv(self, k): self.saved.setdefault(k, os.environ.get(k)); os.environ.pop(k, None)
    def undo(self):
        for k, v in self.saved.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
        self.saved = {}


ctx = capi.Context(0)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
bad = 0
for seed in range(lo, hi):
    for name, fn, needs_mp in (("count", T.test_three_count_paths_agree_on_random_configurations, True),
                               ("filter", T.test_filter_matches_oracle_on_random_configurations, False),
                               ("k4 search", lambda c, sd, mp: T.test_merge_hashlist_query_on_random_configurations(c, sd, "search", mp), True),
                               ("k4 tiles", lambda c, sd, mp: T.test_merge_hashlist_query_on_random_configurations(c, sd, "tiles", mp), True)):
        mp = MP()
        try:
            fn(ctx, seed, mp) if needs_mp else fn(ctx, seed)
        except AssertionError as e:
            bad += 1
            print("FAIL", name, seed, str(e)[:300], flush=True)
        finally:
            mp.undo()
print(f"soak seeds [{lo},{hi}): {bad} failures in {time.time() - t0:.0f} s")
