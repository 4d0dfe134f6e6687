import sys, random, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "sh-assembly_amd"))
import torch
import shk, f4_scenarios as F
rng = random.Random(5)
def mk(qb):
    return shk.Context(qb=qb, k=21, max_batch_keys=1 << 22)
F.check_counted_and_dump(mk, 12, F.pairs(rng, 12, 1500, 1 << 20)); print("ok 1", flush=True)
F.check_counted_and_dump(mk, 14, F.pairs(rng, 14, 7000, 1 << 24)); print("ok 2", flush=True)
F.check_counted_and_dump(mk, 12, F.pairs(rng, 12, 900, 1 << 16, cluster=(2000, 300))); print("ok 3", flush=True)
F.check_counted_and_dump(mk, 12, F.pairs(rng, 12, 400, 1 << 16, cluster=(4000, 96))); print("ok 4", flush=True)
F.check_counted_and_dump(mk, 10, F.pairs(rng, 10, 40, 1 << 35), batches=2); print("ok 5", flush=True)
F.check_counted_and_dump(mk, 12, F.pairs(rng, 12, 1400, 3, cluster=(1024, 256), special=0.0), batches=1); print("ok 6", flush=True)
