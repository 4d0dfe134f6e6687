#!/usr/bin/env python3
"""Randomised parity run of the Contiger slice on the GPU: random genomes with repeats, error rates, k and seed densities;
shk_unitigs_from_seeds / shk_find_unitigs against the oracle's get_unitig_forward and the Python closure over it."""
import argparse, os, random, sys, tempfile, pathlib, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sh-assembly_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch  # noqa: F401
    import shk
    from test_emu_kernels import _find_unitigs_case
    rnd = random.Random(args.seed)
    bad = 0
    t0 = time.time()
    for case in range(args.cases):
        k = rnd.choice([21, 31, 47, 63, 64])
        G = rnd.choice([1500, 4000, 9000])
        L = rnd.choice([max(k + 20, 80), 150])
        cov = rnd.choice([8, 15, 30])
        nreads = G * cov // L
        qb = 15 if nreads * (L - k + 1) < 9000 else 17 if nreads * (L - k + 1) < 40000 else 19
        cfg = dict(qb=qb, k=k, G=G, nreads=nreads, L=L, err=rnd.choice([0.0, 0.002, 0.01]), repeat=rnd.choice([0, k + 5, 3 * k]),
                   seed_every=rnd.choice([1, 7, 40]))
        if os.environ.get("FUZZ_STEP"):
            os.environ["SHK_WALK_STEP"] = str(rnd.choice([13, 100, 5000]))
        try:
            with tempfile.TemporaryDirectory() as d:
                _find_unitigs_case(lambda **kw: shk.Context(**kw), pathlib.Path(d), **cfg)
        except AssertionError as e:
            bad += 1
            print("MISMATCH case", case, cfg, str(e)[:200])
    print(f"fuzz_walk: {args.cases} cases, {bad} mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
