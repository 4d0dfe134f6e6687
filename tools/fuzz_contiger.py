#!/usr/bin/env python3
"""Randomised whole-pipeline Contiger comparison: the device path (seeds from reads, walks, queued contigs, duplicate
removal, numbering, links, unitigs.fa) against the sequential restatement of the whole program
(oracle/contiger_pipeline.cpp) -- tests/contiger_cases.py on random genomes (repeats, plasmids, errors), k, read lengths,
thresholds and schedules. --emu runs the kernels in the CPU emulator build."""
import argparse, os, pathlib, random, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sh-assembly_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--emu", action="store_true")
    ap.add_argument("--scale", type=int, default=1, help="genome size multiplier (1: 300-900 bases, for the emulator)")
    args = ap.parse_args()
    if not args.emu:
        import torch  # noqa: F401
    import shk, contiger_cases as CC
    lib = os.path.join(ROOT, "tests", "emu", "libshk_emu.so") if args.emu else None
    kw = dict(threads_per_group=64, hash_groups=2) if args.emu else {}
    mk = lambda **k: shk.Context(lib_path=lib, **kw, **k)
    tmp = pathlib.Path(tempfile.mkdtemp())
    rnd = random.Random(args.seed)
    bad = skipped = 0
    tot = dict(unitigs=0, links=0, km_differ=0, circles=0, stale_links=0, one_sided=0)
    t0 = time.time()
    for i in range(args.cases):
        k = rnd.choice([21, 25, 31, 47, 63, 64])
        L = rnd.choice([2 * k + 9, 3 * k])
        G = rnd.choice([300, 500, 900]) * args.scale
        nreads = G * rnd.choice([8, 14, 25]) // L
        err = rnd.choice([0.0, 0.004, 0.01, 0.02])
        plasmid = rnd.choice([0, k + 30, 2 * k + 11])
        amin = rnd.choice([2, 2, 3, 1])
        per_read = rnd.random() < (0.5 if args.scale == 1 else 0.15)
        # a seed below the extension threshold (x < s) is a unitig of its own only if nobody's walk has looked at it yet:
        # schedule-dependent in the reference itself, so only the sequential schedule is compared there
        xmin = rnd.choice([2, 3, 1]) if per_read else max(amin, rnd.choice([2, 3, 1]))
        rule = 1      # the device takes seeds by processDataChunk's rule; the master's rule seeds other k-mers, i.e. possibly other components
        qb = 13
        while (1 << qb) < 3 * (G + nreads * L * err * k):
            qb += 1
        fq = CC.reads(G, nreads, L, err, plasmid, seed=rnd.randrange(1 << 20))
        cfg = dict(k=k, L=L, G=G, nreads=nreads, err=err, plasmid=plasmid, amin=amin, xmin=xmin, per_read=per_read, rule=rule, qb=qb)
        try:
            r = CC.run_case(mk, shk.UnitigSet, tmp, k=k, qb=qb, fq=fq, chunk_reads=rnd.choice([20, 50]) * args.scale, amin=amin, xmin=xmin,
                            per_read=per_read, rule=rule, max_len=8 * G + 1000)
            for x in tot:
                tot[x] += r[x]
        except AssertionError as e:
            bad += 1
            print("MISMATCH case", i, cfg, str(e)[:300], flush=True)
        except shk.ShkError as e:
            skipped += 1
    print(f"fuzz_contiger: {args.cases} cases, {bad} mismatches, {skipped} skipped, {tot}, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
