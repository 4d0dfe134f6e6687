#!/usr/bin/env python3
"""Contiger on one GPU at scale: synthetic reads (text resident in HBM) -> filter (shk_count_chunks) -> unitig set
(shk_unitigs_add_reads, batches of chunks) -> unitigs.fa (shk_unitig_set_write). Prints one JSON line: seconds per
phase, device time of the walk kernels against the host wall time around them, unitig statistics."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sh-assembly_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build(args):
    import torch
    import bench
    import shk
    dev = torch.device("cuda:0")
    L, K = 150, args.k
    genome = torch.randint(0, 4, (args.genome,), dtype=torch.uint8, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
    text = bench.gen_batch_torch(torch, genome, args.reads, L, args.err, 0, 1, dev)
    rec = int(text.numel()) // args.reads
    offs, lens = bench.chunk_table(args.reads, rec)
    ctx = shk.Context(qb=args.qb, k=K, max_batch_bytes=64, max_batch_keys=args.reads * (L - K + 2) + 4096, max_batch_reads=args.reads + 1024)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.count_chunks(text.data_ptr(), offs, lens, on_device=True, text_bytes=int(text.numel()))
    torch.cuda.synchronize()
    t_count = time.perf_counter() - t0
    return torch, shk, ctx, text, offs, lens, t_count


def walk(shk, ctx, text, offs, lens, args, out_path):
    u = shk.UnitigSet(ctx)
    ctx.profile(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    nseeds = 0
    B = args.batch_chunks
    for a in range(0, len(offs), B):
        nseeds += u.add_reads(text.data_ptr(), offs[a:a + B], lens[a:a + B], args.k, args.amin, args.xmin, 1000000, args.max_len,
                              text_bytes=int(text.numel()))
    t_walk = time.perf_counter() - t0
    pg = ctx.profile_get()
    walk_kernel_ms = pg.get("k_ug_walk", (0, 0.0))[1] + pg.get("k_extend_forward+k_select_seeds", (0, 0.0))[1]
    t0 = time.perf_counter()
    st = u.write(args.k, out_path)
    t_write = time.perf_counter() - t0
    prof = ctx.profile_get()
    u.close()
    return nseeds, st, t_walk, t_write, walk_kernel_ms, prof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=20_000_000)
    ap.add_argument("--reads", type=int, default=8_000_000)
    ap.add_argument("--err", type=float, default=0.00234)
    ap.add_argument("--k", type=int, default=47)
    ap.add_argument("--qb", type=int, default=29)
    ap.add_argument("--amin", type=int, default=4, help="-s: minimum count of a k-mer used to extend (4 suits 60x; the default 2 of the command line suits ~10x)")
    ap.add_argument("--xmin", type=int, default=4, help="-x: minimum count of a seed")
    ap.add_argument("--max-len", type=int, default=1 << 26)
    ap.add_argument("--batch-chunks", type=int, default=64)
    ap.add_argument("--out", default="/tmp/unitigs.fa")
    args = ap.parse_args()
    torch, shk, ctx, text, offs, lens, t_count = build(args)
    nseeds, st, t_walk, t_write, walk_ms, prof = walk(shk, ctx, text, offs, lens, args, args.out)
    lens_ = []
    with open(args.out, "rb") as f:
        for ln in f:
            if not ln.startswith(b">"):
                lens_.append(len(ln) - 1)
    lens_.sort(reverse=True)
    tot, acc, n50 = sum(lens_), 0, 0
    for x in lens_:
        acc += x
        if acc * 2 >= tot:
            n50 = x
            break
    ext = st["extensions"]
    print(json.dumps({"metric": "Contiger on the device", "genome": args.genome, "reads": args.reads, "k": args.k, "qb": args.qb,
                      "abundance_min": args.amin, "count_s": t_count, "seeds": nseeds, "walk_wall_s": t_walk,
                      "walk_kernel_s": walk_ms / 1e3, "host_share_of_walk": 1.0 - walk_ms / 1e3 / t_walk if t_walk else None,
                      "finish_and_write_s": t_write, "unitigs": st["unitigs"], "total_len": st["total_len"], "n50": n50,
                      "longest": lens_[0] if lens_ else 0, "rounds": st["rounds"], "extended_bases": ext,
                      "duplicates": st["duplicates"], "bases_per_s_walk_kernel": ext / (walk_ms / 1e3) if walk_ms else None,
                      "lookups_per_s_walk_kernel": 7 * ext / (walk_ms / 1e3) if walk_ms else None,
                      "GBps_at_97B_per_lookup": 97 * 7 * ext / (walk_ms / 1e3) / 1e9 if walk_ms else None,
                      "kernel_ms": {k: round(v[1], 2) for k, v in prof.items()}}))
    ctx.close()


if __name__ == "__main__":
    main()
