#!/usr/bin/env python3
"""Throughput of the unitig extension (Contiger, first slice) on one GPU: builds a filter from synthetic reads
with the library, takes the reads' middle k-mers as seeds (Contiger's seed rule), and times
shk_unitigs_from_seeds. Prints one JSON line (extended bases/s, lookups/s, bytes/s at SURVEY 8d's 97 B/lookup)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sh-assembly_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=4_000_000)
    ap.add_argument("--reads", type=int, default=800_000)
    ap.add_argument("--seeds", type=int, default=200_000)
    ap.add_argument("--k", type=int, default=47)
    ap.add_argument("--qb", type=int, default=25)
    ap.add_argument("--max-len", type=int, default=4096)
    args = ap.parse_args()
    import torch
    import bench
    import shk
    dev = torch.device("cuda:0")
    L, K = 150, args.k
    genome = torch.randint(0, 4, (args.genome,), dtype=torch.uint8, device=dev)
    text = bench.gen_batch_torch(torch, genome, args.reads, L, 0.00234, 0, 1, dev)
    rec = int(text.numel()) // args.reads
    offs, lens = bench.chunk_table(args.reads, rec)
    ctx = shk.Context(qb=args.qb, k=K, max_batch_bytes=int(text.numel()) + 4096, max_batch_keys=args.reads * (L - K + 2))
    torch.cuda.synchronize()
    ctx.count_chunks(text.data_ptr(), offs, lens, on_device=True, text_bytes=int(text.numel()))
    # seeds: middle k-mer of each read (contig_assembly.cpp:1858-1862)
    tcpu = text.cpu().numpy().tobytes()
    import cqflibs
    O = cqflibs.oracle()
    seeds, keys = [], []
    hb = args.qb + 8
    seen = set()
    for r in range(min(args.seeds * 2, args.reads)):
        line = tcpu[r * rec:(r + 1) * rec].split(b"\n")[1]
        mid = len(line) // 2 - K // 2
        km = line[mid:mid + K]
        if b"N" in km or km in seen:
            continue
        seen.add(km)
        fh, rh = O.nthash(km, K)
        seeds.append(km); keys.append(min(fh, rh) & ((1 << hb) - 1))
        if len(seeds) >= args.seeds:
            break
    cnt, _ = ctx.lookup(keys, mode=2)
    sel = [(s, c) for s, c in zip(seeds, cnt) if 2 <= c <= 1000000]
    seeds, counts = [s for s, _ in sel], [c for _, c in sel]
    ctx.profile(True)
    t0 = time.perf_counter()
    out = ctx.unitigs_from_seeds(seeds, counts, K, 2, args.max_len)
    dt = time.perf_counter() - t0
    prof = ctx.profile_get()
    ext = sum(len(s) - K for s, _, _ in out)
    kms = prof.get("k_extend_forward+k_select_seeds", (0, 0.0))[1]
    steps = ext + 2 * len(out)          # every walk also pays the step at which it stops, twice
    print(json.dumps({"metric": "unitig extension (Contiger first slice)", "seeds": len(out), "extended_bases": ext,
                      "mean_unitig_len": (ext / max(len(out), 1)) + K, "wall_s": dt, "kernel_ms": kms,
                      "bases_per_s_kernel": ext / (kms / 1e3) if kms else None,
                      "lookups_per_s_kernel": 7 * steps / (kms / 1e3) if kms else None,
                      "GBps_at_97B_per_lookup": 97 * 7 * steps / (kms / 1e3) / 1e9 if kms else None,
                      "stops": {str(k): sum(1 for _, _, st in out for x in st if x == k) for k in (1, 2, 3, 4)}}))
    ctx.close()

if __name__ == "__main__":
    main()
