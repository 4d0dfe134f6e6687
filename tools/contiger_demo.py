#!/usr/bin/env python3
"""End-to-end demo at a moderate scale on one GPU: synthetic reads -> filter (library) -> .cqf -> bin/Contiger -> unitigs.fa.
Prints one JSON line with the timings and the N50 of the unitigs."""
import argparse, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sh-assembly_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=4_000_000)
    ap.add_argument("--reads", type=int, default=800_000)
    ap.add_argument("--k", type=int, default=47)
    ap.add_argument("--qb", type=int, default=25)
    ap.add_argument("--extra", default="", help="extra Contiger flags")
    args = ap.parse_args()
    import torch, bench, shk
    dev = torch.device("cuda:0")
    L, K = 150, args.k
    genome = torch.randint(0, 4, (args.genome,), dtype=torch.uint8, device=dev)
    text = bench.gen_batch_torch(torch, genome, args.reads, L, 0.00234, 0, 1, dev)
    rec = int(text.numel()) // args.reads
    offs, lens = bench.chunk_table(args.reads, rec)
    ctx = shk.Context(qb=args.qb, k=K, max_batch_bytes=int(text.numel()) + 4096, max_batch_keys=args.reads * (L - K + 2))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.count_chunks(text.data_ptr(), offs, lens, on_device=True, text_bytes=int(text.numel()))
    t_count = time.perf_counter() - t0
    d = tempfile.mkdtemp(prefix="contiger_demo_")
    cqf = os.path.join(d, "k.cqf")
    ctx.export_cqf(cqf)
    ctx.close()
    with open(os.path.join(d, "reads.fq"), "wb") as f:
        f.write(text.cpu().numpy().tobytes())
    with open(os.path.join(d, "files.txt"), "w") as f:
        f.write("reads.fq\n")
    del text
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    r = subprocess.run([os.path.join(ROOT, "sh-assembly_amd", "bin", "Contiger"), "-k", str(K), "-i", os.path.join(d, "files.txt"), "-c", cqf,
                        "-o", os.path.join(d, "unitigs.fa")] + args.extra.split(), capture_output=True, text=True, cwd=d)
    t_walk = time.perf_counter() - t0
    lens_ = []
    if r.returncode == 0:
        with open(os.path.join(d, "unitigs.fa"), "rb") as f:
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
    print(json.dumps({"genome": args.genome, "reads": args.reads, "k": K, "count_s": t_count, "contiger_s": t_walk, "rc": r.returncode,
                      "unitigs": len(lens_), "total_len": tot, "n50": n50, "longest": lens_[0] if lens_ else 0,
                      "stderr_tail": r.stderr.strip().split("\n")[-2:]}))

if __name__ == "__main__":
    main()
