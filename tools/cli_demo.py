#!/usr/bin/env python3
"""bin/CQF-deNoise end to end on plain FASTQ files (synthetic, written to a temporary directory): wall time and
k-mers/s including file reading, the chunker, the host->device copies and the GPU work."""
import argparse, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sh-assembly_amd"))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=100_000_000)
    ap.add_argument("--reads", type=int, default=8_000_000, help="reads per file")
    ap.add_argument("--files", type=int, default=2)
    args = ap.parse_args()
    import torch, bench
    dev = torch.device("cuda:0")
    L, K = 150, 47
    genome = torch.randint(0, 4, (args.genome,), dtype=torch.uint8, device=dev)
    d = tempfile.mkdtemp(prefix="cli_demo_")
    names = []
    for i in range(args.files):
        text = bench.gen_batch_torch(torch, genome, args.reads, L, 0.00234, i * args.reads, 7 + i, dev)
        with open(os.path.join(d, f"r{i}.fq"), "wb") as f:
            f.write(text.cpu().numpy().tobytes())
        names.append(f"r{i}.fq")
        del text
    with open(os.path.join(d, "files.txt"), "w") as f:
        f.write("\n".join(names) + "\n")
    del genome
    torch.cuda.empty_cache()
    nk = args.files * args.reads * (L - K + 1)
    cmd = [os.path.join(ROOT, "sh-assembly_amd", "bin", "CQF-deNoise"), "-k", str(K), "-N", str(nk), "-n", str(args.genome), "-e", "0.00234",
           "-f", "f", "-i", os.path.join(d, "files.txt"), "-o", os.path.join(d, "out.cqf")]
    subprocess.run(["cat"] + [os.path.join(d, n) for n in names], stdout=subprocess.DEVNULL)    # page cache warm, as for a re-run
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    print(json.dumps({"files": args.files, "reads_per_file": args.reads, "kmers": nk, "wall_s": dt, "kmers_per_s": nk / dt, "rc": r.returncode,
                      "stderr_tail": r.stderr.strip().split("\n")[-3:]}))

if __name__ == "__main__":
    main()
