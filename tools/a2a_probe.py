#!/usr/bin/env python3
"""One-rank RCCL probe for the exchange step: does an all-to-all of N 8-byte key words deliver all of them?
Round 1 saw "only the first 832 MB of ~1.6 GB per peer" arrive and bounded its pieces to 2^25 words without finding
the cause. This sends arange patterns of 2^25 .. 2^29 words (256 MB .. 4 GB) through both forms shk/dist.py can use
(grouped all_to_all on views, all_to_all_single with split sizes), from a torch allocation and from a raw pointer
wrapped through __cuda_array_interface__ (how the library's send buffer reaches torch), and compares every word."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sh-assembly_amd"))
from shk import dist as shkdist  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29877")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
dev = torch.device("cuda:0")
out = []
for lg in (25, 27, 28, 29):
    n = 1 << lg
    src = torch.arange(n, dtype=torch.int64, device=dev) * 2654435761 + 12345
    view = shkdist.wrap_words(src.data_ptr(), n, dev)          # the library's pointer path
    for name, inp in (("tensor", src), ("cai_view", view)):
        dst = torch.zeros(n, dtype=torch.int64, device=dev)
        dist.all_to_all([dst], [inp])
        torch.cuda.synchronize()
        ok1 = bool(torch.equal(dst, src))
        bad1 = int((dst != src).sum().item())
        dst.zero_()
        dist.all_to_all_single(dst, inp, output_split_sizes=[n], input_split_sizes=[n])
        torch.cuda.synchronize()
        ok2 = bool(torch.equal(dst, src))
        bad2 = int((dst != src).sum().item())
        out.append({"words": n, "bytes": n * 8, "source": name, "all_to_all_ok": ok1, "all_to_all_wrong_words": bad1,
                    "all_to_all_single_ok": ok2, "all_to_all_single_wrong_words": bad2})
        print(out[-1], flush=True)
        del dst
    del src, view
    torch.cuda.empty_cache()
print(json.dumps(out))
dist.destroy_process_group()
