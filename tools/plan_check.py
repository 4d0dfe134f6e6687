#!/usr/bin/env python3
"""Check shk.plan.predict_build against the oracle: the bench's workload scaled down by `--scale`
(genome, reads per step and chunk size divided; the sizing arithmetic applied to the scaled numbers),
run chunk by chunk through the t = 1 schedule on oracle/liboracle.so; prints predicted vs measured
occupied slots per step. Test infrastructure (uses the oracle)."""
import argparse
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sh-assembly_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def used_slots(q):
    """slots in use = sum of the entries' encoded lengths (encode_counter, gqf.c:1225-1255)"""
    d = q.dump()
    if not d:
        return 0
    keys = np.array([k for k, _ in d], dtype=np.uint64)
    cnt = np.array([c for _, c in d], dtype=np.uint64)
    rem = (keys & np.uint64(0xFF)).astype(np.int64)
    c = cnt.astype(np.int64) - 1
    nd = np.where(c > 0, 1, 0) + (c >= 128) + (c >= 128 ** 2) + (c >= 128 ** 3)
    top = np.where(nd > 0, (c >> (7 * np.maximum(nd - 1, 0))) & 0x7F, 0) | np.where(nd > 1, 0x80, 0)
    esc = (nd > 0) & (top > rem)
    return int((1 + nd + esc).sum())


def run(scale, steps, head=1.0, R0=8_000_000, G0=119_157_843, verbose=True, rounds=None, trigger=None, profile=None):
    """profile: None = the bench's uniform substitution rate; (lo, hi) = rates rising linearly along the read from lo to hi
    (sequencing errors concentrate at the read end, as an --errorProfile file describes); the mean stays ERR"""
    import torch
    import cqflibs
    from shk import plan
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    K, L, ERR = 47, 150, 0.00234
    G, R = G0 // scale, R0 // scale
    rec = 2 * L + bench.NAME_W + 6
    part = max(rec * 4, (1 << 23) // scale)
    offs, lens = bench.chunk_table(R, rec, part, min(65535, part // 4))
    S = R * (L - K + 1)
    N = int(steps * S * head)
    qb, nd, trig0 = plan.sizing(K, G - K + 1, N, ERR)
    if rounds is not None:
        nd = rounds
    trigger = trigger or trig0
    pred = plan.predict_build(K, G, L, ERR, S / len(offs), len(offs) * steps, trigger, nd, trace_every=len(offs))
    dev = torch.device("cpu")
    err_model = ERR if profile is None else torch.linspace(profile[0], profile[1], L, dtype=torch.float32)
    if profile is not None:
        assert abs(float(err_model.mean()) - ERR) < 1e-5
    genome = torch.randint(0, 4, (G,), dtype=torch.uint8, generator=torch.Generator().manual_seed(2))
    O = cqflibs.oracle()
    q = O.new(qb)
    left, fired, peak, full_at, per_step = nd, 0, 0, None, []
    for s in range(steps):
        text = bench.gen_batch_torch(torch, genome, R, L, err_model, s * R, 1000 + s, dev).numpy().tobytes()
        for a, n in zip(offs, lens):
            q.reads_to_kmers(text[a:a + n], K)
            if q.full() and full_at is None:
                full_at = s
            if left and q.ndistinct() >= trigger:
                u = used_slots(q)
                peak = max(peak, u)
                left -= 1
                fired += 1
                q.denoise_round(1 << 20)
        if full_at is not None:
            break
        u = used_slots(q)
        peak = max(peak, u)
        per_step.append((u, q.ndistinct(), fired))
        if verbose:
            print(f"step {s}: used {u} ({u / (1 << qb):.4f}) distinct {q.ndistinct()} rounds {fired}", flush=True)
    out = {"qb": qb, "rounds": nd, "trigger": trigger, "fired": fired, "peak": peak, "peak_load": peak / (1 << qb),
           "pred_peak_load": pred["peak_slots"] / (1 << qb), "pred_fired": pred["rounds_fired"], "full_at_step": full_at,
           "per_step": per_step, "pred_per_step": [(int(a), int(b), c) for a, b, c in pred["trace"]]}
    q.free()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--head", type=float, default=1.0)
    a = ap.parse_args()
    print(run(a.scale, a.steps, a.head))
