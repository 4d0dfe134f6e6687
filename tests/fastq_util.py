"""helpers shared by the CPU (emulator) and GPU parity tests"""
import ctypes as C

import cqflibs


def chunks_by_records(fq: bytes, recs_per_chunk: int):
    """cut a well-formed FASTQ into chunks of `recs_per_chunk` records"""
    offs, lens, pos = [], [], 0
    lines = fq.split(b"\n")[:-1]
    for i in range(0, len(lines), 4 * recs_per_chunk):
        n = sum(len(x) + 1 for x in lines[i:i + 4 * recs_per_chunk])
        offs.append(pos)
        lens.append(n)
        pos += n
    return offs, lens


def oracle_t1(fq, offs, lens, k, qb, trigger=1 << 62, num_denoise=0, end_denoise=False, min_len=1 << 20):
    """the reference's t = 1 schedule over explicit chunks (cqf/CQF_mt.h:821-931) on the oracle:
    insert chunk, test the trigger, run a deNoise round when it fires"""
    O = cqflibs.oracle()
    q = O.new(qb)
    rounds = removed = 0
    left = num_denoise
    for a, n in zip(offs, lens):
        q.reads_to_kmers(fq[a:a + n], k)
        if left and q.ndistinct() >= trigger:
            left -= 1
            removed += q.denoise_round(min_len)
            rounds += 1
    if end_denoise:
        removed += q.denoise_round(min_len)
        rounds += 1
    return q, rounds, removed


def oracle_header(q):
    b = C.create_string_buffer(128)
    cqflibs.oracle().L.orc_qf_header(q.h, b)
    return b.raw
