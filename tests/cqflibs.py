"""ctypes bindings for the two CPU checkers (test infrastructure):

  Oracle  -> oracle/liboracle.so       (this repo's plain-C restatement, prefix orc_)
  Ref     -> oracle/_ref/libshk_ref.so (the real reference gqf.c + nthash.hpp, prefix ref_)

Both expose the same calls so a test can run the same scenario through either.
"""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libshk_ref.so")
SEED = 2038074761  # src/CQF-deNoise.cpp:83


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")],
                          stdout=subprocess.DEVNULL)


class _QF:
    def __init__(self, lib, prefix, handle):
        self.L, self.p, self.h = lib, prefix, handle

    def _f(self, name):
        return getattr(self.L, self.p + name)

    def insert(self, key, count=1):
        return self._f("qf_insert")(self.h, key, count)

    def count(self, key):
        return self._f("qf_count")(self.h, key)

    def count_set_traveled(self, key):
        c = C.c_uint64(0)
        t = self._f("qf_count_set_traveled")(self.h, key, C.byref(c))
        return t, c.value

    def count_is_traveled(self, key):
        c = C.c_uint64(0)
        t = self._f("qf_count_is_traveled")(self.h, key, C.byref(c))
        return t, c.value

    def blocks(self):
        return C.string_at(self._f("qf_blocks")(self.h), self._f("qf_size")(self.h))

    def nelts(self):
        return self._f("qf_nelts")(self.h)

    def ndistinct(self):
        return self._f("qf_ndistinct")(self.h)

    def check_offset(self):
        return bool(self._f("qf_check_offset")(self.h))

    def denoise_round(self, min_len=1 << 20):
        return self._f("denoise_round_t1")(self.h, min_len)

    def dump(self):
        n = self._f("qf_dump")(self.h, None, None, 0)
        k = (C.c_uint64 * max(n, 1))()
        c = (C.c_uint64 * max(n, 1))()
        self._f("qf_dump")(self.h, k, c, n)
        return [(k[i], c[i]) for i in range(n)]

    def serialize(self, path):
        self._f("qf_serialize")(self.h, path.encode())

    def reads_to_kmers(self, chunk: bytes, k):
        self._f("reads_to_kmers")(self.h, chunk, len(chunk), k)

    def unitig_from_seed(self, seed: bytes, seed_count, k, abundance_min, max_len):
        """(sequence, median abundance, (stop1, stop2)): Contiger's two extensions of one seed (oracle only)"""
        out = C.create_string_buffer(max_len + 1)
        ln, md = C.c_uint32(), C.c_int()
        st = (C.c_uint8 * 2)()
        rc = self._f("unitig_from_seed")(self.h, seed, seed_count, k, abundance_min, max_len, out, C.byref(ln), C.byref(md), st)
        assert rc == 0
        return out.raw[:ln.value], md.value, (st[0], st[1])

    def contiger(self, text: bytes, offs, lens, k, abundance_min=2, solid_min=2, solid_max=1000000, rule=1, drain_per_read=True):
        """the whole of Contiger as one sequential program (oracle/contiger_pipeline.cpp; oracle only): returns
        (text of unitigs.fa, stats). Sets the filter's traveled bits like the reference does. rule: 0 = every chunk by the
        master's loop (seed at len/2), 1 = by processDataChunk (len/2 - K/2), 2 = alternating"""
        n = len(offs)
        o = (C.c_uint64 * max(n, 1))(*offs)
        ln = (C.c_uint64 * max(n, 1))(*lens)
        out_len = C.c_uint64()
        st = (C.c_uint64 * 6)()
        p = self.L.orc_contiger_run(self.h, text, o, ln, n, k, abundance_min, solid_min, solid_max,
                                    (rule << 1) | (0 if drain_per_read else 1), C.byref(out_len), st)
        fa = C.string_at(p, out_len.value)
        self.L.orc_contiger_free(p)
        return fa, dict(zip(("seeds", "queued", "cleared", "lookups", "contigs", "map_entries"), list(st)))

    def extend_forward(self, seq: bytes, median, k, abundance_min, max_len):
        """(sequence, median, stop, branch mask, neighbour counts[8]): one get_unitig_forward (oracle only)"""
        buf = C.create_string_buffer(seq, max_len + 1)
        ln, md, br = C.c_uint32(len(seq)), C.c_int(median), C.c_uint8()
        nc = (C.c_uint32 * 8)()
        st = self._f("extend_forward")(self.h, buf, C.byref(ln), k, abundance_min, max_len, C.byref(md), C.byref(br), nc)
        assert st >= 0
        return buf.raw[:ln.value], md.value, st, br.value, list(nc)

    def time_chunks_mt(self, text: bytes, offs, lens, k, nthreads, budget_s):
        """(seconds, k-mers inserted, chunks taken): `nthreads` threads insert under the reference's region locks"""
        n = len(offs)
        o = (C.c_uint64 * n)(*offs)
        ln = (C.c_uint64 * n)(*lens)
        km, ch = C.c_uint64(), C.c_uint32()
        dt = self._f("time_chunks_mt")(self.h, text, o, ln, n, k, nthreads, budget_s, C.byref(km), C.byref(ch))
        return dt, km.value, ch.value

    def find_first_empty_slot(self, frm):
        return getattr(self.L, self.p + "find_first_empty_slot")(self.h, frm)

    def find_first_nonempty_slot(self, frm):
        return getattr(self.L, self.p + "find_first_nonempty_slot")(self.h, frm)

    def build_t1(self, files, k, trigger, num_denoise, end_denoise=False,
                 part_size=1 << 23, overhead=65535, min_len=1 << 20):
        arr = (C.c_char_p * len(files))(*[f.encode() for f in files])
        st = (C.c_uint64 * 3)()
        self._f("build_t1")(self.h, arr, len(files), k, trigger, num_denoise,
                            1 if end_denoise else 0, part_size, overhead, min_len, st)
        return {"rounds": st[0], "removed": st[1], "chunks": st[2]}

    def merge_from(self, a, b):
        """self := a + b through the reference's qf_merge (ref only; self must be empty)"""
        self._f("qf_merge")(a.h, b.h, self.h)

    def multi_merge_from(self, qs):
        arr = (C.c_void_p * len(qs))(*[q.h for q in qs])
        self._f("qf_multi_merge")(arr, len(qs), self.h)

    def full(self):
        return bool(self.L.orc_qf_full(self.h)) if self.p == "orc_" else False

    def free(self):
        if self.h:
            self._f("qf_free")(self.h)
            self.h = None


class _Lib:
    def __init__(self, path, prefix):
        self.L = C.CDLL(path)
        self.p = prefix
        L, p = self.L, prefix
        u64, vp, i32, u32 = C.c_uint64, C.c_void_p, C.c_int, C.c_uint32

        def sig(name, res, args):
            f = getattr(L, p + name)
            f.restype, f.argtypes = res, args

        sig("nthash", None, [C.c_char_p, C.c_uint, C.POINTER(u64), C.POINTER(u64)])
        sig("nthash_roll", None, [C.c_ubyte, C.c_ubyte, C.c_uint, C.POINTER(u64), C.POINTER(u64)])
        sig("qf_new", vp, [u64, u64, u32])
        sig("qf_free", None, [vp])
        sig("qf_insert", i32, [vp, u64, u64])
        sig("qf_count", u64, [vp, u64])
        sig("qf_count_set_traveled", i32, [vp, u64, C.POINTER(u64)])
        sig("qf_count_is_traveled", i32, [vp, u64, C.POINTER(u64)])
        sig("qf_blocks", vp, [vp])
        sig("qf_size", u64, [vp])
        sig("qf_nelts", u64, [vp])
        sig("qf_ndistinct", u64, [vp])
        sig("qf_check_offset", i32, [vp])
        sig("denoise_round_t1", u64, [vp, u64])
        sig("qf_dump", u64, [vp, C.POINTER(u64), C.POINTER(u64), u64])
        sig("qf_serialize", None if p == "ref_" else i32, [vp, C.c_char_p])
        sig("qf_load", vp, [C.c_char_p])
        sig("reads_to_kmers", None, [vp, C.c_char_p, u64, C.c_uint])
        sig("find_first_empty_slot", u64, [vp, u64])
        sig("find_first_nonempty_slot", u64, [vp, u64])
        sig("chunk_sizes", u64, [C.c_char_p, u64, u32, C.POINTER(u64), u64])
        sig("build_t1", None, [vp, C.POINTER(C.c_char_p), i32, C.c_uint, u64, u32, i32,
                               u64, u32, u64, C.POINTER(u64)])
        if p == "orc_":
            sig("extend_forward", i32, [vp, C.c_char_p, C.POINTER(u32), C.c_uint, u64, u32, C.POINTER(i32), C.POINTER(C.c_uint8),
                                        C.POINTER(u32)])
            sig("unitig_from_seed", i32, [vp, C.c_char_p, u32, C.c_uint, u64, u32, C.c_char_p, C.POINTER(u32),
                                          C.POINTER(i32), C.POINTER(C.c_uint8)])
            L.orc_contiger_run.restype = C.c_void_p
            L.orc_contiger_run.argtypes = [vp, C.c_char_p, C.POINTER(u64), C.POINTER(u64), u32, C.c_uint, u64, u64, u64, u32,
                                           C.POINTER(u64), C.POINTER(u64)]
            L.orc_contiger_free.argtypes = [vp]
        if p == "ref_":
            sig("encode_counter", i32, [vp, u64, u64, C.POINTER(u64)])
            if hasattr(L, "ref_qf_merge"):
                sig("qf_merge", None, [vp, vp, vp])
                sig("qf_multi_merge", None, [C.POINTER(vp), i32, vp])
            if hasattr(L, "ref_time_chunks_mt"):   # (a prebuilt oracle/_ref from before this entry existed lacks it)
                sig("time_chunks_mt", C.c_double, [vp, C.c_char_p, C.POINTER(u64), C.POINTER(u64), u32, C.c_uint, C.c_uint,
                                                   C.c_double, C.POINTER(u64), C.POINTER(u32)])
        else:
            sig("encode_counter", i32, [u64, u64, C.POINTER(u64)])
            sig("chunk_keys", u64, [C.c_char_p, u64, C.c_uint, u64, C.POINTER(u64), u64])
            sig("mean_cdf2denoise", i32, [C.c_double, C.c_double])
            sig("qf_header", None, [vp, C.c_char_p])
            sig("qf_full", i32, [vp])

    def new(self, qb, hb=None, seed=SEED):
        h = getattr(self.L, self.p + "qf_new")(qb, hb if hb is not None else qb + 8, seed)
        return _QF(self.L, self.p, h)

    def load(self, path):
        return _QF(self.L, self.p, getattr(self.L, self.p + "qf_load")(path.encode()))

    def nthash(self, seq: bytes, k):
        a, b = C.c_uint64(0), C.c_uint64(0)
        getattr(self.L, self.p + "nthash")(seq, k, C.byref(a), C.byref(b))
        return a.value, b.value

    def nthash_roll(self, out, inn, k, fh, rh):
        a, b = C.c_uint64(fh), C.c_uint64(rh)
        getattr(self.L, self.p + "nthash_roll")(out, inn, k, C.byref(a), C.byref(b))
        return a.value, b.value

    def chunk_sizes(self, path, part_size=1 << 23, overhead=65535):
        n = getattr(self.L, self.p + "chunk_sizes")(path.encode(), part_size, overhead, None, 0)
        arr = (C.c_uint64 * max(n, 1))()
        getattr(self.L, self.p + "chunk_sizes")(path.encode(), part_size, overhead, arr, n)
        return [arr[i] for i in range(n)]

    def encode_counter(self, rem, count, qf=None):
        out = (C.c_uint64 * 70)()
        if self.p == "ref_":
            n = self.L.ref_encode_counter(qf.h, rem, count, out)
        else:
            n = self.L.orc_encode_counter(rem, count, out)
        return [out[i] for i in range(n)]

    def seq_keys(self, seqs, k, hb):
        """filter keys (canonical ntHash mod 2^hb) of every k-mer of every sequence, as a numpy uint64 array in order"""
        import numpy as np
        assert self.p == "orc_"
        text = b"".join(b"@\n" + s + b"\n+\n\n" for s in seqs)
        n = sum(max(0, len(s) - k + 1) for s in seqs)
        out = np.zeros(max(n, 1), dtype=np.uint64)
        got = self.L.orc_chunk_keys(text, len(text), k, hb, out.ctypes.data_as(C.POINTER(C.c_uint64)), n)
        assert got == n, (got, n)
        return out[:n]

    def chunk_keys(self, chunk: bytes, k, hb):
        assert self.p == "orc_"
        n = self.L.orc_chunk_keys(chunk, len(chunk), k, hb, None, 0)
        arr = (C.c_uint64 * max(n, 1))()
        self.L.orc_chunk_keys(chunk, len(chunk), k, hb, arr, n)
        return [arr[i] for i in range(n)]


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        _oracle = _Lib(ORACLE_SO, "orc_")
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        _ref = _Lib(REF_SO, "ref_")
    return _ref
