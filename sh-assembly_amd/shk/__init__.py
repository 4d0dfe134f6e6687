"""ctypes binding of libshk.so (include/shk.h) for tests, bench.py and smoke().

This is plumbing, not the product: the product is the C ABI and the C++ host tools in
sh-assembly_amd/host. The binding opens the in-tree libshk.so built by hipcc and fails
loudly when it is missing -- there is no CPU fallback. (The CPU test-suite passes the
path of the emulator build explicitly; see tests/emu.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libshk.so")

MAX_CHUNKS = 4096


class Config(C.Structure):
    _fields_ = [("qb", C.c_uint32), ("hb", C.c_uint32), ("seed", C.c_uint32), ("k", C.c_uint32),
                ("ndistinct_for_denoise", C.c_uint64), ("num_denoise", C.c_uint32), ("reserved0", C.c_uint32),
                ("min_denoise_len", C.c_uint64), ("max_batch_bytes", C.c_uint64), ("max_batch_keys", C.c_uint64),
                ("max_batch_reads", C.c_uint64), ("device", C.c_int32), ("shard_index", C.c_uint32),
                ("num_shards", C.c_uint32), ("threads_per_group", C.c_uint32), ("hash_groups", C.c_uint32),
                ("max_level_bits", C.c_uint32)]


class BatchStats(C.Structure):
    _fields_ = [("kmers", C.c_uint64), ("new_distinct", C.c_uint64), ("removed", C.c_uint64),
                ("denoise_rounds", C.c_uint32), ("chunks", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Totals(C.Structure):
    _fields_ = [("nelts", C.c_uint64), ("ndistinct", C.c_uint64), ("rounds_left", C.c_uint32),
                ("rounds_done", C.c_uint32), ("nslots", C.c_uint64), ("xnslots", C.c_uint64),
                ("nblocks", C.c_uint64), ("table_bytes", C.c_uint64), ("free_pointer", C.c_uint64)]


class Summary(C.Structure):
    _fields_ = [("new_distinct", C.c_uint64), ("added", C.c_uint64), ("removed", C.c_uint64), ("before", C.c_uint64),
                ("hist", C.c_uint64 * 32), ("err_bits", C.c_uint32), ("reserved", C.c_uint32)]


class Point(C.Structure):
    """shk_point (include/shk.h): one shard's share of a one-pass deNoise point"""
    _fields_ = [("new_after", C.c_uint64), ("added_after", C.c_uint64), ("removed", C.c_uint64), ("added_before", C.c_uint64),
                ("islots", C.c_uint64), ("ifin", C.c_uint64), ("first_used", C.c_uint32), ("err_bits", C.c_uint32)]


SOFT_BITS = 0x0A
HASH_FULL_BIT = 0x04
LOOKBACK_BIT = 0x100
HIST_BINS = 32


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char_p), ("launches", C.c_uint64), ("ms", C.c_double)]


class ShkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libshk error {code}: {msg}")
        self.code = code


EXPORTS = ["shk_create", "shk_destroy", "shk_count_chunks", "shk_hash_chunks", "shk_count_words", "shk_route_words", "shk_stage_words",
           "shk_stage_summary", "shk_stage_commit", "shk_stage_try", "shk_stage_try_denoise", "shk_stage_accept", "shk_stage_chunk_hist", "shk_stage_sample", "shk_stage_point_try", "shk_stage_round_try", "shk_stage_point_walk", "shk_stage_point_finish", "shk_upload_text", "shk_prepare_chunks", "shk_count_prepared", "shk_prepare_reserve", "shk_stage_words_pair", "shk_route_reserve", "shk_hash_route_chunks", "shk_host_alloc", "shk_host_free", "shk_extend_forward", "shk_unitigs_from_seeds", "shk_find_unitigs", "shk_unitig_set_new", "shk_unitig_set_free",
           "shk_unitigs_add_seeds", "shk_unitig_set_write", "shk_select_seeds", "shk_denoise",
           "shk_stats", "shk_header", "shk_export_blocks", "shk_export_cqf", "shk_import_cqf", "shk_import_blocks",
           "shk_lookup", "shk_profile_enable", "shk_profile_get", "shk_profile_reset", "shk_strerror",
           "shk_last_error_bits", "shk_insert_counted", "shk_dump", "shk_merge", "shk_multi_merge", "shk_import_shards", "shk_table_ptr", "shk_unitigs_add_reads"]

_libs = {}


def load(path=None):
    path = path or LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: build it with `make -C sh-assembly_amd` (hipcc, gfx950). "
                          "There is no CPU fallback.")
    L = C.CDLL(path)
    u64, u32, i32, vp = C.c_uint64, C.c_uint32, C.c_int, C.c_void_p
    pu64 = C.POINTER(u64)
    L.shk_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.shk_destroy.argtypes = [vp]
    L.shk_destroy.restype = None
    L.shk_count_chunks.argtypes = [vp, vp, i32, u64, pu64, pu64, u32, C.POINTER(BatchStats)]
    L.shk_hash_chunks.argtypes = [vp, vp, i32, u64, pu64, pu64, u32, C.POINTER(vp), pu64]
    L.shk_upload_text.argtypes = [vp, vp, u64, C.POINTER(vp)]
    L.shk_count_words.argtypes = [vp, vp, u64, u32, C.POINTER(BatchStats)]
    L.shk_route_words.argtypes = [vp, u64, u32, C.POINTER(vp), pu64]
    L.shk_stage_words.argtypes = [vp, vp, u64]
    L.shk_stage_summary.argtypes = [vp, u32, u32, u32, u32, i32, C.POINTER(Summary)]
    L.shk_stage_commit.argtypes = [vp, u32, u32, C.POINTER(Summary)]
    L.shk_stage_try.argtypes = [vp, u32, u32, u32, u32, i32, C.POINTER(Summary)]
    L.shk_stage_accept.argtypes = [vp, C.POINTER(Summary)]
    L.shk_stage_try_denoise.argtypes = [vp, u32, u32, C.POINTER(Summary)]
    L.shk_stage_chunk_hist.argtypes = [vp, C.POINTER(u64), u32]
    L.shk_stage_sample.argtypes = [vp, u32, u32, C.POINTER(u64), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.shk_stage_point_try.argtypes = [vp, u32, u32, u32, C.POINTER(Point)]
    L.shk_stage_round_try.argtypes = [vp, C.POINTER(Point)]
    L.shk_stage_point_walk.argtypes = [vp, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.POINTER(u32)]
    L.shk_stage_point_finish.argtypes = [vp, C.POINTER(Point), C.POINTER(Summary)]
    L.shk_extend_forward.argtypes = [vp, C.c_char_p, C.c_char_p, u32, u32, u64, i32, u32, C.c_char_p, C.POINTER(u32),
                                     C.POINTER(u32), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(u32)]
    L.shk_find_unitigs.argtypes = [vp, C.c_char_p, C.POINTER(u32), u32, u32, u64, u32, C.c_char_p, vp]
    L.shk_unitigs_from_seeds.argtypes = [vp, C.c_char_p, C.POINTER(u32), u32, u32, u64, u32, C.c_char_p, C.POINTER(u32),
                                         C.POINTER(C.c_int32), C.POINTER(C.c_uint8)]
    L.shk_denoise.argtypes = [vp, pu64]
    L.shk_stats.argtypes = [vp, C.POINTER(Totals)]
    L.shk_header.argtypes = [vp, C.c_char_p]
    L.shk_export_blocks.argtypes = [vp, vp, u64]
    L.shk_export_cqf.argtypes = [vp, C.c_char_p]
    L.shk_import_cqf.argtypes = [vp, C.c_char_p]
    L.shk_import_blocks.argtypes = [vp, vp, u64, u64, u64]
    L.shk_lookup.argtypes = [vp, vp, u64, i32, i32, vp, vp]
    L.shk_insert_counted.argtypes = [vp, vp, vp, u64, i32, C.POINTER(BatchStats)]
    L.shk_dump.argtypes = [vp, vp, vp, u64, i32, i32, pu64]
    L.shk_merge.argtypes = [vp, vp, C.POINTER(BatchStats)]
    L.shk_multi_merge.argtypes = [vp, C.POINTER(vp), u32, C.POINTER(BatchStats)]
    L.shk_import_shards.argtypes = [vp, C.POINTER(vp), pu64, u32, i32, u64, u64]
    L.shk_table_ptr.argtypes = [vp, C.POINTER(vp), pu64]
    L.shk_profile_enable.argtypes = [vp, i32]
    L.shk_profile_get.argtypes = [vp, C.POINTER(KernelTime), i32]
    L.shk_profile_reset.argtypes = [vp]
    L.shk_strerror.argtypes = [i32]
    L.shk_strerror.restype = C.c_char_p
    L.shk_last_error_bits.argtypes = [vp]
    L.shk_last_error_bits.restype = u32
    _libs[path] = L
    return L


class UnitigSet:
    """shk_unitig_set: the unitigs found so far, kept on the device of the context that feeds it"""

    def __init__(self, ctx):
        self.ctx, self.L = ctx, ctx.L
        self.L.shk_unitig_set_new.restype = C.c_void_p
        self.L.shk_unitig_set_free.argtypes = [C.c_void_p]
        self.L.shk_unitigs_add_seeds.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32,
                                                 C.c_uint64, C.c_uint32, C.c_int]
        self.L.shk_unitigs_add_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.POINTER(C.c_uint64),
                                                 C.POINTER(C.c_uint64), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64,
                                                 C.c_uint32, C.POINTER(C.c_uint64)]
        self.L.shk_unitig_set_write.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_void_p]
        self.h = C.c_void_p(self.L.shk_unitig_set_new())

    def add_seeds(self, seeds, counts, k, abundance_min, max_len, mark_traveled=True):
        n = len(seeds)
        sc = (C.c_uint32 * max(n, 1))(*counts)
        self.ctx._chk(self.L.shk_unitigs_add_seeds(self.ctx.h, self.h, b"".join(seeds), sc, n, k, abundance_min, max_len,
                                                   1 if mark_traveled else 0))

    def add_reads(self, text, chunk_off, chunk_len, k, abundance_min, count_min, count_max, max_len, text_bytes=None):
        """seeds of the reads in the FASTQ chunks + their walks; returns the number of seeds taken.
        text: bytes, or an integer device pointer with text_bytes given"""
        n = C.c_uint64()
        if isinstance(text, int):
            self.ctx._chk(self.L.shk_unitigs_add_reads(self.ctx.h, self.h, C.c_void_p(text), 1, int(text_bytes), self.ctx._tab(chunk_off),
                                                       self.ctx._tab(chunk_len), len(chunk_off), k, abundance_min, count_min, count_max,
                                                       max_len, C.byref(n)))
            return n.value
        buf = (C.c_char * len(text)).from_buffer_copy(text)
        self.ctx._chk(self.L.shk_unitigs_add_reads(self.ctx.h, self.h, C.cast(buf, C.c_void_p), 0, len(text), self.ctx._tab(chunk_off),
                                                   self.ctx._tab(chunk_len), len(chunk_off), k, abundance_min, count_min, count_max,
                                                   max_len, C.byref(n)))
        return n.value

    def write(self, k, path):
        st = (C.c_uint64 * 6)()
        self.ctx._chk(self.L.shk_unitig_set_write(self.h, k, path.encode(), C.cast(st, C.c_void_p)))
        return dict(zip(("unitigs", "total_len", "rounds", "extensions", "duplicates", "truncated"), list(st)))

    def close(self):
        if self.h:
            self.L.shk_unitig_set_free(self.h)
            self.h = None


def fixed_chunks(total_bytes, part_size):
    """helper: chunk table for a buffer cut every `part_size` bytes (caller guarantees record boundaries)"""
    offs = list(range(0, total_bytes, part_size))
    lens = [min(part_size, total_bytes - o) for o in offs]
    return offs, lens


class Context:
    """One filter (or one shard of it) on one GPU."""

    def __init__(self, qb, k, trigger=(1 << 62), num_denoise=0, min_denoise_len=0, max_batch_bytes=1 << 26,
                 max_batch_keys=1 << 26, max_batch_reads=0, device=0, shard_index=0, num_shards=1, seed=2038074761,
                 threads_per_group=0, hash_groups=0, max_level_bits=0, lib_path=None):
        self.L = load(lib_path)
        self.lib_path = lib_path
        self.cfg = Config(qb=qb, hb=qb + 8, seed=seed, k=k, ndistinct_for_denoise=trigger, num_denoise=num_denoise,
                          min_denoise_len=min_denoise_len, max_batch_bytes=max_batch_bytes,
                          max_batch_keys=max_batch_keys, max_batch_reads=max_batch_reads, device=device,
                          shard_index=shard_index, num_shards=num_shards, threads_per_group=threads_per_group,
                          hash_groups=hash_groups, max_level_bits=max_level_bits)
        h = C.c_void_p()
        self._chk(self.L.shk_create(C.byref(self.cfg), C.byref(h)))
        self.h = h

    def _chk(self, rc):
        if rc != 0:
            raise ShkError(rc, self.L.shk_strerror(rc).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.shk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _tab(vals):
        return (C.c_uint64 * len(vals))(*vals)

    def count_chunks(self, text, chunk_off, chunk_len, on_device=False, text_bytes=None):
        """text: bytes (host) or an integer device pointer (on_device=True, text_bytes given)"""
        st = BatchStats()
        if on_device or isinstance(text, int):      # an integer is a raw pointer (device, or host memory the caller keeps alive)
            ptr, n = C.c_void_p(int(text)), int(text_bytes)
        else:
            buf = (C.c_char * len(text)).from_buffer_copy(text) if not isinstance(text, C.Array) else text
            ptr, n = C.cast(buf, C.c_void_p), len(text)
        self._chk(self.L.shk_count_chunks(self.h, ptr, 1 if on_device else 0, n, self._tab(chunk_off),
                                          self._tab(chunk_len), len(chunk_off), C.byref(st)))
        return st.as_dict()

    def upload_text(self, host_ptr, nbytes):
        """start the copy of host text for a later count_chunks(..., on_device=True); returns the device pointer"""
        dp = C.c_void_p()
        self._chk(self.L.shk_upload_text(self.h, C.c_void_p(int(host_ptr)), int(nbytes), C.byref(dp)))
        return dp.value

    def prepare_chunks(self, text, chunk_off, chunk_len, on_device=False, text_bytes=None):
        """start the front end (parse, hash, partition) of a batch on the context's second stream; returns at once.
        `text` must stay alive until count_prepared() has returned for this batch (host bytes are kept here)."""
        if on_device:
            ptr, n = C.c_void_p(int(text)), int(text_bytes)
        else:
            buf = (C.c_char * len(text)).from_buffer_copy(text)
            if not hasattr(self, "_prepared_bufs"):
                self._prepared_bufs = []
            self._prepared_bufs.append(buf)
            ptr, n = C.cast(buf, C.c_void_p), len(text)
        self.L.shk_prepare_chunks.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint32]
        self._chk(self.L.shk_prepare_chunks(self.h, ptr, 1 if on_device else 0, n, self._tab(chunk_off), self._tab(chunk_len), len(chunk_off)))

    def prepare_reserve(self):
        """allocate the front end's buffers now rather than in the first prepare_chunks"""
        self.L.shk_prepare_reserve.argtypes = [C.c_void_p]
        self._chk(self.L.shk_prepare_reserve(self.h))

    def count_prepared(self):
        """the rebuild half for the oldest prepared batch; returns the same statistics as count_chunks"""
        st = BatchStats()
        self.L.shk_count_prepared.argtypes = [C.c_void_p, C.c_void_p]
        rc = self.L.shk_count_prepared(self.h, C.byref(st))
        if getattr(self, "_prepared_bufs", None):
            self._prepared_bufs.pop(0)
        self._chk(rc)
        return st.as_dict()

    def hash_route_chunks(self, text, chunk_off, chunk_len, nshards, on_device=False, text_bytes=None):
        """hash + bin by owner in one pass (shk_hash_route_chunks): returns (device pointer of the binned words, counts per shard, nwords)"""
        dp, nw = C.c_void_p(), C.c_uint64()
        counts = (C.c_uint64 * nshards)()
        if on_device:
            ptr, n = C.c_void_p(int(text)), int(text_bytes)
        else:
            buf = (C.c_char * len(text)).from_buffer_copy(text)
            ptr, n = C.cast(buf, C.c_void_p), len(text)
        self.L.shk_hash_route_chunks.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                                 C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        self._chk(self.L.shk_hash_route_chunks(self.h, ptr, 1 if on_device else 0, n, self._tab(chunk_off), self._tab(chunk_len),
                                               len(chunk_off), nshards, C.byref(dp), counts, C.byref(nw)))
        return dp.value, list(counts), nw.value

    def hash_chunks(self, text, chunk_off, chunk_len, on_device=False, text_bytes=None):
        """returns (device pointer, nwords)"""
        dp, nw = C.c_void_p(), C.c_uint64()
        if on_device:
            ptr, n = C.c_void_p(int(text)), int(text_bytes)
        else:
            buf = (C.c_char * len(text)).from_buffer_copy(text)
            ptr, n = C.cast(buf, C.c_void_p), len(text)
        self._chk(self.L.shk_hash_chunks(self.h, ptr, 1 if on_device else 0, n, self._tab(chunk_off),
                                         self._tab(chunk_len), len(chunk_off), C.byref(dp), C.byref(nw)))
        self._words_ptr = dp.value
        return dp.value, nw.value

    def count_words(self, d_words, nwords, nchunks=1):
        st = BatchStats()
        self._chk(self.L.shk_count_words(self.h, C.c_void_p(int(d_words) if d_words else 0), nwords, nchunks,
                                         C.byref(st)))
        return st.as_dict()

    def words_ptr(self):
        """pointer of the buffer shk_hash_chunks fills (valid until the next call)"""
        return self._words_ptr

    def route_words(self, nwords, nshards):
        """bin the words left by hash_chunks by owner; returns (device pointer, [count per owner])"""
        dp = C.c_void_p()
        cnt = (C.c_uint64 * nshards)()
        self._chk(self.L.shk_route_words(self.h, nwords, nshards, C.byref(dp), cnt))
        return dp.value, [cnt[i] for i in range(nshards)]

    def route_reserve(self):
        """allocate the send buffers of the routing now rather than in its first calls"""
        self.L.shk_route_reserve.argtypes = [C.c_void_p]
        self._chk(self.L.shk_route_reserve(self.h))

    def stage_words_pair(self, d_a, na, d_b, nb):
        """partition the words of TWO device buffers for the collective flow (shk_stage_words_pair)"""
        self.L.shk_stage_words_pair.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        self._chk(self.L.shk_stage_words_pair(self.h, C.c_void_p(int(d_a) if d_a else 0), na, C.c_void_p(int(d_b) if d_b else 0), nb))

    def stage_words(self, d_words, nwords):
        self._chk(self.L.shk_stage_words(self.h, C.c_void_p(int(d_words) if d_words else 0), nwords))

    def stage_summary(self, lo, hi, hist_base=0, hist_shift=0, want_hist=False):
        s = Summary()
        self._chk(self.L.shk_stage_summary(self.h, lo, hi, hist_base, hist_shift, int(want_hist), C.byref(s)))
        return s

    def stage_try(self, lo, hi, hist_base=0, hist_shift=0, want_hist=False):
        s = Summary()
        self._chk(self.L.shk_stage_try(self.h, lo, hi, hist_base, hist_shift, int(want_hist), C.byref(s)))
        return s

    def stage_chunk_hist(self, n):
        """exact per-chunk histogram of the last pass run with want_hist=2, or None when it has none"""
        out = (C.c_uint64 * max(n, 1))()
        if self.L.shk_stage_chunk_hist(self.h, out, n) != 0:
            return None
        return [out[i] for i in range(n)]

    def unitigs_from_seeds(self, seeds, seed_counts, k, abundance_min, max_len):
        """[(sequence, median abundance, (stop1, stop2))] -- one maximal unitig per seed k-mer (bytes of length k)"""
        n = len(seeds)
        sc = (C.c_uint32 * max(n, 1))(*seed_counts)
        out = C.create_string_buffer(max(n, 1) * max_len)
        ln = (C.c_uint32 * max(n, 1))()
        md = (C.c_int32 * max(n, 1))()
        st = (C.c_uint8 * (2 * max(n, 1)))()
        self._chk(self.L.shk_unitigs_from_seeds(self.h, b"".join(seeds), sc, n, k, abundance_min, max_len, out, ln, md, st))
        raw = out.raw
        return [(raw[i * max_len:i * max_len + ln[i]], md[i], (st[2 * i], st[2 * i + 1])) for i in range(n)]

    def find_unitigs(self, seeds, seed_counts, k, abundance_min, max_len, out_path):
        """writes the FASTA; returns dict(unitigs, total_len, rounds, extensions, duplicates, truncated)"""
        n = len(seeds)
        sc = (C.c_uint32 * max(n, 1))(*seed_counts)
        st = (C.c_uint64 * 6)()
        self._chk(self.L.shk_find_unitigs(self.h, b"".join(seeds), sc, n, k, abundance_min, max_len, out_path.encode(),
                                          C.cast(st, C.c_void_p)))
        return dict(zip(("unitigs", "total_len", "rounds", "extensions", "duplicates", "truncated"), list(st)))

    def stage_sample(self, lo, hi):
        """(hist[0..hi] of the sampled regions, regions, sampled regions, kernel flag bits)"""
        out = (C.c_uint64 * (hi + 1))()
        nr, ns, eb = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._chk(self.L.shk_stage_sample(self.h, lo, hi, out, C.byref(nr), C.byref(ns), C.byref(eb)))
        return list(out), nr.value, ns.value, eb.value

    def stage_point_try(self, lo, split, hi):
        p = Point()
        self._chk(self.L.shk_stage_point_try(self.h, lo, split, hi, C.byref(p)))
        return p

    def stage_round_try(self):
        p = Point()
        self._chk(self.L.shk_stage_round_try(self.h, C.byref(p)))
        return p

    def stage_point_walk(self, carry, prev_fp, last, next_first_used, state):
        """-> (state for the next shard, protected singletons on this shard, kernel flag bits)"""
        sin = (C.c_uint64 * 2)(*state)
        sout = (C.c_uint64 * 2)()
        n, eb = C.c_uint64(), C.c_uint32()
        self._chk(self.L.shk_stage_point_walk(self.h, carry, prev_fp, int(last), int(next_first_used), sin, sout, C.byref(n), C.byref(eb)))
        return (sout[0], sout[1]), n.value, eb.value

    def stage_point_finish(self, point):
        acc = Summary()
        self._chk(self.L.shk_stage_point_finish(self.h, C.byref(point), C.byref(acc)))
        return acc

    def stage_try_denoise(self, lo, hi):
        s = Summary()
        self._chk(self.L.shk_stage_try_denoise(self.h, lo, hi, C.byref(s)))
        return s

    def stage_accept(self, summary):
        self._chk(self.L.shk_stage_accept(self.h, C.byref(summary)))

    def stage_commit(self, lo, hi, summary):
        self._chk(self.L.shk_stage_commit(self.h, lo, hi, C.byref(summary)))

    def error_for_bits(self, bits):
        """raise the library's error for raw kernel flag bits (commit of an empty Summary does the mapping)"""
        s = Summary()
        s.err_bits = bits
        self._chk(self.L.shk_stage_commit(self.h, 0, 0, C.byref(s)))

    def denoise(self):
        r = C.c_uint64()
        self._chk(self.L.shk_denoise(self.h, C.byref(r)))
        return r.value

    def totals(self):
        t = Totals()
        self._chk(self.L.shk_stats(self.h, C.byref(t)))
        return t

    def header(self):
        b = C.create_string_buffer(128)
        self._chk(self.L.shk_header(self.h, b))
        return b.raw

    def blocks(self):
        n = self.totals().table_bytes
        b = C.create_string_buffer(n)
        self._chk(self.L.shk_export_blocks(self.h, C.cast(b, C.c_void_p), n))
        return b.raw

    def export_cqf(self, path):
        self._chk(self.L.shk_export_cqf(self.h, path.encode()))

    def import_cqf(self, path):
        self._chk(self.L.shk_import_cqf(self.h, path.encode()))

    def import_blocks(self, data, nelts=0, ndistinct=0):
        b = (C.c_char * len(data)).from_buffer_copy(data)
        self._chk(self.L.shk_import_blocks(self.h, C.cast(b, C.c_void_p), len(data), nelts, ndistinct))

    def lookup(self, keys, mode=2):
        n = len(keys)
        k = (C.c_uint64 * max(n, 1))(*keys)
        c = (C.c_uint64 * max(n, 1))()
        t = (C.c_uint8 * max(n, 1))()
        self._chk(self.L.shk_lookup(self.h, C.cast(k, C.c_void_p), n, 0, mode, C.cast(c, C.c_void_p),
                                    C.cast(t, C.c_void_p)))
        return [c[i] for i in range(n)], [t[i] for i in range(n)]

    def insert_counted(self, keys, counts):
        """add counts[i] occurrences of keys[i] (host lists); returns the batch statistics"""
        n = len(keys)
        k = (C.c_uint64 * max(n, 1))(*keys)
        c = (C.c_uint64 * max(n, 1))(*counts)
        st = BatchStats()
        self._chk(self.L.shk_insert_counted(self.h, C.cast(k, C.c_void_p), C.cast(c, C.c_void_p), n, 0, C.byref(st)))
        return st.as_dict()

    def dump(self, ref_iterator_end=False):
        """[(key, count)] in the order of the reference's iterator; ref_iterator_end: stop where its qfi_next stops"""
        n = C.c_uint64()
        self._chk(self.L.shk_dump(self.h, None, None, 0, 0, 0, C.byref(n)))
        m = n.value
        k = (C.c_uint64 * max(m, 1))()
        c = (C.c_uint64 * max(m, 1))()
        self._chk(self.L.shk_dump(self.h, C.cast(k, C.c_void_p), C.cast(c, C.c_void_p), m, 0, 1 if ref_iterator_end else 0,
                                  C.byref(n)))
        assert n.value <= m
        return [(k[i], c[i]) for i in range(n.value)]

    def merge(self, other):
        """self += other (qf_merge)"""
        st = BatchStats()
        self._chk(self.L.shk_merge(self.h, other.h, C.byref(st)))
        return st.as_dict()

    def multi_merge(self, others):
        st = BatchStats()
        arr = (C.c_void_p * max(len(others), 1))(*[o.h for o in others])
        self._chk(self.L.shk_multi_merge(self.h, arr, len(others), C.byref(st)))
        return st.as_dict()

    def import_shards(self, shard_blocks, nelts=0, ndistinct=0):
        """replace the table by the union of the shards' tables (bytes objects, shard order)"""
        n = len(shard_blocks)
        bufs = [(C.c_char * len(b)).from_buffer_copy(b) for b in shard_blocks]
        ptrs = (C.c_void_p * n)(*[C.cast(b, C.c_void_p) for b in bufs])
        sizes = (C.c_uint64 * n)(*[len(b) for b in shard_blocks])
        self._chk(self.L.shk_import_shards(self.h, ptrs, sizes, n, 0, nelts, ndistinct))

    def table_ptr(self):
        """(device pointer, bytes) of the live table"""
        p, n = C.c_void_p(), C.c_uint64()
        self._chk(self.L.shk_table_ptr(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def import_shards_device(self, ptrs, nbytes, nelts=0, ndistinct=0):
        """like import_shards, from tables that already live on this context's device (16 readable bytes behind each)"""
        n = len(ptrs)
        arr = (C.c_void_p * n)(*ptrs)
        sizes = (C.c_uint64 * n)(*([nbytes] * n))
        self._chk(self.L.shk_import_shards(self.h, arr, sizes, n, 1, nelts, ndistinct))

    def profile(self, on=True):
        self._chk(self.L.shk_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        self._chk(self.L.shk_profile_reset(self.h))

    def profile_get(self):
        arr = (KernelTime * 32)()
        n = self.L.shk_profile_get(self.h, arr, 32)
        return {arr[i].name.decode(): (arr[i].launches, arr[i].ms) for i in range(n) if arr[i].launches}

    def last_error_bits(self):
        return self.L.shk_last_error_bits(self.h)
