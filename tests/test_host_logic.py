"""CPU tests of host-side logic that bench.py and the C ABI rely on."""
import importlib.util
import os
import sys

import cqflibs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_chunk_table_matches_chunker(tmp_path):
    """bench.py's closed form of fastq_read_parts for fixed-size records == the oracle's chunker"""
    b = _bench()
    O = cqflibs.oracle()
    for L, nrec, part, ov in [(150, 700, 20000, 4095), (100, 1000, 9000, 2047), (150, 300, 1 << 23, 65535),
                              (60, 4000, 1 << 16, 8191), (150, 630, 19970, 4095)]:
        rec = 2 * L + b.NAME_W + 6
        one = b"@" + b"0" * b.NAME_W + b"\n" + b"A" * L + b"\n+\n" + b"I" * L + b"\n"
        assert len(one) == rec
        p = tmp_path / "f.fq"
        p.write_bytes(one * nrec)
        offs, lens = b.chunk_table(nrec, rec, part, ov)
        ref = [x for x in O.chunk_sizes(str(p), part, ov) if x]
        assert lens == ref
        assert offs == [sum(lens[:i]) for i in range(len(lens))]
        assert sum(lens) == nrec * rec


def test_sizing_matches_oracle():
    import ctypes as C
    b = _bench()

    class S(C.Structure):
        _fields_ = [("num_true", C.c_uint64), ("num_false", C.c_uint64), ("qb", C.c_uint64), ("hb", C.c_uint64),
                    ("nd", C.c_int), ("trigger", C.c_uint64), ("lb", C.c_int), ("ub", C.c_int)]
    O = cqflibs.oracle()
    O.L.orc_size_filter.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.c_int, C.c_double,
                                    C.POINTER(S)]
    for K, n, N, e in [(47, 119157843, 16506371070, 0.00234), (28, 4600000, 123000000, 0.01),
                       (31, 2900000000, 90000000000, 0.005)]:
        s = S()
        O.L.orc_size_filter(K, n, N, e, 0.0, -1, 0.0, C.byref(s))
        qb, nd, trig = b.sizing(K, n, N, e)
        assert (qb, nd, trig) == (s.qb, s.nd, s.trigger)


def _hostlib():
    import ctypes as C
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "sh-assembly_amd"),
                           os.path.join(ROOT, "sh-assembly_amd", "libshkhost.so")])
    L = C.CDLL(os.path.join(ROOT, "sh-assembly_amd", "libshkhost.so"))
    L.shkh_chunk_sizes.restype = C.c_uint64
    L.shkh_chunk_sizes.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_uint64, C.c_uint32,
                                   C.POINTER(C.c_uint64), C.c_uint64]
    L.shkh_size_filter.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_double, C.c_int, C.c_double,
                                   C.POINTER(C.c_uint64)]
    return L


def test_cpp_chunker_matches_oracle(tmp_path):
    """sh-assembly_amd/host/fastq_chunker.cpp (seqFile_batch) against the oracle chunker and the
    golden chunk sizes; round robin over two files; gzip == bzip2 == plain"""
    import ctypes as C
    import gzip
    import json
    L = _hostlib()
    O = cqflibs.oracle()
    G = os.path.join(ROOT, "tests", "golden")
    fx = json.load(open(os.path.join(G, "fastq_builds.json")))

    def sizes(paths, mode, ps, ov):
        arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
        out = (C.c_uint64 * 4096)()
        n = L.shkh_chunk_sizes(arr, len(paths), mode, ps, ov, out, 4096)
        return [out[i] for i in range(n)]
    for key, golden in fx["chunks"].items():
        f, ps, ov = key.split(":")
        assert sizes([os.path.join(G, f)], 0, int(ps), int(ov)) == golden
    f0, f1 = os.path.join(G, "reads0.fq"), os.path.join(G, "reads1.fq")
    a, b = O.chunk_sizes(f0, 20000, 4095), O.chunk_sizes(f1, 20000, 4095)
    inter = []
    for i in range(max(len(a), len(b))):
        if i < len(a):
            inter.append(a[i])
        if i < len(b):
            inter.append(b[i])
    assert sizes([f0, f1], 0, 20000, 4095) == inter
    gz = str(tmp_path / "r0.fq.gz")
    with gzip.open(gz, "wb") as g:
        g.write(open(f0, "rb").read())
    assert sizes([gz], 1, 20000, 4095) == a
    import bz2
    bz = str(tmp_path / "r0.fq.bz2")
    with bz2.open(bz, "wb") as g:
        g.write(open(f0, "rb").read())
    assert sizes([bz], 2, 20000, 4095) == a          # bzip2 through libbz2's BZ2_bzRead, as the reference reads it


def test_cpp_sizing_matches_oracle():
    import ctypes as C
    L = _hostlib()
    b = _bench()
    for K, n, N, e in [(47, 119157843, 16506371070, 0.00234), (28, 4600000, 123000000, 0.01)]:
        out = (C.c_uint64 * 8)()
        L.shkh_size_filter(K, n, N, e, -1, 0.0, out)
        assert (out[0], out[2], out[3]) == b.sizing(K, n, N, e)
