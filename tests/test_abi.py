"""The product library (hipcc build) loads and exports every symbol include/shk.h declares.
No compute is attempted here (no GPU in the CPU suite)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_libshk_exports_every_declared_symbol():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "sh-assembly_amd")])
    import shk
    L = shk.load()
    hdr = open(os.path.join(ROOT, "include", "shk.h")).read()
    declared = set(re.findall(r"\b(shk_[a-z_]+)\s*\(", hdr))
    assert declared == set(shk.EXPORTS)
    for name in declared:
        assert getattr(L, name) is not None


def test_no_gpu_fails_loudly():
    """without a GPU shk_create must return an error, not fall back to anything"""
    import torch
    if torch.cuda.is_available():
        return
    import shk
    try:
        shk.Context(qb=10, k=21, max_batch_bytes=1 << 16, max_batch_keys=1 << 12)
    except shk.ShkError as e:
        assert e.code == -2
    else:
        raise AssertionError("shk_create succeeded without a GPU")


def test_libshkhost_exports_the_gqf_named_surface():
    """every function include/gqf_compat.h declares (the reference's own names, cqf/gqf.h:106-225) is exported by
    libshkhost.so, and a .cqf the reference wrote reads back through it: header fields, entries in iterator order and
    counts equal the oracle's reading of the same file"""
    import ctypes as C
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cqflibs
    from test_host_logic import _CompatQF
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "sh-assembly_amd"), os.path.join(ROOT, "sh-assembly_amd", "libshkhost.so")])
    L = C.CDLL(os.path.join(ROOT, "sh-assembly_amd", "libshkhost.so"))
    hdr = open(os.path.join(ROOT, "include", "gqf_compat.h")).read()
    declared = set(re.findall(r"^\s*(?:void|bool|int|uint64_t)\s+(\w+)\s*\(", hdr, flags=re.M))
    assert {"qf_init", "qf_insert_advance", "qf_count_key_value", "qfi_next", "qf_clean_singleton", "check_offset",
            "qf_count_key_value_set_traveled", "find_first_empty_slot", "qf_serialize", "qf_deserialize"} <= declared
    for name in declared:
        assert getattr(L, name) is not None
    path = os.path.join(ROOT, "tests", "golden", "build1.cqf")
    a = _CompatQF(L, path=path)
    o = cqflibs.oracle().load(path)
    assert a.dump() == o.dump() and a.blocks() == o.blocks() and L.check_offset(a.p)
    for key, c in o.dump()[:200]:
        assert a.count(key) == c
    a.free()
    o.free()
