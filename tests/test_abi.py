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
