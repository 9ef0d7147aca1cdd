"""CPU: the two SIMD loops of the host parsers (aln_db.cpp classifyColumns = parseDiff, ParseAlignTools.cpp:8-26; seq_db.cpp packBases =
CompressedSeq's packing, CompressedSeq.cpp:8-38) against their scalar restatements — every length 0..200 and random longer ones,
alphabets with gaps, lower case, NULs and the characters one bit away from C / G / T, reference rows shorter and longer than the
query row."""
import ctypes as C
import os

import numpy as np

import aligngraph2_amd
import pagctl


def _host():
    aligngraph2_amd.load_hip()  # (the host library links the C-ABI library: loaded first, by its path)
    lib = C.CDLL(os.path.join(pagctl.ROOT, "aligngraph2_amd", "libpagraph_host.so"))
    lib.pagh_debug_classify_columns.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.pagh_debug_classify_columns.restype = None
    lib.pagh_debug_pack_bases.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_int]
    lib.pagh_debug_pack_bases.restype = None
    return lib


def test_column_classes_wide_equal_scalar():
    lib = _host()
    rng = np.random.default_rng(11)
    alphabet = np.frombuffer(b"ACGTacgt-N\x00-", dtype=np.uint8)
    lengths = list(range(0, 201)) + [int(x) for x in rng.integers(201, 20000, size=60)]
    for n in lengths:
        q = bytes(rng.choice(alphabet, size=n))
        for rn in {n, max(0, n - 1), max(0, n - 33), n + 5, n // 2}:
            flips = rng.random(rn) < 0.3
            base = np.frombuffer((q + b"A" * rn)[:rn], dtype=np.uint8)
            r = bytes(np.where(flips, rng.choice(alphabet, size=rn), base).astype(np.uint8))
            got = []
            for scalar in (0, 1):
                w = np.zeros((n + 15) // 16 + 2, dtype=np.uint32)
                e, a = C.c_uint32(), C.c_uint32()
                lib.pagh_debug_classify_columns(q, n, r, rn, w.ctypes.data, C.byref(e), C.byref(a), scalar)
                got.append((w.tobytes(), e.value, a.value))
            assert got[0] == got[1], f"{n} columns, reference row of {rn}"


def test_packed_bases_wide_equal_scalar():
    lib = _host()
    rng = np.random.default_rng(12)
    alphabet = np.frombuffer(b"ACGTacgtNnSsWwBbDd\x03\x23\x47\x67-*", dtype=np.uint8)
    lengths = list(range(0, 201)) + [int(x) for x in rng.integers(201, 50000, size=60)]
    for n in lengths:
        sq = bytes(rng.choice(alphabet, size=n))
        got = []
        for scalar in (0, 1):
            out = np.zeros((n + 3) // 4 + 8, dtype=np.uint8)
            lib.pagh_debug_pack_bases(sq, n, out.ctypes.data, scalar)
            got.append(out.tobytes())
        assert got[0] == got[1], f"{n} bases"
