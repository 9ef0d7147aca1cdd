"""CPU: function-level golden tables written by the compiled reference (oracle/ref_harness/func_golden.cpp ->
tests/golden/func_mapper.txt, func_edit.txt) against the product's own copies of those functions:

  PositionMapper (position/PositionMapper.cpp:16-64)   csrc/host/position_mapper.hpp  (graph input, writers)
                                                       Mapper in csrc/hip/k5_travel_host.hip (traversal control)
  PAlgorithm::editDistance (PAlgorithm.cpp:46-69)      edit_distance in csrc/hip/k5_travel_host.hip (seed ordering)

The hooks are plain host functions; no GPU is needed to call them."""
import ctypes as C
import os

import numpy as np
import pytest

import goldens
import pagctl

LENS = np.array([100, 37, 250], dtype=np.uint32)  # the three sequences of func_golden's "mapper" table


def _hip_host_only():
    if not os.path.exists(pagctl.HIP_LIB):
        pytest.skip("libpagraph_hip.so not built")
    lib = C.CDLL(pagctl.HIP_LIB)
    lib.pag_debug_edit_distance.argtypes = [C.c_char_p, C.c_char_p]
    lib.pag_debug_edit_distance.restype = C.c_uint64
    return lib


def _bind_mapper(lib, prefix):
    d2s = getattr(lib, prefix + "_d2s")
    d2s.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64]
    d2s.restype = C.c_uint64
    s2d = getattr(lib, prefix + "_s2d")
    s2d.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    s2d.restype = None
    extra = getattr(lib, prefix + "_extra")
    extra.argtypes = [C.c_void_p, C.c_uint64]
    extra.restype = C.c_uint64
    return d2s, s2d, extra


@pytest.mark.parametrize("which", ["host position_mapper.hpp", "hip library Mapper"])
def test_position_mapper_equals_reference_table(which):
    if which.startswith("host"):
        d2s, s2d, extra = _bind_mapper(pagctl.test_lib(), "pagt_mapper")
    else:
        d2s, s2d, extra = _bind_mapper(_hip_host_only(), "pag_debug_mapper")
    n_rows = 0
    for line in open(os.path.join(goldens.GOLDEN, "func_mapper.txt")):
        p = line.split()
        if p[0] == "extra":
            assert extra(LENS.ctypes.data, len(LENS)) == int(p[1])
        elif p[0] == "d2s":
            assert d2s(LENS.ctypes.data, len(LENS), int(p[1]), int(p[2])) == int(p[3]), line
        elif p[0] == "s2d":
            i, q = C.c_int64(), C.c_int64()
            s2d(LENS.ctypes.data, len(LENS), int(p[1]), C.byref(i), C.byref(q))
            assert (i.value, q.value) == (int(p[2]), int(p[3])), line
        n_rows += 1
    assert n_rows > 250


def test_edit_distance_equals_reference_table():
    lib = _hip_host_only()
    n = 0
    for line in open(os.path.join(goldens.GOLDEN, "func_edit.txt")):
        a, b, d = line.split()
        assert lib.pag_debug_edit_distance(a.encode(), b.encode()) == int(d), line
        n += 1
    assert n == 400
