"""ctypes bindings used by the tests / bench / smoke: the HIP library (product), the C oracle (checker)
and the test-support loader that builds the flat pag_build_input from a pagraph input directory.

Nothing here is product code.  The oracle is only ever used as the checker.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_LIB = os.path.join(ROOT, "aligngraph2_amd", "libpagraph_hip.so")
ORACLE_LIB = os.path.join(ROOT, "oracle", "libpag_oracle.so")
TEST_LIB = os.path.join(ROOT, "tests", "harness", "bin", "libpagh_test.so")
WALK_TEST_LIB = os.path.join(ROOT, "tests", "harness", "bin", "libpagh_walk_test.so")
REF_DIR = os.path.join(ROOT, "oracle", "_ref")


import sys  # noqa: E402

sys.path.insert(0, ROOT)
from aligngraph2_amd.parallel import BuildStats  # noqa: E402,F401  (pag_build_stats, include/pagraph_hip.h)


class Csr(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("n_pos", C.c_uint64), ("n_edges", C.c_uint64),
                ("node_code", C.c_void_p), ("pos_off", C.c_void_p), ("pos_ctg", C.c_void_p), ("pos_ref", C.c_void_p),
                ("pos_cnt", C.c_void_p), ("edge_off", C.c_void_p), ("edge_to", C.c_void_p), ("edge_step", C.c_void_p)]


def ensure_built(targets=("harness", "oracle")):
    """(Re)build test-side binaries if missing.  The product library is built by __graft_entry__.build()."""
    need = [t for t, f in (("harness", TEST_LIB), ("oracle", ORACLE_LIB)) if t in targets and not os.path.exists(f)]
    for t in need:
        subprocess.run(["make", "-C", ROOT, t], check=True, capture_output=True)


def _bind(lib, prefix):
    vp, u64p = C.c_void_p, C.POINTER(C.c_uint64)
    getattr(lib, prefix + "_process").argtypes = [vp, vp, C.POINTER(BuildStats)]
    getattr(lib, prefix + "_process").restype = C.c_int
    getattr(lib, prefix + "_csr_sizes").argtypes = [vp, u64p, u64p, u64p]
    getattr(lib, prefix + "_export_csr").argtypes = [vp, C.POINTER(Csr)]
    getattr(lib, prefix + "_debug_stream_sizes").argtypes = [vp, u64p, u64p]
    getattr(lib, prefix + "_debug_streams").argtypes = [vp, vp, vp, vp, vp]
    getattr(lib, prefix + "_reset").argtypes = [vp]
    getattr(lib, prefix + "_destroy").argtypes = [vp]
    getattr(lib, prefix + "_destroy").restype = None
    getattr(lib, prefix + "_solid_count").argtypes = [vp]
    getattr(lib, prefix + "_solid_count").restype = C.c_uint64


_libs = {}


def hip_lib():
    if "hip" not in _libs:
        import sys
        sys.path.insert(0, ROOT)
        import aligngraph2_amd
        lib = aligngraph2_amd.load_hip()  # (raises if the library is missing: no CPU fallback exists)
        _bind(lib, "pag")
        lib.pag_create.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(C.c_int)]
        lib.pag_create.restype = C.c_void_p
        _libs["hip"] = lib
    return _libs["hip"]


def oracle_lib():
    if "oracle" not in _libs:
        ensure_built(("oracle",))
        lib = C.CDLL(ORACLE_LIB)
        _bind(lib, "pago")
        lib.pago_create.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
        lib.pago_create.restype = C.c_void_p
        lib.pago_debug_enable.argtypes = [C.c_void_p, C.c_int]
        lib.pago_check_position.argtypes = [C.c_uint32] * 6 + [C.c_double]
        lib.pago_edge_similar.argtypes = [C.c_uint32] * 4 + [C.c_int, C.c_uint64, C.c_double]
        lib.pago_cluster_similar.argtypes = [C.c_uint32] * 4 + [C.c_uint64]
        lib.pago_kmer_codes.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_void_p]
        lib.pago_kmer_codes.restype = C.c_uint64
        _libs["oracle"] = lib
    return _libs["oracle"]


def test_lib():
    if "test" not in _libs:
        ensure_built(("harness",))
        lib = C.CDLL(TEST_LIB)
        lib.pagh_load.argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.c_uint64, C.c_uint64]
        lib.pagh_load.restype = C.c_void_p
        lib.pagh_view.argtypes = [C.c_void_p]
        lib.pagh_view.restype = C.c_void_p
        lib.pagh_raw_view.argtypes = [C.c_void_p]
        lib.pagh_raw_view.restype = C.c_void_p
        lib.pagh_kmer_words.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        lib.pagh_kmer_words.restype = C.c_void_p
        lib.pagh_total_read_bases.argtypes = [C.c_void_p]
        lib.pagh_total_read_bases.restype = C.c_uint64
        lib.pagh_free.argtypes = [C.c_void_p]
        lib.pagh_free.restype = None
        _libs["test"] = lib
    return _libs["test"]


def walk_test_lib():
    """pagt_traverse_hostwalk: device graph exported, walk by the host restatement of the reference's PAlgorithm (test
    infrastructure, tests/harness/host_walk.cpp); same argument list as pagh_traverse."""
    if "walk" not in _libs:
        hip_lib()  # (the HIP runtime has to come from torch, see hip_lib)
        if not os.path.exists(WALK_TEST_LIB):
            subprocess.run(["make", "-C", ROOT, WALK_TEST_LIB[len(ROOT) + 1:]], check=True, capture_output=True)
        _libs["walk"] = C.CDLL(WALK_TEST_LIB)
    return _libs["walk"]


class LoadedInput:
    """A pagraph input directory parsed by the product's host pipeline into the flat C-ABI input."""

    def __init__(self, in_dir: str, threads: int = 1, eps: int = 10, cov: int = 2, block: int = 0):
        self.lib = test_lib()
        self.h = self.lib.pagh_load(in_dir.encode(), block, threads, eps, cov)
        if not self.h:
            raise RuntimeError(f"pagh_load failed for {in_dir}")
        self.view = self.lib.pagh_view(self.h)          # the HOST restatement of the preparation stage (checker)
        self.raw_view = self.lib.pagh_raw_view(self.h)  # what the product hands to pag_prepare
        n, k = C.c_uint64(), C.c_uint64()
        self.kmer_words = self.lib.pagh_kmer_words(self.h, C.byref(n), C.byref(k))
        self.n_kmer_words, self.k = n.value, k.value
        self.n_bases = self.lib.pagh_total_read_bases(self.h)

    def close(self):
        if self.h:
            self.lib.pagh_free(self.h)
            self.h = None


def _prepared_view(lib, g, inp: LoadedInput):
    """pag_prepare: the product's device-side preparation of the block -> a device-resident pag_build_input"""
    import biggen
    out = biggen.PagBuildInput()
    lib.pag_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    rc = lib.pag_prepare(g, inp.raw_view, C.byref(out))
    if rc != 0:
        raise RuntimeError(f"pag_prepare failed rc={rc} {lib.pag_last_error().decode()}")
    return out


def device_bytes(ptr, n):
    """n bytes of device memory as a numpy array (through torch's HIP runtime)"""
    import torch
    rt = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    buf = np.empty(n, dtype=np.uint8)
    if n:
        rc = rt.hipMemcpy(buf.ctypes.data, ptr, n, 2)
        if rc != 0:
            raise RuntimeError(f"hipMemcpy failed ({rc})")
    return buf


def _run(lib, prefix, g, inp: LoadedInput, streams: bool, prepared=None):
    st = BuildStats()
    rc = getattr(lib, prefix + "_process")(g, C.byref(prepared) if prepared is not None else inp.view, C.byref(st))
    if rc != 0:
        msg = lib.pag_last_error().decode() if prefix == "pag" else ""
        raise RuntimeError(f"{prefix}_process failed rc={rc} {msg}")
    nn, np_, ne = C.c_uint64(), C.c_uint64(), C.c_uint64()
    getattr(lib, prefix + "_csr_sizes")(g, C.byref(nn), C.byref(np_), C.byref(ne))
    out = {
        "node_code": np.zeros(nn.value, np.uint32), "pos_off": np.zeros(nn.value + 1, np.uint64),
        "pos_ctg": np.zeros(np_.value, np.uint32), "pos_ref": np.zeros(np_.value, np.uint32),
        "pos_cnt": np.zeros(np_.value, np.uint16), "edge_off": np.zeros(nn.value + 1, np.uint64),
        "edge_to": np.zeros(ne.value, np.uint32), "edge_step": np.zeros(ne.value, np.int32),
    }
    csr = Csr(nn.value, np_.value, ne.value, *[out[k].ctypes.data for k in
                                                ("node_code", "pos_off", "pos_ctg", "pos_ref", "pos_cnt", "edge_off",
                                                 "edge_to", "edge_step")])
    rc = getattr(lib, prefix + "_export_csr")(g, C.byref(csr))
    if rc != 0:
        msg = lib.pag_last_error().decode() if prefix == "pag" else ""
        raise RuntimeError(f"{prefix}_export_csr failed rc={rc} {msg}")
    res = {"stats": st, "csr": out}
    if streams:
        nt, ne2 = C.c_uint64(), C.c_uint64()
        getattr(lib, prefix + "_debug_stream_sizes")(g, C.byref(nt), C.byref(ne2))
        s = {"tkey": np.zeros(nt.value, np.uint32), "tval": np.zeros(nt.value, np.uint64),
             "ekey": np.zeros(ne2.value, np.uint32), "eval": np.zeros(ne2.value, np.uint64)}
        getattr(lib, prefix + "_debug_streams")(g, *[s[k].ctypes.data for k in ("tkey", "tval", "ekey", "eval")])
        res["streams"] = s
    return res


def run_oracle(inp: LoadedInput, streams: bool = False):
    lib = oracle_lib()
    g = lib.pago_create(inp.kmer_words, inp.n_kmer_words, inp.k)
    try:
        lib.pago_debug_enable(g, 1 if streams else 0)
        return _run(lib, "pago", g, inp, streams)
    finally:
        lib.pago_destroy(g)


def run_hip(inp: LoadedInput, streams: bool = False, device: int = 0, prepare: bool = True):
    """The product path: pag_prepare (device) -> pag_process.  prepare=False feeds pag_process the host restatement's
    arrays instead (the path the C ABI also accepts: host-resident pag_build_input)."""
    lib = hip_lib()
    if streams:
        os.environ["PAG_DEBUG_KEEP_STREAMS"] = "1"
    else:
        os.environ.pop("PAG_DEBUG_KEEP_STREAMS", None)
    err = C.c_int()
    g = lib.pag_create(inp.kmer_words, inp.n_kmer_words, inp.k, device, C.byref(err))
    if not g:
        raise RuntimeError(f"pag_create failed rc={err.value}: {lib.pag_last_error().decode()}")
    try:
        return _run(lib, "pag", g, inp, streams, _prepared_view(lib, g, inp) if prepare else None)
    finally:
        lib.pag_destroy(g)


def first_diff(a: np.ndarray, b: np.ndarray):
    n = min(len(a), len(b))
    d = np.flatnonzero(a[:n] != b[:n])
    if len(d):
        return int(d[0])
    return None if len(a) == len(b) else n


def compare_results(hip, ora, label=""):
    """Raise AssertionError with a precise message at the first differing stage."""
    msgs = []
    if "streams" in hip and "streams" in ora:
        for k in ("tkey", "tval", "ekey", "eval"):
            a, b = hip["streams"][k], ora["streams"][k]
            i = first_diff(a, b)
            if i is not None:
                lo = max(0, i - 2)
                msgs.append(f"{label} stream {k}: len hip={len(a)} oracle={len(b)} first diff at {i}: "
                            f"hip={a[lo:i + 3].tolist()} oracle={b[lo:i + 3].tolist()}")
    hs, os_ = hip["stats"], ora["stats"]
    if hs.counts() != os_.counts():
        msgs.append(f"{label} count lines differ: hip={hs.counts()} oracle={os_.counts()}")
    for k in ("n_tuples", "n_edges"):
        if tuple(getattr(hs, k)) != tuple(getattr(os_, k)):
            msgs.append(f"{label} {k}: hip={tuple(getattr(hs, k))} oracle={tuple(getattr(os_, k))}")
    for k, a in hip["csr"].items():
        b = ora["csr"][k]
        i = first_diff(a, b)
        if i is not None:
            lo = max(0, i - 2)
            msgs.append(f"{label} csr {k}: len hip={len(a)} oracle={len(b)} first diff at {i}: "
                        f"hip={a[lo:i + 3].tolist()} oracle={b[lo:i + 3].tolist()}")
    if msgs:
        raise AssertionError("\n".join(msgs))


def run_reference(in_dir: str, out_dir: str, threads: int = 1, eps: int = 10, cov: int = 2, tool: str = "pagraph"):
    """Run the compiled reference (oracle/_ref) on an input directory; serialised threads for -t > 1."""
    import synth
    env = dict(os.environ)
    if threads > 1:
        env["LD_PRELOAD"] = os.path.join(REF_DIR, "libserial_threads.so")
    os.makedirs(out_dir, exist_ok=True)
    argv = synth.pagraph_argv(os.path.join(REF_DIR, tool), in_dir, out_dir, threads=threads, epsilon=eps, cov=cov)
    return subprocess.run(argv, env=env, capture_output=True, text=True)
