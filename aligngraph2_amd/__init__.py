"""aligngraph2_amd — MI355X-native PAGraph hot path of AlignGraph2.

The product is native: `aligngraph2_amd/bin/pagraph` (host C++, drop-in for the reference's pagraph
command line) on top of `aligngraph2_amd/libpagraph_hip.so` (hand-written HIP kernels for gfx950 behind
the C ABI of include/pagraph_hip.h).  This Python package only locates / builds / launches them.
There is no CPU fallback: without the HIP library (or without a gfx950 device) everything raises.
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(PKG, "libpagraph_hip.so")
PAGRAPH = os.path.join(PKG, "bin", "pagraph")


def build(targets=("product",)):
    """Compile the HIP library (hipcc --offload-arch=gfx950) and the pagraph executable in-tree."""
    subprocess.run(["make", "-C", ROOT, *targets], check=True)


def require_built():
    for f in (LIB, PAGRAPH):
        if not os.path.exists(f):
            raise RuntimeError(f"{f} is missing — run aligngraph2_amd.build(); there is no CPU fallback")


def run_pagraph(argv, **kw):
    """Run the drop-in executable with the reference's argv (AlignGraph2.py:414-427) minus argv[0]."""
    require_built()
    return subprocess.run([PAGRAPH, *argv], **kw)


_hip = {}


def load_hip():
    """libpagraph_hip.so through ctypes (the C ABI of include/pagraph_hip.h).  Raises if it has not been built: there is no CPU
    fallback.  One HIP runtime per process: torch bundles its own libamdhip64 (same SONAME as /opt/rocm's); if our library
    were loaded first it would pull in the system runtime and a later `import torch` would mix it with torch's HSA ("no
    ROCm-capable device"), so torch is loaded first and both share torch's."""
    if "lib" not in _hip:
        import ctypes as C
        if not os.path.exists(LIB):
            raise RuntimeError(f"{LIB} is missing: run __graft_entry__.build() (no CPU fallback exists)")
        import torch
        torch.cuda.is_available()
        lib = C.CDLL(LIB)
        lib.pag_last_error.restype = C.c_char_p
        lib.pag_device_available.restype = C.c_int
        # handles are pointers: every entry point that takes one gets its argument types (a bare Python int would be passed as
        # a 32-bit C int)
        vp, u64p = C.c_void_p, C.POINTER(C.c_uint64)
        lib.pag_create_from_bitmap.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_int)]
        lib.pag_create_from_bitmap.restype = vp
        lib.pag_process.argtypes = [vp, vp, vp]
        lib.pag_process.restype = C.c_int
        lib.pag_prepare.argtypes = [vp, vp, vp]
        lib.pag_prepare.restype = C.c_int
        lib.pag_reset.argtypes = [vp]
        lib.pag_destroy.argtypes = [vp]
        lib.pag_destroy.restype = None
        lib.pag_csr_sizes.argtypes = [vp, u64p, u64p, u64p]
        lib.pag_export_csr.argtypes = [vp, vp]
        lib.pag_export_csr.restype = C.c_int
        lib.pag_travel.argtypes = [vp, vp, vp, vp, C.c_uint64, vp, vp]
        lib.pag_travel.restype = C.c_int
        lib.pag_travel_prepare.argtypes = [vp, vp, vp, C.c_uint64, vp, vp]
        lib.pag_travel_prepare.restype = C.c_int
        lib.pag_travel_prepare_for.argtypes = [vp, vp, vp, vp, C.c_uint64, vp, vp]
        lib.pag_travel_prepare_for.restype = C.c_int
        lib.pag_travel_view_sizes.argtypes = [vp, u64p, u64p, u64p, u64p, C.POINTER(C.c_int), u64p]
        lib.pag_travel_view_sizes.restype = C.c_int
        lib.pag_travel_path_oriented.argtypes = [vp, C.c_uint64, C.c_int, u64p]
        lib.pag_travel_path_oriented.restype = vp
        _hip["lib"] = lib
    return _hip["lib"]


def pagraph_argv(binary, in_dir, out_dir, threads=1, epsilon=10, cov=2, min_len=50):
    """The argv AlignGraph2.py uses for pagraph (reference AlignGraph2.py:414-427), incl. the doubled -r."""
    return [binary, "-t", str(threads), "-r", "dummy", "-k", os.path.join(in_dir, "kmer.bin"),
            "-c", os.path.join(in_dir, "ctg.fasta"), "-R", os.path.join(in_dir, "ref.fasta"),
            "-p", in_dir, "-a", os.path.join(in_dir, "aln"), "-o", out_dir, "-r", str(min_len),
            "--epsilon", str(epsilon), "-v", str(cov)]
