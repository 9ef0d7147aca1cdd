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
