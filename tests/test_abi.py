"""CPU: the C-ABI library loads and exports every symbol include/pagraph_hip.h declares, and the
product fails loudly (no CPU fallback) when there is no GPU."""
import ctypes
import os
import re
import subprocess

import pytest

import pagctl

HEADER = os.path.join(pagctl.ROOT, "include", "pagraph_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pag_[a-z_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(pagctl.HIP_LIB):
        subprocess.run(["make", "-C", pagctl.ROOT, "product"], check=True, capture_output=True)
    return ctypes.CDLL(pagctl.HIP_LIB)


def test_header_declares_the_expected_surface():
    fns = declared_functions()
    for f in ("pag_create", "pag_destroy", "pag_reset", "pag_process", "pag_export_csr", "pag_csr_sizes",
              "pag_solid_count", "pag_last_error", "pag_device_available"):
        assert f in fns


def test_library_exports_every_declared_symbol(lib):
    for f in declared_functions():
        assert hasattr(lib, f), f"libpagraph_hip.so does not export {f}"


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib.pag_create.restype = ctypes.c_void_p
    lib.pag_create.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    lib.pag_last_error.restype = ctypes.c_char_p
    err = ctypes.c_int(0)
    codes = (ctypes.c_uint64 * 2)(8, 3)
    g = lib.pag_create(codes, 2, 8, 0, ctypes.byref(err))
    assert not g and err.value == -19, "pag_create must fail with PAG_ENODEV when no gfx950 device exists"
    # and the drop-in executable exits non-zero instead of computing anything on the CPU
    exe = os.path.join(pagctl.ROOT, "aligngraph2_amd", "bin", "pagraph")
    r = subprocess.run([exe, "-k", "/nonexistent", "-p", "/nonexistent", "-o", "/tmp"], capture_output=True, text=True)
    assert r.returncode != 0


def test_cli_contract_without_gpu():
    exe = os.path.join(pagctl.ROOT, "aligngraph2_amd", "bin", "pagraph")
    assert subprocess.run([exe], capture_output=True).returncode == 0          # no args: usage, exit 0
    assert subprocess.run([exe, "-h"], capture_output=True).returncode == 0     # help: exit 0
    assert subprocess.run([exe, "--bogus", "1"], capture_output=True).returncode == 1  # unknown flag: exit 1
    assert subprocess.run([exe, "-t", "abc"], capture_output=True).returncode == 1
