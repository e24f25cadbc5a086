"""CPU-only: the C-ABI library builds, loads, and exports every symbol include/octic_hip.h declares.
No compute call is made (there is no GPU here)."""
import ctypes
import os

from octic_vits_amd import _lib
from octic_vits_amd.build import build


def test_library_builds_and_exports_header_symbols():
    path = build()
    assert os.path.exists(path)
    declared = _lib.header_symbols()
    assert len(declared) >= 25
    L = ctypes.CDLL(path)
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, f"declared in octic_hip.h but not exported: {missing}"
    # the ctypes prototypes cover exactly the header
    assert sorted(_lib._PROTOS) == declared
    # ... and the library exports nothing else under the octic_ prefix (no undeclared developer switches)
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("octic_")})
    assert exported == declared, sorted(set(exported) ^ set(declared))


def test_route_override_is_the_only_switch_and_round_trips():
    L = _lib.lib()
    assert L.octic_route_override(-1, 1) == -1 and L.octic_route_override(99, 1) == -1
    for knob in range(9):
        assert _lib.route_override(knob, 3) == 0
        assert _lib.route_override(knob, 0) == 3


def test_loader_checks_abi_version_and_errors_render():
    L = _lib.lib()
    assert L.octic_abi_version() == _lib.ABI_VERSION == 20
    assert b"shape" in L.octic_strerror(-1)
    assert b"align" in L.octic_strerror(-2)


def test_argument_validation_without_gpu():
    """Rejected arguments return negative codes before any launch."""
    L = _lib.lib()
    v = _lib.OcticView()
    for i in range(5):
        v.ptr[i] = 4096 + 64 * i
        v.ld[i] = 64
    # c not a multiple of 8
    assert L.octic_gelu_d8_fwd(ctypes.byref(v), ctypes.byref(v), 4, 12, _lib.F32, None) == -1
    # misaligned pointer
    v.ptr[2] = 4096 + 4
    assert L.octic_gelu_d8_fwd(ctypes.byref(v), ctypes.byref(v), 4, 8, _lib.F32, None) == -2
    v.ptr[2] = 0
    assert L.octic_gelu_d8_fwd(ctypes.byref(v), ctypes.byref(v), 4, 8, _lib.F32, None) == -4


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from octic_vits_amd import ops
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.pview(torch.zeros(2, 3, 64), 8)


def test_every_exported_symbol_is_documented_for_integrators():
    """INTEGRATION.md names, for every entry point of include/octic_hip.h, the reference code it stands in for."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = [s for s in _lib.header_symbols() if s not in text]
    assert not missing, missing
