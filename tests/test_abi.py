"""The C-ABI library loads on a CPU-only host and exports every symbol that
include/iago_hip.h declares.  No compute calls (no GPU here)."""
import os
import re
import subprocess

import pytest

from iago_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def so():
    return build.build()


def header_symbols(name="iago_hip.h"):
    text = open(os.path.join(ROOT, "include", name)).read()
    return sorted(set(re.findall(r"IAGO_API[^;(]*?\b(iago_\w+)\s*\(", text)))


def test_header_matches_symbol_list(so):
    assert header_symbols() == sorted(_lib.SYMBOLS)
    # the layer-level entry points have a header of their own; the schedules that measured slower, the per-phase forms
    # the one-launch descent superseded are fenced off in a third (VERDICT r04 task 8, r05 task 7)
    assert header_symbols("iago_hip_layers.h") == sorted(_lib.LAYER_SYMBOLS)
    assert header_symbols("iago_hip_experimental.h") == sorted(_lib.EXPERIMENTAL_SYMBOLS)
    assert len(set(_lib.SYMBOLS) | set(_lib.LAYER_SYMBOLS) | set(_lib.EXPERIMENTAL_SYMBOLS)) == \
        len(_lib.SYMBOLS) + len(_lib.LAYER_SYMBOLS) + len(_lib.EXPERIMENTAL_SYMBOLS)
    # the boundary stays a boundary (VERDICT r05 task 7): 33 entry points, a header under 700 lines
    assert len(_lib.SYMBOLS) <= 33
    assert len(open(os.path.join(ROOT, "include", "iago_hip.h")).read().splitlines()) < 700


def test_library_exports_every_symbol(so):
    out = subprocess.check_output(["nm", "-D", "--defined-only", so]).decode()
    exported = set(re.findall(r" T (iago_\w+)", out))
    every = header_symbols() + header_symbols("iago_hip_layers.h") + header_symbols("iago_hip_experimental.h")
    assert set(every) <= exported
    L = _lib.lib()
    for name in every:
        assert hasattr(L, name), name
    assert L.iago_abi_version() == _lib.ABI_VERSION == 13


def test_gfx950_code_object(so):
    data = open(so, "rb").read()
    assert b"gfx950" in data


def test_no_cpu_fallback_in_product():
    """Nothing under iago_amd/ may import the oracle."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "iago_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "liboracle" not in text, f


def test_ops_refuse_cpu_tensors(so):
    import torch
    from iago_amd import ops
    t = torch.zeros(4, dtype=torch.int64)
    with pytest.raises(_lib.IagoError):
        ops.legal_moves(t, t)


def test_invalid_arguments_are_reported(so):
    L = _lib.lib()
    assert L.iago_legal_moves(None, None, None, 4, None) == -1
    assert b"iago_legal_moves" in L.iago_last_error()
    assert L.iago_legal_moves(None, None, None, 0, None) == 0
    assert L.iago_rollout(None, None) == -1


def test_rollout_table_builder_host_side(so):
    """iago_rollout_build_table runs on the host: product form keeps every factor in
    (0, 1] whatever the common offset of the logits, the uniform policy is all ones,
    and a wide logit range selects the log form."""
    import ctypes as C
    import numpy as np
    L = _lib.lib()
    n = _lib.IAGO_ROLLOUT_TABLE_FLOATS
    rs = np.random.RandomState(0)

    def build(w, b):
        blob = np.empty(n, np.float32)
        wp = C.c_void_p(w.ctypes.data) if w is not None else None
        bp = C.c_void_p(b.ctypes.data) if b is not None else None
        assert L.iago_rollout_build_table(wp, bp, C.c_void_p(blob.ctypes.data)) == 0
        return blob

    w = rs.randn(18).astype(np.float32)
    for offset in (0.0, 500.0, -3000.0):
        b = (rs.randn(64) + offset).astype(np.float32)
        blob = build(w, b)
        mode = _lib.ROLLOUT_MODE_INDEX
        assert blob[mode] == 1.0                     # product form
        body = np.concatenate([blob[:mode], blob[mode + 4:]])
        assert np.all(np.isfinite(body)) and body.min() > 0.0 and body.max() == 1.0
        # bias factors: exp(b - max b)
        assert np.allclose(blob[mode - 64:mode], np.exp(b.astype(np.float64) - b.max()), rtol=1e-6)
    uni = build(None, None)
    mode = _lib.ROLLOUT_MODE_INDEX
    assert np.all(uni[:mode] == 1.0) and uni[mode] == 1.0 and np.all(uni[mode + 4:] == 1.0)
    wide = build((12 * rs.randn(18)).astype(np.float32), (5 * rs.randn(64)).astype(np.float32))
    assert wide[mode] == 0.0                         # log form: raw sums
    assert L.iago_rollout_build_table(C.c_void_p(w.ctypes.data), None, None) == -1


def test_workspace_bytes_is_a_64_bit_result(so):
    """iago_policy_grad_workspace_bytes returns bytes: beyond 2^31 from ~7,000 rows on (the bindings once read it as an
    int and a 8,192-row chunk got a workspace 'too small')."""
    L = _lib.lib()
    assert L.iago_policy_grad_workspace_bytes(10000) > 2 ** 31
    assert L.iago_policy_grad_workspace_bytes(10000) > L.iago_policy_grad_workspace_bytes(4096) > 304 * 1024 * 4096
