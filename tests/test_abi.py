"""CPU-side checks of the C-ABI boundary: the library builds, loads, and exports exactly the entry
points include/espnet_amd.h declares (no kernels are launched here)."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT

HDR = os.path.join(ROOT, "include", "espnet_amd.h")
LIB = os.path.join(ROOT, "espnet_amd", "csrc", "libespnet_amd_hip.so")


def declared_symbols():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(eamd_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.dirname(LIB), "-j4"])
    return ctypes.CDLL(LIB)


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert "eamd_gemm" in syms and "eamd_ctc_loss" in syms and len(syms) >= 35


def test_library_exports_every_declared_symbol(lib):
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_python_binding_lists_every_symbol():
    from espnet_amd import _lib
    assert sorted(_lib.SYMBOLS) == declared_symbols()


def test_abi_version(lib):
    assert lib.eamd_abi_version() == 1


def test_bad_arguments_are_rejected_without_a_gpu(lib):
    # NULL operands must be refused on the host (negative code), never launched
    assert lib.eamd_gemm(None, None) < 0
    lib.eamd_ctc_workspace_bytes.restype = ctypes.c_int64
    assert lib.eamd_ctc_workspace_bytes(2, 10, 3) > 0
    assert lib.eamd_layernorm_fwd(None, None, None, None, None, None, 4, 8, ctypes.c_float(1e-12), None) < 0


def test_struct_layout_matches_header():
    from espnet_amd._lib import GatherT, GemmT, RowMapT
    assert ctypes.sizeof(GatherT) == 4 * 27
    assert ctypes.sizeof(RowMapT) == 4 * 9
    # 6 pointers, 5 int32, 5 int64, 2 int32, 6 int64, 2 float, 6 int32, gather, rowmap (8-byte aligned)
    assert ctypes.sizeof(GemmT) % 8 == 0 and ctypes.sizeof(GemmT) >= 48 + 20 + 40 + 8 + 48 + 8 + 24 + 108 + 36


def test_round3_struct_layouts_and_null_arguments():
    """eamd_ffn_t / eamd_lstm_seq_*_t as ctypes sees them = as the header lays them out; the new entry points reject NULL
    arguments before touching a device"""
    from espnet_amd import _lib
    # 9 pointers, 4 int32, 2 float, u64, float (+pad), u64, pointer, 2 int32; LayerNorm in front: 5 pointers, float, int32;
    # round 4, LayerNorm backward behind eamd_ffn_bwd: 7 pointers, u64, float, int32
    assert ctypes.sizeof(_lib.FfnT) == 136 + 48 + 72
    assert ctypes.sizeof(_lib.RowProjT) == 240 and ctypes.sizeof(_lib.RowProjPackT) == 32       # include/espnet_amd.h: eamd_rowproj_t, eamd_rowproj_pack_t
    assert ctypes.sizeof(_lib.LstmSeqFwdT) == 8 * 8 + 8    # 8 pointers, 2 int32
    assert ctypes.sizeof(_lib.LstmSeqBwdT) == 6 * 8 + 8    # 6 pointers, 2 int32
    lib = _lib.lib()
    assert lib.eamd_lstm_seq_sync_bytes() == 4096
    assert lib.eamd_lstm_seq_fwd(None, 1, 4, 2, 64, None, None) < 0
    assert lib.eamd_lstm_seq_bwd(None, 2, 4, 2, 64, None, None) < 0
    assert lib.eamd_ffn_fwd(None, None) < 0 and lib.eamd_ffn_bwd(None, None) < 0
    assert lib.eamd_ctc_prefix_score_batch(None, None, 1, 1, None, None, None, None, None, None, 1, 1, 1, 0, 1, None) < 0
    # eamd_gemm_multi validates every descriptor before it launches anything: NULL table, n < 1, and a descriptor without operands
    assert lib.eamd_gemm_multi(None, 2, None) < 0
    arr = (_lib.GemmT * 2)()
    assert lib.eamd_gemm_multi(arr, 0, None) < 0 and lib.eamd_gemm_multi(arr, 2, None) < 0


def test_zero_arena_host_logic():
    """ops.zeros: torch.zeros until a step has been measured; then distinct 256-byte aligned slices of one fresh buffer per
    begin (slices of an earlier step keep their own storage), torch.zeros again beyond the buffer and after zero_arena_off"""
    import torch
    from espnet_amd import ops
    a = ops._zarena
    ops.zero_arena_off()
    a.cap = 0
    ops.zeros(1 << 16, device="cpu")                        # demand outside a training step (an eval forward, a search): not counted
    assert a.need == 0
    ops.zero_arena_begin("cpu")
    assert not a.active
    x, y = ops.zeros(3, 5, device="cpu"), ops.zeros((7,), device="cpu")
    assert x.shape == (3, 5) and y.shape == (7,) and a.need == 512
    ops.zero_arena_begin("cpu")
    assert a.active and a.cap >= 512
    p, q = ops.zeros(3, 5, device="cpu"), ops.zeros((7,), device="cpu")
    assert p.data_ptr() == a.buf.data_ptr() and q.data_ptr() - p.data_ptr() == 256
    p += 1.0
    big = ops.zeros(1 << 20, device="cpu")                  # beyond the buffer: an ordinary allocation
    assert big.data_ptr() < a.buf.data_ptr() or big.data_ptr() >= a.buf.data_ptr() + a.cap
    ops.zero_arena_begin("cpu")
    r = ops.zeros(3, 5, device="cpu")
    assert float(r.sum()) == 0.0 and float(p.sum()) == 15.0 and r.data_ptr() != p.data_ptr()   # the old slice lives on
    ops.zero_arena_off()
    z = ops.zeros(4, device="cpu")
    assert a.buf is None and float(z.sum()) == 0.0 and a.need == 0
    # the arena follows the last step's demand down as well as up
    ops.zero_arena_begin("cpu")
    ops.zeros(1 << 18, device="cpu")
    ops.zero_arena_begin("cpu")
    assert a.cap >= (1 << 20)
    ops.zeros(16, device="cpu")
    ops.zero_arena_begin("cpu")
    assert a.cap == 4096
    ops.zero_arena_off()
    a.cap = 0


def test_product_path_has_no_cpu_fallback():
    import torch
    from espnet_amd import ops, _lib
    x = torch.zeros(4, 8)
    with pytest.raises(_lib.EamdError):
        ops.layernorm_fwd(x, torch.ones(8), torch.zeros(8), 1e-12)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "espnet_amd")):
        for f in files:
            if f.endswith(".py"):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().replace("# oracle", ""), f
