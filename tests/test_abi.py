"""The C-ABI library loads on a CPU-only host and exports exactly what include/ep24.h declares."""
import ctypes
import os
import subprocess

import pytest

from ep24 import _lib


def test_header_declares_the_hot_path_entry_points():
    protos = _lib.parse_header()
    for name in ["ep24_conv_fwd_bf16", "ep24_conv_dgrad_bf16", "ep24_conv_wgrad_bf16", "ep24_bn_act_fwd",
                 "ep24_head_decode_fwd", "ep24_assign_candidates", "ep24_assign_cost", "ep24_dynamic_k",
                 "ep24_assign_resolve", "ep24_loss_terms", "ep24_loss_finalize", "ep24_loss_grad",
                 "ep24_circle_pairwise", "ep24_circle_matched_fwd", "ep24_circle_matched_bwd", "ep24_sgd_nesterov",
                 "ep24_sector_map", "ep24_sector_gather"]:
        assert name in protos, name
    # plain C types only: no torch / C++ types in any signature
    allowed = {"int", "int32_t", "int64_t", "float", "double"}
    for name, (ret, params) in protos.items():
        for tstr, _ in params:
            base = tstr.replace("const", "").replace("*", "").strip()
            assert base in allowed | {"void", "float", "double", "uint8_t", "uint32_t", "uint64_t", "int32_t", "int64_t", "char"}, (name, tstr)


def test_library_loads_and_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "libep24.so not built: run __graft_entry__.build()"
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for name in _lib.parse_header():
        assert hasattr(cdll, name), "declared in ep24.h but not exported: " + name
    L = _lib.lib()
    assert L.fn["ep24_abi_version"]() == 1
    assert L.last_error() == ""


def test_no_undeclared_ep24_exports():
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("ep24_")}
    declared = set(_lib.parse_header())
    assert exported == declared, exported ^ declared


def test_product_path_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ep24 import loss
    with pytest.raises(_lib.Ep24Error):
        loss.bboxes_iou(torch.zeros(2, 50), torch.zeros(3, 26))
    with pytest.raises(IndexError):
        loss.bboxes_iou(torch.zeros(2, 49), torch.zeros(3, 26))
