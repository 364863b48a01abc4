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
    assert L.fn["ep24_abi_version"]() == 3            # EP24_ABI_VERSION of include/ep24.h (round 5: the update writes the packed forward copy)
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


def _device_code_objects(lib_path, tmp):
    """The gfx950 code objects (ELF, e_machine AMDGPU) bundled in the library's .hip_fatbin section, one per translation unit."""
    import struct
    objcopy = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([objcopy, "--dump-section", ".hip_fatbin=" + fat, lib_path, os.path.join(tmp, "unused.so")], check=True)
    blob = open(fat, "rb").read()
    out, pos = [], 0
    while True:
        pos = blob.find(b"\x7fELF", pos)
        if pos < 0:
            break
        hdr = blob[pos:pos + 64]
        if hdr[4] == 2 and struct.unpack_from("<H", hdr, 18)[0] == 224:          # ELF64, EM_AMDGPU
            shoff, = struct.unpack_from("<Q", hdr, 40)
            shentsize, shnum = struct.unpack_from("<HH", hdr, 58)
            size = shoff + shentsize * shnum
            path = os.path.join(tmp, "co%d.elf" % len(out))
            open(path, "wb").write(blob[pos:pos + size])
            out.append(path)
            pos += size
        else:
            pos += 4
    return out


def test_no_half_swapped_packed_fp32(tmp_path):
    """No kernel of the library contains a packed fp32 VALU instruction whose LOW result reads the HIGH half of a source
    (v_pk_mul/add/fma_f32 with a 1 in op_sel).  On gfx950 that form returned wrong values in lanes 48..63 whenever an MFMA wave
    of another kernel shared the SIMD (tools/hazard_probe.hip; profiles/r03_packed_fp32_hazard.txt) - the cause of round 2's
    flipped SimOTA candidate bits.  The library is built with the vectorisers off (csrc/Makefile NOVEC); this test reads the
    shipped binary, so a hand-written float2 expression or a changed flag cannot bring the instruction back unnoticed."""
    import re
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(objdump) and os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objcopy")):
        pytest.skip("llvm-objdump / llvm-objcopy not installed")
    cos = _device_code_objects(_lib.LIB_PATH, str(tmp_path))
    assert len(cos) >= 10, "expected one gfx950 code object per translation unit, found %d" % len(cos)
    bad, n_kernels, n_instr = [], 0, 0
    pat = re.compile(r"\bv_pk_(mul|add|fma)_f32\b.*\bop_sel:\[([01,]+)\]")
    for co in cos:
        txt = subprocess.run([objdump, "-d", "--mcpu=gfx950", co], capture_output=True, text=True, check=True).stdout
        kernel = None
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                kernel = m.group(1)
                n_kernels += 1
                continue
            n_instr += 1
            m = pat.search(line)
            if m and "1" in m.group(2):
                bad.append((kernel, line.strip().split("//")[0].strip()))
    assert n_kernels > 50 and n_instr > 100000, (n_kernels, n_instr)          # the disassembly really covered the library
    assert not bad, "packed fp32 with a half-swapping op_sel in: %s" % bad[:5]


def test_graft_entry_build_runs():
    """The driver's "does it build" check: make (a no-op on a built tree), import of the package and of the oracle, and the ABI
    version of the built library against the header's (a stale constant in build() went unnoticed through the version bump of round 4)."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    ge = importlib.import_module("__graft_entry__")
    ge.build()
