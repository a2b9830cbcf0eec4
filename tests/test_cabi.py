"""CPU: the C-ABI library loads and exports every symbol include/b3d.h declares; size queries and
argument validation work without a GPU (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from batch3dmot_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "b3d.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(b3d_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = declared_symbols()
    assert "b3d_pose_forward" in syms and "b3d_graph_build" in syms
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/b3d.h but not exported"


def test_version_and_workspace_queries():
    lib = _lib.load()
    assert lib.b3d_version() >= 100
    assert lib.b3d_graph_workspace_bytes(3000, 30000) > 4 * 30000 * 4
    inf = lib.b3d_pose_workspace_bytes(3000, 30000, 6, 0)
    tr = lib.b3d_pose_workspace_bytes(3000, 30000, 6, _lib.B3D_FLAG_TRAINING)
    assert 0 < inf < tr
    assert lib.b3d_pose_workspace_bytes(3000, 30000, 0, 0) == 0      # unsupported depth


def test_null_arguments_are_rejected_not_crashed():
    lib = _lib.load()
    g = _lib.b3d_graph()
    st = lib.b3d_graph_build(None, 10, 10, None, 0, C.byref(g), None)
    assert st == -1 and b"null" in lib.b3d_last_error()
    w = _lib.b3d_pose_weights()
    st = lib.b3d_pose_forward(C.byref(w), C.byref(g), None, None, None, 6, 0, None, 0, None, None, None)
    assert st == -1


def test_mlp_operator_descriptor_validation():
    """b3d_mlp_* / b3d_xattn_node_affine_* (SURVEY.md 8b families): size queries and descriptor checks run without a GPU."""
    lib = _lib.load()
    for s in ("b3d_mlp_forward", "b3d_mlp_backward", "b3d_xattn_node_affine_forward", "b3d_xattn_node_affine_backward"):
        assert hasattr(lib, s), s
    d = _lib._mlp_desc([640, 512, 384, 256, 128, 64], 0b01111, False)
    inf, tr = lib.b3d_mlp_workspace_bytes(C.byref(d), 30000, 0), lib.b3d_mlp_workspace_bytes(C.byref(d), 30000, _lib.B3D_FLAG_TRAINING)
    assert 0 < inf < tr and tr >= 30000 * (512 + 384 + 256 + 128) * 4
    assert lib.b3d_mlp_backward_scratch_bytes(C.byref(d), 30000) >= 2 * 30000 * 640 * 4
    bad = _lib._mlp_desc([64, 2000], 0, False)                          # width out of range
    assert lib.b3d_mlp_workspace_bytes(C.byref(bad), 10, 0) == 0
    layers = (_lib.b3d_linear * 1)()
    assert lib.b3d_mlp_forward(C.byref(bad), layers, None, 10, 0, None, 0, None, None) == -1
    assert b"widths" in lib.b3d_last_error()
    both = _lib._mlp_desc([8, 4], 0b1, True)                            # ReLU and Sigmoid behind the last layer
    assert lib.b3d_mlp_forward(C.byref(both), layers, None, 10, 0, None, 0, None, None) == -1
    ok = _lib._mlp_desc([8, 4], 0, False)
    assert lib.b3d_mlp_forward(C.byref(ok), layers, None, 10, 0, None, 0, None, None) == -1 and b"null" in lib.b3d_last_error()
    assert lib.b3d_xattn_node_affine_workspace_bytes(3000, 96, _lib.B3D_FLAG_TRAINING) >= 3000 * 96 * 4
    att = _lib.b3d_mha()
    assert lib.b3d_xattn_node_affine_forward(C.byref(att), 96, None, 10, 0, None, 0, None, None) == -1


def test_product_path_has_no_cpu_fallback():
    import torch
    from batch3dmot_amd import synth
    from batch3dmot_amd.pose_gnn import PoseGNN
    data = synth.make_graph(50, None, k=3, graph_idx=1)
    with pytest.raises(ValueError, match="GPU"):
        PoseGNN()(data)          # CPU tensors: must raise, never compute on the host
    for f in ("pose_gnn.py", "_lib.py", "data.py", "synth.py"):
        txt = open(os.path.join(ROOT, "batch3dmot_amd", f)).read()
        assert "oracle" not in txt.replace("the oracle", ""), f


def test_ctypes_structs_have_the_layout_of_the_header(tmp_path):
    """include/b3d.h compiled by the host C compiler: sizeof of every struct the ctypes stub mirrors, and the offset of the last
    field of the ones that grew (b3d_clr_inputs.encoders_ready), must equal what batch3dmot_amd/_lib.py declares -- a field added
    on one side only would shift every pointer behind it without any symbol changing."""
    import ctypes as C
    import shutil
    import subprocess
    from batch3dmot_amd import _lib
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no host C compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = ["b3d_graph", "b3d_linear", "b3d_gat", "b3d_batchnorm", "b3d_mp_weights", "b3d_pose_weights", "b3d_pose_grads",
             "b3d_mha", "b3d_clr_weights", "b3d_clr_grads", "b3d_clr_inputs", "b3d_mlp_desc"]
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "b3d.h"\nint main(void) {\n'
                   + "".join(f'  printf("{n} %zu\\n", sizeof({n}));\n' for n in names)
                   + '  printf("b3d_clr_inputs.encoders_ready %zu\\n", offsetof(b3d_clr_inputs, encoders_ready));\n  return 0;\n}\n')
    exe = tmp_path / "sizes"
    subprocess.run([cc, "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for n in names:
        assert int(out[n]) == C.sizeof(getattr(_lib, n)), n
    assert int(out["b3d_clr_inputs.encoders_ready"]) == _lib.b3d_clr_inputs.encoders_ready.offset


def test_counted_vmcnt_waits_match_the_shipped_isa():
    """tools/audit_vmcnt.py on the built library: every rendezvous of the fragment-streamed edge kernels waits for at most
    as many younger vector-memory instructions as the code object really holds behind the last LDS-DMA piece, and those
    kernels use no scratch."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "audit_vmcnt.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("violations 0") >= 4 and r.stdout.strip().endswith("OK"), r.stdout


def test_star_import_resolves_every_public_name():
    """``from batch3dmot_amd import *`` (what a notebook / the reference's ``from batch_3dmot.models... import *`` style does)
    must resolve every name of ``__all__`` -- the lazy ``__getattr__`` once forgot ``CausalMessagePassing``."""
    import batch3dmot_amd
    ns = {}
    exec("from batch3dmot_amd import *", ns)
    for name in batch3dmot_amd.__all__:
        assert name in ns, name
    assert ns["CausalMessagePassing"] is batch3dmot_amd.pose_gnn.CausalMessagePassing
    with pytest.raises(AttributeError):
        batch3dmot_amd.no_such_name
