"""The C-ABI library builds, loads without a GPU and exports every symbol include/*.h declares (mistral_hip.h: the product
boundary; mistral_hip_debug.h: engine diagnostics)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header=None):
    names = set()
    for h in ([header] if header else sorted(os.listdir(os.path.join(ROOT, "include")))):
        if not h.endswith(".h"):
            continue
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_debug_entry_points_live_in_their_own_header():
    """Diagnostics (trace, knobs, sabotage) are not part of the boundary a maintainer binds."""
    assert not [n for n in _declared("mistral_hip.h") if n.startswith("mi_debug_")]
    assert all(n.startswith("mi_debug_") for n in _declared("mistral_hip_debug.h"))


def test_header_symbols_exported():
    from mistral_inference import _hip
    handle = ctypes.CDLL(_hip.LIB_PATH)
    names = _declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(handle, n), f"{n} declared in mistral_hip.h but not exported"
    assert set(names) == set(_hip.EXPORTED_SYMBOLS), set(names) ^ set(_hip.EXPORTED_SYMBOLS)


def test_abi_version_and_error_strings():
    from mistral_inference import _hip
    L = _hip.lib()
    assert L.mi_abi_version() == _hip.MI_ABI_VERSION
    assert L.mi_error_string(0) == b"ok"
    assert b"shape" in L.mi_error_string(-2)
    # argument checks run before any device work, so they are testable on a box without a GPU
    assert L.mi_rmsnorm(None, None, None, 1, 8, 1e-5, None) == -1
    assert b"mi_rmsnorm" in L.mi_last_error_detail()
    assert L.mi_workspace_bytes(None, 1, 1, 1) == 0


def test_no_oracle_import_in_product():
    """The shipped package must never reach into oracle/ (it has no CPU path to fall back to)."""
    pkg = os.path.join(ROOT, "mistral-inference_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cuh")):
                text = open(os.path.join(dirpath, f)).read()
                assert "mistral_oracle" not in text and "import oracle" not in text, os.path.join(dirpath, f)


def test_unsupported_shapes_fail_loudly_before_any_device_work():
    """mi_forward validates the model description first: wrong head_dim / head counts / MoE width come back as MI_ERR_SHAPE
    with a message, never as a silent wrong answer (checked without a GPU: validation precedes every launch)."""
    import ctypes as C
    from mistral_inference import _hip
    L = _hip.lib()
    layers = (_hip.MiLayer * 1)()
    m = _hip.MiModel()
    m.dim, m.n_heads, m.n_kv_heads, m.head_dim, m.hidden_dim, m.vocab_size, m.n_layers = 256, 4, 2, 64, 512, 100, 1
    m.layers = C.cast(layers, C.POINTER(_hip.MiLayer))
    bt = _hip.MiBatch()
    assert L.mi_forward(C.byref(m), C.byref(bt), None) == -2 and b"head_dim" in L.mi_last_error_detail()
    m.head_dim, m.n_heads, m.n_kv_heads = 128, 7, 2   # 7 query heads over 2 kv heads: not a GQA layout
    assert L.mi_forward(C.byref(m), C.byref(bt), None) == -2 and b"n_heads" in L.mi_last_error_detail()
    m.n_heads = 6                                     # ratio 3 is fine (query heads are grouped 3 x 1)
    assert L.mi_forward(C.byref(m), C.byref(bt), None) == -1
    m.n_heads, m.num_experts, m.top_k = 4, 32, 2
    assert L.mi_forward(C.byref(m), C.byref(bt), None) == -2 and b"MoE" in L.mi_last_error_detail()
    m.num_experts = m.top_k = 0
    assert L.mi_forward(C.byref(m), C.byref(bt), None) == -1     # valid model, empty batch -> MI_ERR_ARG
    assert L.mi_attn_decode(1, 1, 128, 1, 1, 16, 1, 4, 2, 64, 1, 1, 0, None) == -2   # head_dim 64
    assert L.mi_attn_decode(1, 1, 128, 1, 1, 16, 1, 4, 2, 128, 1, 1, 2, None) == -1  # ABI v7: an unknown K/V ring layout code
    bt.kv_layout = 7
    assert L.mi_forward(C.byref(m), C.byref(bt), None) == -1
    bt.kv_layout = 0


def test_generic_entry_takes_what_mi_forward_declines():
    """mi_forward_generic (ABI v6): the shapes above that are MI_ERR_SHAPE for the tuned kernels pass ITS validation (and
    stop at the empty batch, MI_ERR_ARG); an unknown dtype code, an odd head_dim and top_k > num_experts are refused before any
    device work.  The host-side predicate that routes a model (`tuned_kernels_take`) agrees with the library on every case."""
    import ctypes as C
    from mistral_inference import _hip
    from mistral_inference.args import MoeArgs, TransformerArgs
    from mistral_inference.transformer import tuned_kernels_take
    L = _hip.lib()
    layers = (_hip.MiLayer * 1)()
    bt = _hip.MiBatch()

    def both(dim=256, n_heads=4, n_kv_heads=2, head_dim=128, hidden_dim=512, E=0, k=0):
        m = _hip.MiModel()
        m.dim, m.n_heads, m.n_kv_heads, m.head_dim, m.hidden_dim, m.vocab_size, m.n_layers = dim, n_heads, n_kv_heads, head_dim, hidden_dim, 100, 1
        m.num_experts, m.top_k = E, k
        m.layers = C.cast(layers, C.POINTER(_hip.MiLayer))
        a = TransformerArgs(dim=dim, n_layers=1, head_dim=head_dim, hidden_dim=hidden_dim, n_heads=n_heads, n_kv_heads=n_kv_heads,
                            norm_eps=1e-5, vocab_size=100, moe=MoeArgs(num_experts=E, num_experts_per_tok=k) if E else None)
        tuned = L.mi_forward(C.byref(m), C.byref(bt), None)
        assert (tuned != _hip.MI_ERR_SHAPE) == tuned_kernels_take(a), (dim, n_heads, n_kv_heads, head_dim, hidden_dim, E, k)
        return tuned, [L.mi_forward_generic(C.byref(m), C.byref(bt), d, None) for d in (0, 1, 2)], m

    assert both()[:2] == (-1, [-1, -1, -1])
    assert both(head_dim=64)[:2] == (-2, [-1, -1, -1])
    assert both(head_dim=256, n_heads=2, n_kv_heads=1)[:2] == (-2, [-1, -1, -1])
    assert both(E=32, k=2)[:2] == (-2, [-1, -1, -1])
    assert both(E=8, k=3)[:2] == (-2, [-1, -1, -1])
    assert both(E=8, k=2)[:2] == (-1, [-1, -1, -1])
    assert both(head_dim=100)[:2] == (-2, [-2, -2, -2])            # not a multiple of 8
    assert both(E=2, k=3)[1] == [-2, -2, -2] and b"top_k" in L.mi_last_error_detail()
    m = both()[2]
    assert L.mi_forward_generic(C.byref(m), C.byref(bt), 7, None) == -1 and b"dtype" in L.mi_last_error_detail()
    assert L.mi_workspace_bytes_generic(C.byref(m), 16, 2) > L.mi_workspace_bytes_generic(C.byref(m), 16, 0) > 4096
    assert L.mi_workspace_bytes_generic(None, 16, 0) == 0


def test_header_is_c99_and_a_c_program_can_drive_the_library(tmp_path):
    """include/mistral_hip.h must be consumable from plain C (the boundary a cgo / JNI / ctypes shim binds): compile it
    with gcc -std=c99 -pedantic, then build examples/abi_probe.c and run it against the in-tree library (argument
    validation only - no device work)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "mistral_hip.h"\nint main(void) { return MI_ABI_VERSION == 0; }\n')
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), "-c",
                    str(src), "-o", str(tmp_path / "hdr.o")], check=True)
    exe = tmp_path / "abi_probe"
    subprocess.run([gcc, "-std=c99", "-Wall", "-I", os.path.join(root, "include"),
                    os.path.join(root, "examples", "abi_probe.c"), "-ldl", "-o", str(exe)], check=True)
    from mistral_inference import _hip
    r = subprocess.run([str(exe), _hip.LIB_PATH], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "head_dim 96" in r.stdout and "workspace" in r.stdout
