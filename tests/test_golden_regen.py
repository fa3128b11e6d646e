"""The golden recipe is reproducible: oracle/make_golden.py, re-run against the unmodified reference with every
`torch.empty` float buffer POISONED with NaN (the reference allocates its K/V cache with torch.empty, cache.py:163-167;
real xformers never reads the padded keys, cache.py:249-254), regenerates tests/golden bit for bit.

Runs only where the reference source exists (the build container); the regeneration happens in a subprocess because it
imports the reference under the package name the product also uses."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MISTRAL_REFERENCE_SRC", "/root/reference/src")

SCRIPT = textwrap.dedent("""
    import os, sys, torch
    sys.path.insert(0, os.path.join({root!r}, "oracle"))
    _empty = torch.empty
    def poisoned(*a, **k):
        t = _empty(*a, **k)
        if t.is_floating_point() and t.numel() and t.device.type == "cpu":
            t.fill_(float("nan"))
        return t
    torch.empty = poisoned
    import make_golden
    from safetensors.torch import load_file
    make_golden.main({out!r})
    bad = []
    for name in make_golden.CASES:
        new = load_file(os.path.join({out!r}, name + ".safetensors"))
        old = load_file(os.path.join({root!r}, "tests", "golden", name + ".safetensors"))
        assert set(new) == set(old), name
        for k in old:
            a, b = new[k], old[k]
            # logprobs are NaN-padded to a rectangle by the recipe itself; everything else must be NaN-free
            if k != "logprobs":
                assert not torch.isnan(a.double()).any(), (name, k)
            if a.dtype != b.dtype or a.shape != b.shape or not torch.equal(torch.nan_to_num(a.double(), nan=7.0),
                                                                            torch.nan_to_num(b.double(), nan=7.0)):
                bad.append((name, k))
    assert not bad, bad
    print("REGEN_OK", len(make_golden.CASES))
""")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "mistral_inference")), reason="reference source not present")
def test_goldens_regenerate_bit_for_bit_with_poisoned_empty(tmp_path):
    code = SCRIPT.format(root=ROOT, out=str(tmp_path))
    env = dict(os.environ, MISTRAL_REFERENCE_SRC=REF)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "REGEN_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
