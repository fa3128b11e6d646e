"""bench.py's algorithmic byte / flop accounting against the figures of SURVEY.md section 8(d) (the numbers the roofline
fractions are computed from)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_decode_bytes_per_token_matches_survey_table():
    b7 = bench.decode_bytes_per_token(dict(bench.MISTRAL_7B), 4096)
    assert abs(b7 / 1e9 - 14.76) < 0.02                       # cfg 2: 14.23 GB weights + 0.537 GB KV
    nemo = bench.decode_bytes_per_token(dict(bench.PRESETS["nemo-12b"][0]), 8192)
    assert abs(nemo / 1e9 - 24.50) < 0.05                     # cfg 3 at ctx 8192
    mix = bench.decode_bytes_per_token(dict(bench.PRESETS["mixtral-8x7b"][0]), 4096)
    assert abs(mix / 1e9 - 26.03) < 0.06                      # cfg 4: 2 of 8 experts
    # the ring caps the KV term at the sliding window
    assert bench.decode_bytes_per_token(dict(bench.MISTRAL_7B), 100000) == bench.decode_bytes_per_token(dict(bench.MISTRAL_7B), 4096)


def test_prefill_flops_matches_survey_table():
    f7 = bench.prefill_flops(dict(bench.MISTRAL_7B), 4096)
    assert abs(f7 / 1e12 - 62.7) < 0.4                        # 57.2 (layers) + 1.1 (LM head) + 4.4 (attention)
    mix = bench.prefill_flops(dict(bench.PRESETS["mixtral-8x7b"][0]), 4096)
    assert abs(mix / 1e12 - 108.9) < 1.5


def test_cpu_baseline_carries_the_stored_reference_number(monkeypatch):
    """On a box without the reference source (the GPU lease) the baseline is the oracle port's, and - for the headline dims
    only - the unmodified reference's number measured once on the build container rides along in the same object."""
    port = {"value": 2.0, "unit": "tokens/s", "cores": 8, "kind": "port", "sample": "stub"}
    monkeypatch.setattr(bench, "reference_baseline", lambda *a, **k: None)
    monkeypatch.setattr(bench, "port_baseline", lambda *a, **k: dict(port))
    out = bench.cpu_baseline(dict(bench.MISTRAL_7B), 4096)
    assert out["kind"] == "port" and out["value"] == 2.0
    assert out["reference_container_value"] == 0.739 and out["reference_container_cores"] == 8
    other = bench.cpu_baseline(dict(bench.PRESETS["mixtral-8x7b"][0]), 4096)
    assert "reference_container_value" not in other
    # where the reference can be timed it IS the baseline, with the port's number alongside
    monkeypatch.setattr(bench, "reference_baseline", lambda *a, **k: {"value": 0.7, "kind": "reference", "cores": 8})
    ref = bench.cpu_baseline(dict(bench.MISTRAL_7B), 4096)
    assert ref["kind"] == "reference" and ref["port_value"] == 2.0


def test_plain_multi_gpu_invocation_re_executes_under_torchrun(monkeypatch):
    """`python bench.py --gpus 4` without a torchrun environment must not die on a WORLD_SIZE assertion: it re-executes
    itself as the documented torch.distributed.run line (one process per GPU, loopback rendezvous)."""
    import subprocess
    seen = {}

    class R:
        returncode = 0

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5"])
    assert bench.respawn_under_torchrun(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "4", "--steps", "5"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_stage_bytes_without_lm_head():
    """A pipeline stage that does not own the LM head reads (vocab x dim + dim) bf16 less per token (bench.stage_roofline)."""
    p = dict(bench.MISTRAL_7B)
    full, stage = bench.decode_bytes_per_token(p, 4096), bench.decode_bytes_per_token(p, 4096, head=False)
    assert full - stage == 2 * (p["vocab_size"] * p["dim"] + p["dim"])


def test_sub_measurement_reports_an_error_instead_of_raising(monkeypatch):
    """The nemo / mixtral sub-objects of the N = 1 line come from subprocesses of this script: a crash or a time-out there must
    cost the sub-object, never the headline line."""
    import subprocess

    class R:
        stdout, returncode = "not json\n", 1
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: R())
    out = bench.sub_measurement("nemo-12b", 8192, 4, 2)
    assert set(out) == {"error", "wall_s"}

    # a child that prints a JSON line with other keys than expected is an error of the sub-object too, not a
    # KeyError in the parent (the field extraction sits inside the same try)
    class R2:
        stdout, returncode = json.dumps({"value": 1.0, "config": {}}) + "\n", 0
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: R2())
    out = bench.sub_measurement("nemo-12b", 8192, 4, 2)
    assert "KeyError" in out["error"]

    # a well-formed child line (batch form: bytes per STEP, no prefill object) is reduced to the sub-object's fields
    class R3:
        stdout, returncode = json.dumps({"value": 800.0, "ms_per_step": 3.7, "steps": 32, "warmup": 2, "per_sequence_tokens_per_s": 266.7,
                                         "config": {"workload": "w", "context_at_timing": 4100, "decode_launch": "launch path"},
                                         "hbm_roofline_step": {"frac": 0.52, "bytes_per_step": 123}}) + "\n", 0
    seen = {}
    def run3(cmd, **k):
        seen["cmd"] = cmd
        return R3()
    monkeypatch.setattr(subprocess, "run", run3)
    out = bench.sub_measurement("mistral-7b", 4096, 32, 2, extra=("--batch", "3"))
    assert out["tokens_per_s"] == 800.0 and out["bytes_per_step"] == 123 and "error" not in out and seen["cmd"][-2:] == ["--batch", "3"]

    def boom(*a, **k):
        raise subprocess.TimeoutExpired("bench.py", 1)
    monkeypatch.setattr(subprocess, "run", boom)
    assert "TimeoutExpired" in bench.sub_measurement("mixtral-8x7b", 4096, 4, 2)["error"]


def test_reference_bytecode_recipe():
    """oracle/build_ref.py: the unmodified reference byte-compiled into oracle/_ref/ (what lets bench.py's cpu_baseline time the
    reference itself on the GPU box).  Where /root/reference exists: rebuild, then import the SOURCELESS package through the shim."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.isdir("/root/reference/src/mistral_inference"):
        import pytest
        pytest.skip("no reference source on this machine (the GPU box uses the prebuilt oracle/_ref)")
    r = subprocess.run([sys.executable, os.path.join(root, "oracle", "build_ref.py")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    code = ("import sys, os; sys.path[:0] = [%r, %r]; import mistral_inference.transformer as t, mistral_inference.cache as c; "
            "assert t.__file__.endswith('.pyc') and c.__file__.endswith('.pyc'), t.__file__; "
            "assert not os.path.exists(t.__file__[:-1]); print('ok')"
            % (os.path.join(root, "oracle", "shim"), os.path.join(root, "oracle", "_ref")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-1500:]
