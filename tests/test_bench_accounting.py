"""bench.py's algorithmic byte / flop accounting against the figures of SURVEY.md section 8(d) (the numbers the roofline
fractions are computed from)."""
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
