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
