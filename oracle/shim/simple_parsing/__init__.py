"""Stand-in for simple-parsing 0.1.5 (reference args.py:4, moe.py:6, lora.py:9) -- test infrastructure only."""
