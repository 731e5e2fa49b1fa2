"""Deterministic synthetic workloads (frames, masks, bank contents, recipe weights) shared by the
tests, bench.py, smoke() and the golden generator.  Input generation only: no reference
arithmetic lives here (that is oracle/), and the product package never imports it."""
