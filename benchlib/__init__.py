"""Pieces of bench.py (the 1,400-line monolith split): constants of the workloads, the launcher / JSON emitter,
the roofline accounting and the CPU baseline.  bench.py keeps the argument parser, the timed runs and main().  Measurement
infrastructure, not product: nothing under 3d-sis_amd/ imports it."""
