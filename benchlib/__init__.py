"""Pieces of bench.py that are not its headline: the stand-in scene and fields (scene.py), the auxiliary legs (aux_legs.py: configs[2]
PropNet step, configs[4] scene sweep) and the profiled pass (profiler.py).  bench.py keeps the contract, the timed loops, the CPU
baseline (the only code outside tests/ that touches oracle/) and the JSON line."""
