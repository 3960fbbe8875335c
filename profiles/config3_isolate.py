"""Experiment: which ingredient of config 3 is slow (fp32 accumulator, Ng=512, per-step actions)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import configs as c
for tag, Ng, accum, modes in (("f32/f32 Ng=512 actions", 512, "float32", 3), ("f32/f32 Ng=512 no actions", 512, "float32", 0),
                              ("f32/f64 Ng=512 no actions", 512, "float64", 0), ("f32/f32 Ng=256 no actions", 256, "float32", 0),
                              ("f32/f64 Ng=256 no actions", 256, "float64", 0)):
    c.run_batched(tag, "two", 64, 1_000_000, Ng, "float32", accum, 20, actions_modes=modes)
