"""Experiment: in-place streaming copy with dependent fp64 work between load and store."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ocplasma_amd
from ocplasma_amd.env.batched import BatchedPIC
env = BatchedPIC(1, 100000, 256)
for mb in (96, 512):
    os.environ["PICSTEP_PROBE_MB"] = str(mb)
    for work in (0, 10, 25, 50, 100, 200, 400):
        os.environ["PICSTEP_PROBE_WORK"] = str(work)
        g = [env.stream_probe(20) for _ in range(3)]
        print(f"2 x {mb:4d} MB, {4*work:5d} dependent FMAs per lane-iteration : {max(g):7.0f} GB/s", flush=True)
