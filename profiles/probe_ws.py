"""Experiment: in-place streaming copy bandwidth vs working-set size (is the Infinity Cache worth
scheduling for?).  PICSTEP_PROBE_MB = MB per array; two arrays are read and written."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ocplasma_amd
from ocplasma_amd.env.batched import BatchedPIC
env = BatchedPIC(1, 100000, 256)
for mb in (8, 16, 32, 48, 64, 96, 128, 192, 256, 384, 512, 1024):
    os.environ["PICSTEP_PROBE_MB"] = str(mb)
    g = [env.stream_probe(20) for _ in range(3)]
    print(f"working set 2 x {mb:5d} MB : {max(g):7.0f} GB/s (runs {[round(x) for x in g]})", flush=True)
