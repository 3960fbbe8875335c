"""Further seeds and shapes of tests/test_gpu_api_fuzz.py (random ABI call sequences against an oracle-backed model), as a one-off
robustness run: python profiles/fuzz_more.py [first_seed] [count]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import test_gpu_api_fuzz as t

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 40)
shapes = [(513, 33), (700, 33), (2049, 96), (3000, 96), (5000, 250), (8192, 128), (9000, 128), (20000, 250), (5121, 250), (4096, 128)]
n = 0
for seed in range(first, first + count):
    N, Ng = shapes[seed % len(shapes)]
    for bpe in ((0, -1, 2) if N <= 8192 else (0, 3)):      # automatic schedule, resident forced, sweeps forced
        t.test_random_call_sequences(seed * 7 + bpe + 1, N, Ng, bpe)
        n += 1
print("fuzz sequences passed:", n, flush=True)
