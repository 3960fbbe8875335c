"""Leave the device the way a finished test suite leaves it: take `GB` gigabytes in 8 GB pieces, write to all of them, exit.
What the driver wipes afterwards (released memory is cleared before it is handed out again) is what the next process's
hipMalloc waits for.    python3 profiles/hog.py [GB]"""
import sys
import time
import torch
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 200.0
t = time.perf_counter()
blocks = []
while sum(b.numel() for b in blocks) < gb * 1e9:
    blocks.append(torch.empty(8 << 30, dtype=torch.uint8, device="cuda:0").fill_(1))
torch.cuda.synchronize()
print(f"held and wrote {sum(b.numel() for b in blocks) / 1e9:.0f} GB in {time.perf_counter() - t:.1f} s; free now "
      f"{torch.cuda.mem_get_info()[0] / 1e9:.0f} GB", flush=True)
