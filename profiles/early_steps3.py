"""The slow first steps follow an IDLE DEVICE, not a new handle (early_steps2.py: the same handle is slow again after 50 ms of
idleness).  What brings the device to its working state, and how long may it rest before it has left it?   (round 4)

    python3 profiles/early_steps3.py [timeline.csv]

One config-2 handle; every case starts from `rest` seconds of idleness, then does something, then times steps in chunks of 10
with no pause in between:

  cold               nothing in between                                         -- the slow start
  probe N            N passes of the copy probe (pic_stream_probe, a bare read-modify-write stream)
  steps N            N steps (one call), not timed
  small T            T seconds of the resident schedule on 256 small environments (all CUs busy, almost no HBM traffic)
  gap G              60 steps, then G seconds of idleness                       -- how long a rest is "idle"

The sysfs sensors of this device (clock levels, socket power) are sampled every millisecond; the cold case's samples go to
the CSV.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import torch
    from early_steps2 import Sampler, device_pci
    sam = Sampler(period=0.001, pci=device_pci(torch))
    print("sensors:", sam.card, sorted(sam.files), flush=True)
    sam.start()
    import bench
    from ocplasma_amd.env.batched import BatchedPIC
    E, N, Ng, L = 64, 1_000_000, 256, 50.0
    x0, v0 = bench.synth_bump_on_tail_device(torch, E, N, L, torch.float64, "cuda:0", seed=1234)
    torch.cuda.synchronize()
    env = BatchedPIC(E, N, Ng, L=L, dt=0.1)
    print("placement_info", env._h.placement_info(), flush=True)
    env.reset_device(x0.data_ptr(), v0.data_ptr()); env.sync()
    small = BatchedPIC(256, 5000, 250, L=L, dt=0.1)
    small.reset_sampled("two-stream", seed=3); small.sync()
    env.step(None, 300); env.sync()

    def chunks(label, n=8):
        out, t_begin = [], time.perf_counter()
        for _ in range(n):
            t = time.perf_counter()
            env.step(None, 10)
            env.sync()
            out.append((time.perf_counter() - t) / 10 * 1e6)
        print(f"{label:52s} us/step per 10: {[round(o, 1) for o in out]}", flush=True)
        return t_begin, time.perf_counter()

    rest = 0.5
    time.sleep(rest)
    t0, t1 = chunks("cold (0.5 s idle)", n=20)
    if len(sys.argv) > 1:
        keys = sorted(sam.files)
        with open(sys.argv[1], "w") as f:
            f.write("ms_since_first_step," + ",".join(keys) + "\n")
            for r in sam.rows:
                if t0 - 0.05 <= r["t"] <= t1 + 0.02:
                    f.write(f"{(r['t'] - t0) * 1e3:.2f}," + ",".join(str(r.get(k, "")).replace(",", ";") for k in keys) + "\n")
    for passes in (100, 300, 1000):
        time.sleep(rest)
        t = time.perf_counter()
        gbs = env.stream_probe(passes)
        ms = (time.perf_counter() - t) * 1e3
        chunks(f"probe {passes} passes ({ms:.0f} ms, {gbs:.0f} GB/s), then")
    for k in (10, 30, 100):
        time.sleep(rest)
        env.step(None, k)
        chunks(f"steps {k} untimed, then")
    for T in (0.05, 0.3):
        time.sleep(rest)
        t = time.perf_counter()
        while time.perf_counter() - t < T:
            small.step(None, 200)
            small.sync()
        chunks(f"small resident environments for {T} s, then")
    for G in (0.001, 0.003, 0.010, 0.030, 0.100):
        time.sleep(rest)
        env.step(None, 100); env.sync()
        time.sleep(G)
        chunks(f"100 steps, {G * 1e3:.0f} ms rest, then", n=5)
    # the same rests without the host going to sleep (a spinning host thread: is it the DEVICE's idleness?)
    for G in (0.010, 0.100):
        env.step(None, 100); env.sync()
        t = time.perf_counter()
        while time.perf_counter() - t < G:
            pass
        chunks(f"100 steps, {G * 1e3:.0f} ms host spin, then", n=5)
    sam.stop_flag = True
    small.close()
    env.close()


if __name__ == "__main__":
    main()
