"""Why are the first ~30 steps of a new handle slow at config 2?  (round 4)

    python3 profiles/early_steps2.py <auto|off> [scenario ...]

Step time by chunks of 10 steps, with the device's clocks / power sampled from sysfs by a side thread every ~2 ms, through a
sequence of situations that separate the candidate causes:

  first     new handle (placement as given), first reset                      -- the slow start of DESIGN 6
  again     second reset of the same handle                                    -- fast from step 1 in round 3
  idle      the same handle after 1 s of an idle device                        -- an idle device's clocks alone
  second    the handle destroyed, a NEW handle in the same process             -- new memory, warm device
  other     a second handle beside a stepping one (its memory has never been stepped on, the device is at work)
"""
import glob
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class Sampler(threading.Thread):
    """current sclk / mclk / fclk level and the power sensor of card 0's sysfs, with perf_counter stamps"""

    def __init__(self, period=0.002, pci=None):
        """pci: '0000:05:00.0'-style address of the device to watch (card numbers in sysfs are the host's, not this
        process's device ordinals); None = the first card that has the files"""
        super().__init__(daemon=True)
        self.period, self.rows, self.stop_flag = period, [], False
        self.files, self.card = {}, None
        for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
            if not os.path.exists(os.path.join(dev, "pp_dpm_sclk")):
                continue
            if pci is not None and os.path.basename(os.path.realpath(dev)).lower() != pci.lower():
                continue
            self.card = dev
            for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "gpu_busy_percent", "mem_busy_percent"):
                p = os.path.join(dev, name)
                if os.path.exists(p):
                    self.files[name] = p
            for hw in glob.glob(os.path.join(dev, "hwmon/hwmon*")):
                for name in ("power1_input", "power1_average", "freq1_input", "freq2_input", "temp1_input", "temp3_input"):
                    p = os.path.join(hw, name)
                    if os.path.exists(p):
                        self.files[name] = p
            break

    @staticmethod
    def _cur(text):
        lines = [l for l in text.splitlines() if l.strip()]
        star = [l for l in lines if l.rstrip().endswith("*")]
        return (star[0] if star else (lines[0] if lines else "")).replace("*", "").strip()

    def run(self):
        while not self.stop_flag:
            row = {"t": time.perf_counter()}
            for name, p in self.files.items():
                try:
                    with open(p) as f:
                        row[name] = self._cur(f.read())
                except OSError as e:
                    row[name] = f"err {e.errno}"
            self.rows.append(row)
            time.sleep(self.period)

    def window(self, t0, t1):
        """distinct values seen in [t0, t1] per sensor, in order of first appearance"""
        out = {}
        for r in self.rows:
            if t0 <= r["t"] <= t1:
                for k, v in r.items():
                    if k != "t":
                        out.setdefault(k, [])
                        if not out[k] or out[k][-1] != v:
                            out[k].append(v)
        return {k: (v if len(v) <= 6 else v[:3] + ["..."] + v[-2:]) for k, v in out.items()}


def device_pci(torch, index=0):
    """PCI address of cuda:<index> as sysfs spells it"""
    p = torch.cuda.get_device_properties(index)
    try:
        return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except AttributeError:
        return None


def main():
    placement = sys.argv[1] if len(sys.argv) > 1 else "auto"
    scenarios = sys.argv[2:] or ["first", "again", "idle", "second", "other"]
    import torch
    sam = Sampler(pci=device_pci(torch))
    print("sensors:", sam.card, sorted(sam.files), flush=True)
    sam.start()
    import bench
    from ocplasma_amd.env.batched import BatchedPIC
    E, N, Ng, L = 64, 1_000_000, 256, 50.0
    x0, v0 = bench.synth_bump_on_tail_device(torch, E, N, L, torch.float64, "cuda:0", seed=1234)
    torch.cuda.synchronize()

    def chunks(env, label, n=12, sensors=True):
        out, t_begin = [], time.perf_counter()
        for _ in range(n):
            t = time.perf_counter()
            env.step(None, 10)
            env.sync()
            out.append((time.perf_counter() - t) / 10 * 1e6)
        t_end = time.perf_counter()
        print(f"{label}: us/step per chunk of 10: {[round(o, 1) for o in out]}", flush=True)
        if sensors:
            print(f"    sensors, first 30 ms: {sam.window(t_begin, t_begin + 0.030)}", flush=True)
            print(f"    sensors, last 30 ms:  {sam.window(t_end - 0.030, t_end)}", flush=True)

    def new_env():
        t = time.perf_counter()
        env = BatchedPIC(E, N, Ng, L=L, dt=0.1, placement=placement)
        print(f"  create ({placement}): {1e3 * (time.perf_counter() - t):.1f} ms, placement_info {env._h.placement_info()}", flush=True)
        return env

    env = None
    for sc in scenarios:
        if sc == "first":
            env = new_env()
            env.reset_device(x0.data_ptr(), v0.data_ptr()); env.sync()
            chunks(env, "first handle, first reset")
        elif sc == "again":
            env.reset_device(x0.data_ptr(), v0.data_ptr()); env.sync()
            chunks(env, "same handle, second reset")
        elif sc == "idle":
            time.sleep(1.0)
            chunks(env, "same handle after 1 s idle")
            time.sleep(0.05)
            chunks(env, "same handle after 50 ms idle", n=6)
        elif sc == "second":
            env.close()
            env = new_env()
            env.reset_device(x0.data_ptr(), v0.data_ptr()); env.sync()
            chunks(env, "NEW handle in the warm process, first reset")
        elif sc == "other":
            env.step(None, 50)                      # the device is at work while the other handle is made
            env2 = new_env()
            env2.reset_device(x0.data_ptr(), v0.data_ptr()); env2.sync(); env.sync()
            chunks(env2, "second handle beside the first, first reset")
            env2.close()
        elif sc == "wait":
            # the first handle again, but with 2 s between create and the first step: anything that runs in the background
            # after pic_create (a wipe of what the search released) has finished by then
            env = new_env()
            env.reset_device(x0.data_ptr(), v0.data_ptr()); env.sync()
            time.sleep(2.0)
            chunks(env, "first handle, 2 s after create")
    sam.stop_flag = True
    if env is not None:
        env.close()


if __name__ == "__main__":
    main()
