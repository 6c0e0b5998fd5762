"""Clock and power of a rank's own GPU beside a timed region, from sysfs (bench.py's `per_rank[*].gpu`)."""
import glob
import os
import threading
import time

import torch


class GpuTelemetry:
    """Shader clock and package power of THIS rank's GPU, sampled from sysfs on a side thread - file reads only, never a HIP or SMI
    call, 0.2 - 0.5 ms each (tools/probes/sysfs_probe.py) - and summarised over named windows of the host clock.  Why it is in the
    line: round 5 showed that throughput follows the clock the firmware grants under the package power limit one to one (2311 vs
    2238 MHz for the two batch schedules at the same energy per megapixel), so a multi-GPU line whose
    ranks differ, or whose per-GPU value differs from the N = 1 line's, can be read against `sclk_mhz` / `power_w` / `power_cap_w`
    per rank instead of being guessed at.  A device whose sysfs node is missing or unreadable gives {"available": false, ...}."""

    def __init__(self, dev_index, interval=0.01, hwmon_dir=None):
        """``dev_index``: HIP device index of this process (its PCI address names the sysfs node: the box shows all eight GPUs of its
        host under /sys/class/drm whatever HIP may see); ``hwmon_dir`` (tests): a directory holding freq1_input / power1_input /
        power1_cap instead."""
        self.samples, self.windows, self.interval = [], {}, interval
        self.stop_flag, self.thread, self.cap_w, self.why = False, None, None, None
        try:
            if hwmon_dir is not None:
                self.slot, hw = str(hwmon_dir), [str(hwmon_dir)]
            else:
                p = torch.cuda.get_device_properties(dev_index)
                slot = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
                self.slot = slot
                hw = sorted(glob.glob(f"/sys/bus/pci/devices/{slot}/hwmon/hwmon*"))
                if not hw:
                    raise FileNotFoundError(f"no hwmon node under /sys/bus/pci/devices/{slot}")
            self.f_clk, self.f_pow = os.path.join(hw[0], "freq1_input"), os.path.join(hw[0], "power1_input")
            if not os.path.exists(self.f_pow):
                self.f_pow = os.path.join(hw[0], "power1_average")
            self._read(self.f_clk)
            try:
                self.cap_w = self._read(os.path.join(hw[0], "power1_cap")) / 1e6
            except Exception:
                pass
        except Exception as e:
            self.why = f"{type(e).__name__}: {e}"

    @staticmethod
    def _read(path):
        with open(path) as f:
            return int(f.read())

    def start(self):
        if self.why is None:
            self.thread = threading.Thread(target=self._run, name="gpu-telemetry", daemon=True)
            self.thread.start()
        return self

    def _run(self):
        while not self.stop_flag:
            try:
                self.samples.append((time.perf_counter(), self._read(self.f_clk) / 1e6, self._read(self.f_pow) / 1e6))
            except Exception:
                pass
            time.sleep(self.interval)

    def window(self, name, t0, t1):
        self.windows[name] = (t0, t1)

    def stop(self):
        self.stop_flag = True
        if self.thread is not None:
            self.thread.join(timeout=1.0)
        if self.why is not None:
            return {"available": False, "why": self.why}
        out = {"available": True, "source": f"sysfs hwmon of {self.slot} (freq1_input, power1_input), one sample per {self.interval * 1e3:.0f} ms on a side thread",
               "power_cap_w": self.cap_w}
        for name, (t0, t1) in self.windows.items():
            rows = [r for r in self.samples if t0 <= r[0] <= t1]
            if not rows:
                out[name] = {"samples": 0, "seconds": round(t1 - t0, 3)}
                continue
            clk, pw = sorted(r[1] for r in rows), sorted(r[2] for r in rows)
            out[name] = {"samples": len(rows), "seconds": round(t1 - t0, 3), "sclk_mhz": {"median": round(clk[len(clk) // 2]), "min": round(clk[0]), "max": round(clk[-1])},
                         "power_w": {"median": round(pw[len(pw) // 2]), "max": round(pw[-1])}}
        return out
