#!/usr/bin/env python3
"""What the GPU box's sysfs offers an unprivileged process about its AMD GPUs (clock, power, power cap, identity), and how a HIP
device index maps onto a /sys/class/drm card: input for bench.py's telemetry sampler.   python tools/probes/sysfs_probe.py"""
import glob
import os
import time


def rd(p):
    try:
        with open(p) as f:
            return f.read().strip()
    except Exception as e:
        return f"<{type(e).__name__}>"


for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
    if "-" in os.path.basename(card):
        continue
    dev = os.path.join(card, "device")
    print("==", card, "vendor", rd(os.path.join(dev, "vendor")), "device", rd(os.path.join(dev, "device")))
    print("  uevent:", rd(os.path.join(dev, "uevent")).replace("\n", " | "))
    print("  unique_id:", rd(os.path.join(dev, "unique_id")))
    print("  pp_dpm_sclk:", rd(os.path.join(dev, "pp_dpm_sclk")).replace("\n", " | "))
    print("  gpu_busy_percent:", rd(os.path.join(dev, "gpu_busy_percent")))
    print("  gpu_metrics bytes:", len(open(os.path.join(dev, "gpu_metrics"), "rb").read()) if os.path.exists(os.path.join(dev, "gpu_metrics")) else None)
    for hw in sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*"))):
        for f in sorted(os.listdir(hw)):
            if f.startswith(("power1", "freq1", "freq2", "temp1_input")):
                print(f"  {os.path.basename(hw)}/{f}: {rd(os.path.join(hw, f))}")
try:
    import torch

    for i in range(torch.cuda.device_count()):
        p = torch.cuda.get_device_properties(i)
        print("hip device", i, p.name, "uuid", getattr(p, "uuid", None), "pci", [getattr(p, a, None) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")])
    # cost of one sample and whether the values move under load
    x = torch.randn(8192, 8192, device="cuda")
    card = sorted(c for c in glob.glob("/sys/class/drm/card[0-9]*") if "-" not in os.path.basename(c) and rd(os.path.join(c, "device", "vendor")) == "0x1002")[0]
    hw = sorted(glob.glob(os.path.join(card, "device", "hwmon", "hwmon*")))[0]
    for phase in ("idle", "busy", "busy", "idle"):
        if phase == "busy":
            for _ in range(200):
                y = x @ x
        t0 = time.perf_counter()
        s = rd(os.path.join(card, "device", "pp_dpm_sclk"))
        pw = {f: rd(os.path.join(hw, f)) for f in ("power1_average", "power1_input", "power1_cap", "freq1_input")}
        dt = time.perf_counter() - t0
        print(phase, f"{dt * 1e6:.0f} us per sample:", s.replace("\n", " | "), pw)
        torch.cuda.synchronize()
except Exception as e:
    print("torch part failed:", type(e).__name__, e)
