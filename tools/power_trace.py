"""Power and clock of the MI355X under the decoder sweeps, SAMPLED while they run ("energy-limited" must be a
measurement in profiles/, not an inference from cycle counts).

A sampler thread reads the GPU's socket power and shader clock (hwmon / pp_dpm_sclk in sysfs; `rocm-smi --json` when sysfs has
nothing) every ~50 ms while the main thread runs, back to back for a fixed wall time each:

    idle | one-plane sweeps (asdf_decode_grid_box, N = 256) | ordinary split-half sweeps | fp32-chain sweeps

and prints per phase: launches, ms per launch (HIP events), shader clock from the kernel's own s_memtime stamps (one-plane sweeps),
mean / max sampled power, mean sampled sclk.  Usage: python tools/power_trace.py [--seconds 12] > profiles/r04_power_trace.txt
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from alignsdf_amd import synthetic as syn  # noqa: E402
from alignsdf_amd.hip_decoder import HipSdfDecoder  # noqa: E402


class Sampler(threading.Thread):
    def __init__(self, period=0.05):
        super().__init__(daemon=True)
        self.period, self.samples, self.stop_flag, self.phase = period, [], False, "idle"
        # the box is one GPU of an 8-GPU host whose other cards show up in sysfs too: pick OURS by its PCI address
        card = "card*"
        try:
            props = torch.cuda.get_device_properties(torch.cuda.current_device())
            want = "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, props.pci_device_id)
            for uevent in glob.glob("/sys/class/drm/card*/device/uevent"):
                if any(line.strip().lower().startswith("pci_slot_name=" + want) for line in open(uevent)):
                    card = uevent.split("/")[4]
        except Exception as e:                    # noqa: BLE001  (older torch without the pci fields: all cards, first one wins)
            print("# could not match the PCI address:", e)
        self.card = card
        self.power_files = sorted(glob.glob("/sys/class/drm/%s/device/hwmon/hwmon*/power1_average" % card) +
                                  glob.glob("/sys/class/drm/%s/device/hwmon/hwmon*/power1_input" % card))
        self.sclk_files = sorted(glob.glob("/sys/class/drm/%s/device/hwmon/hwmon*/freq1_input" % card))
        self.dpm_files = sorted(glob.glob("/sys/class/drm/%s/device/pp_dpm_sclk" % card))
        self.source = "sysfs" if self.power_files else "rocm-smi"
        if os.environ.get("ASDF_POWER_SOURCE") == "smi":
            self.power_files, self.source = [], "amd-smi / rocm-smi"

    def read(self):
        power = clock = None
        if self.power_files:
            try:
                power = int(open(self.power_files[0]).read()) * 1e-6
            except (OSError, ValueError):
                pass
            try:
                if self.sclk_files:
                    clock = int(open(self.sclk_files[0]).read()) * 1e-6
                elif self.dpm_files:
                    for line in open(self.dpm_files[0]):
                        if "*" in line:
                            clock = float(line.split(":")[1].strip().split("M")[0])
            except (OSError, ValueError, IndexError):
                pass
        if power is None:
            try:
                out = subprocess.run(["amd-smi", "metric", "--power", "--clock", "--json"], capture_output=True, text=True, timeout=5).stdout
                d = json.loads(out)
                d = d[0] if isinstance(d, list) else d
                d = d.get("gpu_data", [d])[0] if isinstance(d, dict) and "gpu_data" in d else d
                pw = d.get("power", {})
                for key in ("socket_power", "current_socket_power", "average_socket_power"):
                    v = pw.get(key)
                    v = v.get("value") if isinstance(v, dict) else v
                    if isinstance(v, (int, float)):
                        power = float(v)
                        break
                ck = d.get("clock", {})
                g = ck.get("gfx_0", ck.get("gfx", {}))
                v = g.get("clk") if isinstance(g, dict) else None
                v = v.get("value") if isinstance(v, dict) else v
                if isinstance(v, (int, float)):
                    clock = float(v)
            except Exception:
                pass
        if power is None:
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
                card = next(iter(json.loads(out).values()))
                for k, v in card.items():
                    if "power" in k.lower() and "(w)" in k.lower() and power is None:
                        power = float(v)
                    if "sclk clock speed" in k.lower():
                        clock = float(str(v).strip("()").lower().replace("mhz", ""))
            except Exception:
                pass
        return power, clock

    def run(self):
        while not self.stop_flag:
            p, c = self.read()
            self.samples.append((time.perf_counter(), self.phase, p, c))
            time.sleep(self.period)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=12.0)
    ap.add_argument("--grid", type=int, default=256)
    args = ap.parse_args()
    N = args.grid
    dec = HipSdfDecoder(syn.full_state_dict("nerf3"), 256, 3, "nerf")
    dec.set_sample(torch.from_numpy(syn.latent_code(0)).cuda())
    org, vs = [-1.0, -1.0, -1.0], 2.0 / (N - 1)
    dec.coarse_finish(dec.coarse_begin(N, org, vs))              # calibrates scales and allowance
    tau = dec._box_tau
    s = Sampler()
    s.start()
    print("# power / clock source:", s.source, "card", s.card, s.power_files[:1], s.sclk_files[:1] or s.dpm_files[:1])
    try:
        cap = int(open(s.power_files[0].rsplit("/", 1)[0] + "/power1_cap").read()) * 1e-6
        print("# power cap (PPT) of this card: %.0f W" % cap)
    except Exception:                             # noqa: BLE001
        pass
    rows = []

    def phase(name, launch, ticks_of=None):
        torch.cuda.synchronize()
        s.phase = name
        t_end = time.perf_counter() + args.seconds
        ms, ticks = [], []
        while time.perf_counter() < t_end:
            evs = []
            for _ in range(8):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = launch()
                e1.record()
                evs.append((e0, e1, out))
            torch.cuda.synchronize()
            for e0, e1, out in evs:
                ms.append(e0.elapsed_time(e1))
                if ticks_of is not None:
                    t = ticks_of(out)
                    if t:
                        ticks.append(t)
        s.phase = "gap"
        time.sleep(1.0)
        rows.append((name, len(ms), float(np.median(ms)) if ms else 0.0, float(np.median(ticks)) if ticks else None))

    def rec_ticks(out):
        t = out[0][28:32].cpu().numpy().view(np.int64)
        return int(t[1] - t[0])

    s.phase = "idle"
    time.sleep(3.0)
    rows.append(("idle", 0, 0.0, None))
    dec.set_audit(0)
    phase("one-plane sweep (sdf_mlp_f16p1_kernel, box entry point, no audit)", lambda: dec._box_launch(N, org, vs, 0, True, True, tau), rec_ticks)
    phase("ordinary split-half sweep (sdf_mlp_f16_kernel)", lambda: dec.decode_grid(N, org, vs, check_range=False))
    dec.set_math("f32")
    phase("fp32 chain sweep (sdf_mlp_kernel)", lambda: dec.decode_grid(N, org, vs, check_range=False))
    s.stop_flag = True
    s.join()
    print("# %-75s %9s %10s %12s %10s %10s %10s %8s" % ("phase", "launches", "ms/launch", "clock(ticks)", "P mean W", "P max W", "sclk MHz", "samples"))
    for name, n, ms, ticks in rows:
        sel = [(p, c) for _, ph, p, c in s.samples if ph == name and p is not None]
        pw = [p for p, _ in sel]
        ck = [c for _, c in sel if c is not None]
        clock = "%.3f GHz" % (ticks / (ms * 1e6)) if ticks and ms else "-"
        print("  %-75s %9d %10.3f %12s %10.1f %10.1f %10.0f %8d" % (name, n, ms, clock, np.mean(pw) if pw else float("nan"),
                                                                    max(pw) if pw else float("nan"), np.mean(ck) if ck else float("nan"), len(sel)))
    dec.close()


if __name__ == "__main__":
    main()
