"""Board power of the GPU a measurement runs on — measurement plumbing shared by bench.py and tools/ab.py (no counterpart in the
reference).  The flow kernels are POWER-limited on MI355X (socket at 1.28-1.39 kW of the 1.4 kW limit, shader
clock below boost), so what an optimisation buys is what it saves in joules per query; every timing is therefore reported
next to the energy of the same launches.

Source, in order of preference: the amdgpu hwmon files in sysfs (``power1_average`` / ``power1_input`` in microwatts,
``freq1_input`` = sclk in Hz; a read costs microseconds, so the poll runs every 20 ms), else ``rocm-smi --showpower --showclocks
--json`` (a Python program: ~0.3 s per sample).  Everything returns None-filled results where neither exists."""
from __future__ import annotations

import glob
import json
import os
import subprocess
import threading
import time


def _hwmon_dirs(pci_bus_id=None):
    out = []
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
        if "-" in os.path.basename(card):
            continue
        dev = os.path.join(card, "device")
        try:
            if open(os.path.join(dev, "vendor")).read().strip() != "0x1002":
                continue
        except OSError:
            continue
        bus = None
        try:
            bus = int(os.path.basename(os.path.realpath(dev)).split(":")[1], 16)
        except (IndexError, ValueError):
            pass
        for hw in sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*"))):
            if any(os.path.exists(os.path.join(hw, f)) for f in ("power1_average", "power1_input")):
                out.append((bus, hw))
    if pci_bus_id is not None and any(b == pci_bus_id for b, _ in out):
        out = [x for x in out if x[0] == pci_bus_id]
    return [hw for _, hw in out]


class PowerSampler:
    """Polls the board's socket power (W) and shader clock (MHz) on a thread between ``start()`` and ``stop()``."""

    def __init__(self, pci_bus_id=None, allow_rocm_smi=True):
        if pci_bus_id is None:   # sysfs shows every GPU of the node, the process usually sees one: take the current device's
            try:
                import torch
                if torch.cuda.is_available():
                    pci_bus_id = getattr(torch.cuda.get_device_properties(torch.cuda.current_device()), "pci_bus_id", None)
            except Exception:
                pci_bus_id = None
        self.pci_bus_id = pci_bus_id
        hw = _hwmon_dirs(pci_bus_id)
        self.hw = hw[0] if hw else None
        self.source = None
        if self.hw:
            self.pfile = next(os.path.join(self.hw, f) for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(self.hw, f)))
            self.ffile = os.path.join(self.hw, "freq1_input") if os.path.exists(os.path.join(self.hw, "freq1_input")) else None
            try:
                float(open(self.pfile).read())
                self.source = "sysfs hwmon " + os.path.basename(self.pfile)
            except (OSError, ValueError):
                self.hw = None
        if self.source is None and allow_rocm_smi:
            self.source = "rocm-smi"
        self.period = 0.02 if self.hw else 0.3
        self.samples = []
        self._stop = threading.Event()
        self._th = None

    def read(self):
        """(watts, sclk MHz) now; None where unavailable."""
        if self.hw:
            try:
                w = float(open(self.pfile).read()) * 1e-6
                mhz = float(open(self.ffile).read()) * 1e-6 if self.ffile else None
                return w, mhz
            except (OSError, ValueError):
                return None, None
        if self.source == "rocm-smi":
            try:
                r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=15)
                d = next(iter(json.loads(r.stdout).values()))
                w = next((float(v) for k, v in d.items() if "Power" in k and "(W)" in k), None)
                mhz = next((float("".join(c for c in v if c.isdigit() or c == ".")) for k, v in d.items() if k.startswith("sclk clock speed")), None)
                return w, mhz
            except Exception:
                return None, None
        return None, None

    def _poll(self):
        while not self._stop.is_set():
            self.samples.append((time.perf_counter(),) + self.read())
            self._stop.wait(self.period)

    def start(self):
        self.samples, self._stop = [], threading.Event()
        self._th = threading.Thread(target=self._poll, daemon=True)
        self._th.start()

    def stop(self):
        self._stop.set()
        if self._th:
            self._th.join(timeout=20)
        ws = sorted(w for _, w, _ in self.samples if w is not None)
        cs = sorted(c for _, _, c in self.samples if c is not None)
        med = lambda v: v[len(v) // 2] if v else None  # noqa: E731
        return {"socket_power_w": med(ws), "socket_power_w_min": ws[0] if ws else None, "socket_power_w_max": ws[-1] if ws else None,
                "sclk_mhz": med(cs), "samples": len(self.samples), "source": self.source, "pci_bus_id": self.pci_bus_id}


def energy_probe(run, queries_per_call, seconds=1.5, sync=None, pci_bus_id=None, warm_calls=4):
    """Energy of a workload: ``run(k)`` is called back to back for ``seconds`` (after ``warm_calls`` untimed ones) while the board
    power is polled; -> {socket_power_w (median), sclk_mhz, seconds, calls, queries, joule_per_Mquery = median W x elapsed s /
    (queries / 1e6), ...}.  ``sync()`` must block until the GPU is idle (torch.cuda.synchronize)."""
    ps = PowerSampler(pci_bus_id)
    for k in range(warm_calls):
        run(k)
    if sync:
        sync()
    ps.start()
    t0 = time.perf_counter()
    calls = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(4):
            run(warm_calls + calls)
            calls += 1
        if sync:
            sync()
    dt = time.perf_counter() - t0
    res = ps.stop()
    q = calls * queries_per_call
    res.update(seconds=dt, calls=calls, queries=q, Mqueries_per_s=q / dt / 1e6,
               joule_per_Mquery=(res["socket_power_w"] * dt / (q / 1e6)) if res["socket_power_w"] and q else None,
               board_limit_w=1400, boost_clock_mhz=2400)
    return res
