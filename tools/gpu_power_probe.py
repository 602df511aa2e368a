#!/usr/bin/env python3
"""Board power and shader clock (sysfs hwmon, read-only) sampled while the card runs (a) the bulk tile back to back at K = 256, (b) at
K = 512, (c) the same two on all-zero operands (timing is value-independent, power is not: most of what an RBF matrix's trailing update
multiplies is zero), (d) the pipelined batch of evaluations (20 in flight, n = 8192), (e) one factorisation after another: is the batch at
the power cap like the exclusive kernel on random data, and at which clock?  Each phase runs >= 2 s; samples every ~10 ms from a second thread."""
import glob
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsum_amd  # noqa: E402
from sklearn.gaussian_process.kernels import RBF  # noqa: E402


def find_sensors():
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        hw = glob.glob(os.path.join(card, "hwmon", "hwmon*"))
        if not hw:
            continue
        for name in ("power1_average", "power1_input"):
            p = os.path.join(hw[0], name)
            if os.path.exists(p):
                out.setdefault(card, {})["power"] = p
                break
        f = os.path.join(hw[0], "freq1_input")
        if os.path.exists(f):
            out.setdefault(card, {})["sclk"] = f
        cap = os.path.join(hw[0], "power1_cap")
        if os.path.exists(cap):
            out.setdefault(card, {})["cap"] = cap
        for name in ("mem_busy_percent", "gpu_busy_percent"):
            q = os.path.join(card, name)
            if os.path.exists(q):
                out.setdefault(card, {})[name] = q
    return out


def read(path):
    try:
        with open(path) as f:
            return float(f.read().strip())
    except Exception:      # noqa: BLE001
        return float("nan")


sensors = find_sensors()
print("sensors:", {k: sorted(v) for k, v in sensors.items()}, flush=True)
if not sensors:
    print("no readable hwmon sensors: nothing measured")
    sys.exit(0)

ctx = gsum_amd.lab_context(0)
n = 8192
X = 0.1 * np.arange(n)[:, None]
Z = np.concatenate([np.random.RandomState(0).randn(n, 6), np.ones((n, 1))], axis=1)
desc = gsum_amd.describe_kernel(RBF(0.2), 1)
ctx.set_inputs(X, Z)
ctx.set_option("batch_slots", 20)
ctx.lml_resident([desc] * 20, 1e-10)


def phase(name, work):
    stop = threading.Event()
    rows = []

    def sample():
        while not stop.is_set():
            rows.append([read(s.get("power", "")) for s in sensors.values()] + [read(s.get("sclk", "")) for s in sensors.values()]
                        + [read(s.get("mem_busy_percent", "")) for s in sensors.values()] + [read(s.get("gpu_busy_percent", "")) for s in sensors.values()])
            time.sleep(0.01)

    th = threading.Thread(target=sample, daemon=True)
    th.start()
    t0 = time.perf_counter()
    rate = work()
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    a = np.array(rows[len(rows) // 5:])            # drop the ramp
    k = len(sensors)
    busiest = int(np.nanargmax(np.nanmean(a[:, :k], axis=0)))      # the card that is working (several may be visible)
    card = list(sensors)[busiest]
    cap = read(sensors[card].get("cap", "")) / 1e6
    print(f"{name:34s} {dt:5.2f} s  {rate}  power {np.nanmean(a[:, busiest]) / 1e6:7.1f} W (max {np.nanmax(a[:, busiest]) / 1e6:7.1f}, cap {cap:.0f})  "
          f"sclk {np.nanmean(a[:, k + busiest]) / 1e6:6.0f} MHz (min {np.nanmin(a[:, k + busiest]) / 1e6:.0f})  "
          f"mem_busy {np.nanmean(a[:, 2 * k + busiest]):5.1f} %  gpu_busy {np.nanmean(a[:, 3 * k + busiest]):5.1f} %  [{len(a)} samples]", flush=True)


def gemm(K, reps, fill=0):
    def run():
        ctx.set_option("bench_fill", fill)
        tf = [ctx.bench_gemm_nt(7, 7936, 7936, K, tri=True, lda=8208, reps=reps)[0] for _ in range(8)]
        ctx.set_option("bench_fill", 0)
        return f"{np.median(tf):5.1f} TF/s"
    return run


def batch():
    t0 = time.perf_counter()
    ctx.set_option("batch_slots", 20)
    for _ in range(8):
        ctx.lml_resident([desc] * 80, 1e-10)
    return f"{640 / (time.perf_counter() - t0):5.1f} evals/s"


def single():
    ctx.set_option("batch_slots", 1)
    t0 = time.perf_counter()
    for _ in range(350):
        ctx.lml_resident([desc], 1e-10)
    return f"{(time.perf_counter() - t0) / 350 * 1e3:5.2f} ms each"


def medium(nm, count):
    def run():
        Xm = 0.1 * np.arange(nm)[:, None]
        Zm = np.concatenate([np.random.RandomState(0).randn(nm, 4), np.ones((nm, 1))], axis=1)
        ctx.set_inputs(Xm, Zm)
        darr = ctx.desc_array([gsum_amd.describe_kernel(RBF(float(e)), 1) for e in np.linspace(0.15, 0.25, count)])
        ctx.lml_resident(darr, 1e-10)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 2.0:
            ctx.lml_resident(darr, 1e-10)
            reps += 1
        rate = reps * count / (time.perf_counter() - t0)
        ctx.set_inputs(X, Z)
        return f"{rate:8.0f} evals/s ({nm ** 3 / 3.0 * rate / 1e12:4.1f} TF/s)"
    return run


def idle():
    time.sleep(1.5)
    return "idle"


for rnd in range(2):
    phase("idle", idle)
    phase("bulk tile K=256 back to back", gemm(256, 1000))
    phase("bulk tile K=512 back to back", gemm(512, 500))
    phase("bulk tile K=2048 back to back", gemm(2048, 125))
    phase("bulk tile K=256, ALL-ZERO data", gemm(256, 1000, 1))
    phase("bulk tile K=512, ALL-ZERO data", gemm(512, 500, 1))
    phase("batch, 20 in flight", batch)
    phase("one factorisation at a time", single)
    phase("fused path, n = 2048, 1024 evals", medium(2048, 1024))
    phase("fused path, n = 4096, 512 evals", medium(4096, 512))
    phase("fused path, n = 1024, 2048 evals", medium(1024, 2048))
