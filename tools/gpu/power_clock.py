#!/usr/bin/env python
"""GPU-box tool (round-5 verdict item 3): is a product kernel power-bound?  Telemetry, not inference.

Each product kernel -- warp forward / backward-data / weight gradients, field forward / fused backward, hash-grid backward -- is
launched ALONE in a loop for --seconds (cfg3's call: 2 097 152 points, random weights) while a sampler thread reads, every 10 ms,
  * socket power  (hwmon power1_input, microwatts: the number amd-smi / rocm-smi print as "socket power")
  * shader clock  (hwmon freq1_input, Hz: sclk)
and, every 50 ms through the amdsmi library when it imports, the gpu_metrics record (average_socket_power, the per-XCD gfx clocks,
gfx activity).  Beside them the bare streams of tools/micro/mfma_bf16_rate (built on the box): the matrix pipe with nothing else.
Printed per kernel: ms per launch (HIP events over the loop), mean / p95 socket power, mean sclk, the board's power cap.
The effective clock GRBM_GUI_ACTIVE / duration per kernel comes from a rocprofv3 --pmc pass over `--once` (tools/gpu/r6_power.sh).

    python tools/gpu/power_clock.py [--seconds 2.0] [--once] > profiles/r06_power_clock.txt
"""
import argparse
import ctypes
import glob
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from morpheus_amd import _lib, ops, packing  # noqa: E402
from morpheus_amd.ops import ptr, stream  # noqa: E402


def _hwmon():
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if os.path.exists(os.path.join(d, "power1_input")):
            return d
    return None


class Sampler(threading.Thread):
    def __init__(self, period=0.01):
        super().__init__(daemon=True)
        self.period, self.hw = period, _hwmon()
        self.rows, self.metrics, self.on, self.stop_flag = [], [], False, False
        self.smi = None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            self.smi = (amdsmi, amdsmi.amdsmi_get_processor_handles()[0])
        except Exception as e:      # noqa: BLE001
            print("# amdsmi library not usable:", type(e).__name__, str(e)[:80])

    def _read(self, name):
        try:
            return float(open(os.path.join(self.hw, name)).read())
        except Exception:      # noqa: BLE001
            return float("nan")

    def run(self):
        k = 0
        while not self.stop_flag:
            if self.on and self.hw:
                self.rows.append((time.perf_counter(), self._read("power1_input") * 1e-6, self._read("freq1_input") * 1e-6))
                if self.smi and k % 5 == 0:
                    try:
                        m = self.smi[0].amdsmi_get_gpu_metrics_info(self.smi[1])
                        clk = [c for c in m.get("current_gfxclks", []) if isinstance(c, (int, float)) and 0 < c < 10000]
                        self.metrics.append((m.get("average_socket_power"), sum(clk) / max(len(clk), 1) if clk else None,
                                             m.get("average_gfx_activity")))
                    except Exception:      # noqa: BLE001
                        pass
                k += 1
            time.sleep(self.period)

    def window(self):
        self.rows, self.metrics, self.on = [], [], True

    def close(self):
        self.on = False
        r = self.rows
        if not r:
            return dict(n=0)
        p = sorted(x[1] for x in r)
        f = [x[2] for x in r]
        out = dict(n=len(r), power_mean=sum(p) / len(p), power_p95=p[int(0.95 * (len(p) - 1))], power_max=p[-1], sclk_mean=sum(f) / len(f),
                   sclk_min=min(f), sclk_max=max(f))
        def num(v):
            try:
                return float(v)
            except (TypeError, ValueError):
                return None
        mm = [m for m in self.metrics if num(m[0]) is not None]
        if mm:
            out["smi_power_mean"] = sum(num(m[0]) for m in mm) / len(mm)
            cl = [m[1] for m in self.metrics if m[1]]
            out["smi_gfxclk_mean"] = sum(cl) / len(cl) if cl else None
            ac = [num(m[2]) for m in self.metrics if num(m[2]) is not None]
            out["smi_gfx_activity"] = sum(ac) / len(ac) if ac else None
        return out


def build_cases(M, dev):
    lib = _lib.load()
    torch.manual_seed(0)
    ops.set_mlp_mode("b3")
    cases = {}
    # ---- warp nets
    nets = []
    for nout in (3, 2):
        W = [torch.randn(128, 39, device=dev) * 0.15] + [torch.randn(128, 128, device=dev) * 0.1 for _ in range(4)] + \
            [torch.randn(nout, 128, device=dev) * 0.15]
        b = [torch.randn(128, device=dev) * 0.1 for _ in range(5)] + [torch.randn(nout, device=dev) * 0.1]
        nets.append([p.requires_grad_() for p in W + b])
    x = (torch.rand(M, 3, device=dev) * 2 - 1).contiguous()
    b0 = [torch.randn(1, 128, device=dev) * 0.3 for _ in range(2)]
    wop = ops.prepare_warp_operands(nets[0], nets[1])
    acts = torch.empty(lib.mh_warp_acts_floats(M), device=dev)
    dpre = torch.empty(lib.mh_warp_dpre_floats(M), device=dev)
    deform, topo = torch.empty(M, 3, device=dev), torch.empty(M, 2, device=dev)
    g_d, g_t, g_x = torch.randn(M, 3, device=dev), torch.randn(M, 2, device=dev), torch.empty(M, 3, device=dev)
    (bd, bt) = wop.b
    n_tiles = lib.mh_mlp_tiles(M)

    def warp_fwd():
        ops.check(lib.mh_warp_fwd_b3(ptr(x), None, ptr(b0[0]), ptr(b0[1]), ptr(wop.w3[0]), ptr(wop.w3[1]), ptr(bd), ptr(bt), 6, ptr(deform),
                                     ptr(topo), ptr(acts), M, stream()), "fwd")

    def warp_bwd():
        ops.check(lib.mh_warp_bwd_data_b3(ptr(x), ptr(g_d), ptr(g_t), ptr(wop.wT3[0]), ptr(wop.wT3[1]), 6, ptr(acts), ptr(dpre), ptr(g_x), M, 0,
                                          stream()), "bwd")

    def warp_wgrad():
        ops._wgrad(lib, acts, dpre, ops.WARP_ACT_ROWS * 32, ops.WARP_DPRE_ROWS * 32, ops._WARP_WG[0], ops._WARP_WG[1], ops._WARP_WG[2],
                   ops._WARP_WG[3], n_tiles, dev, "warp", b3=True)

    cases["mh_warp_fwd_b3"] = warp_fwd
    cases["mh_warp_bwd_data_b3"] = warp_bwd
    cases["mh_mlp_wgrad_b3[warp]"] = warp_wgrad
    # ---- field nets
    Ws = [torch.randn(64, 73, device=dev) * 0.2, torch.randn(64, 64, device=dev) * 0.2, torch.randn(33, 64, device=dev) * 0.2]
    Wc = [torch.randn(64, 64, device=dev) * 0.2, torch.randn(64, 64, device=dev) * 0.2, torch.randn(3, 64, device=dev) * 0.2]
    bs = [torch.randn(64, device=dev) * 0.1, torch.randn(64, device=dev) * 0.1, torch.randn(33, device=dev) * 0.1]
    bc = [torch.randn(64, device=dev) * 0.1, torch.randn(64, device=dev) * 0.1, torch.randn(3, device=dev) * 0.1]
    fop = ops.prepare_field_operands([p.requires_grad_() for p in Ws + Wc + bs + bc])
    fs, fc = torch.randn(M, 32, device=dev) * 0.1, torch.randn(M, 32, device=dev) * 0.1
    tp = torch.randn(M, 2, device=dev) * 0.1
    beta = torch.tensor([0.1], device=dev)
    state = {}

    def field_fwd():
        state["f"] = ops._field_fwd(lib, x, fs, fc, tp, beta, 6, True, fop, True)

    g_sdf, g_sig, g_alb = torch.randn(M, device=dev), torch.randn(M, device=dev) * 0.01, torch.randn(M, 3, device=dev)
    wT, b3 = ops._field_wT(fop, True)

    def field_bwd():
        sdf, sigma, albedo, facts = state["f"]
        ops._field_bwd(lib, x, wT, beta, facts, sdf, albedo, g_sdf, g_sig, g_alb, 6, True, True, True, True, fop.jp, b3=b3)

    cases["mh_field_fwd_b3"] = field_fwd
    cases["mh_field_bwd_fused_b3"] = field_bwd
    # ---- hash grid backward (one table, d/dx)
    from morpheus_amd import synth
    from morpheus_amd.ops import level_resolutions
    offs, s = synth.grid_offsets()
    res = level_resolutions(16, s, 16)
    emb = (torch.rand(int(offs[-1]), 2, device=dev) * 2 - 1) * 0.1
    import numpy as np
    o_np, r_np = np.ascontiguousarray(offs, dtype=np.int32), np.ascontiguousarray(res, dtype=np.int32)
    o_p, r_p = o_np.ctypes.data_as(ctypes.c_void_p), r_np.ctypes.data_as(ctypes.c_void_p)
    gfeat = torch.randn(M, 32, device=dev) * 0.01
    xg = (torch.rand(M, 3, device=dev) * 2 - 1).contiguous()
    binned = ops._bin_points(lib, xg, 1.01)

    def grid_bwd():
        ops._grid_bwd(lib, xg, [emb], [gfeat], o_p, r_p, 16, 16, 1.01, True, binned=binned)

    def grid_fwd():
        ops._grid_fwd(lib, xg, [emb], o_p, r_p, 16, 16, 1.01, 1)

    cases["mh_grid_encode_bwd_binned"] = grid_bwd
    cases["mh_grid_encode_fwd_binned"] = grid_fwd
    state["keep"] = (o_np, r_np)
    field_fwd()
    warp_fwd()
    warp_bwd()
    torch.cuda.synchronize()
    return cases


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--points", type=int, default=128 * 128 * 128)
    ap.add_argument("--once", action="store_true", help="three launches of every kernel and nothing else (the rocprofv3 --pmc pass)")
    ap.add_argument("--micro", default=None, help="path of a built tools/micro binary to run under the sampler as well")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cases = build_cases(a.points, dev)
    if a.once:
        for name, fn in cases.items():
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        return
    smp = Sampler()
    smp.start()
    hw = smp.hw
    cap = float(open(os.path.join(hw, "power1_cap")).read()) * 1e-6 if hw and os.path.exists(os.path.join(hw, "power1_cap")) else float("nan")
    print(f"# device {torch.cuda.get_device_name(dev)}; hwmon {hw}; power cap {cap:.0f} W; loop {a.seconds} s per kernel, sampled every 10 ms")
    smp.window()
    time.sleep(1.0)
    idle = smp.close()
    print(f"idle                          power {idle.get('power_mean', float('nan')):6.1f} W   sclk {idle.get('sclk_mean', float('nan')):6.0f} MHz")
    print(f"{'kernel':30s} {'ms/launch':>9s} {'power W':>8s} {'p95 W':>7s} {'of cap':>7s} {'sclk MHz':>9s} {'min':>6s} {'smi W':>7s} {'smi clk':>8s} {'act %':>6s}")
    for name, fn in cases.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 0
        t_end = time.perf_counter() + 0.3           # settle the clock first
        while time.perf_counter() < t_end:
            fn()
        torch.cuda.synchronize()
        smp.window()
        e0.record()
        t_end = time.perf_counter() + a.seconds
        while time.perf_counter() < t_end:
            for _ in range(8):
                fn()
            n += 8
            torch.cuda.synchronize()                # keep the queue short: the window is the kernel, not a backlog
        e1.record()
        torch.cuda.synchronize()
        w = smp.close()
        ms = e0.elapsed_time(e1) / max(n, 1)
        f = lambda k, fmt: (fmt % w[k]) if w.get(k) is not None else "   n/a"
        print(f"{name:30s} {ms:9.3f} {w.get('power_mean', float('nan')):8.1f} {w.get('power_p95', float('nan')):7.1f} "
              f"{w.get('power_mean', float('nan')) / cap:7.2f} {w.get('sclk_mean', float('nan')):9.0f} {w.get('sclk_min', float('nan')):6.0f} "
              f"{f('smi_power_mean', '%7.1f')} {f('smi_gfxclk_mean', '%8.0f')} {f('smi_gfx_activity', '%6.0f')}", flush=True)
    if a.micro and os.path.exists(a.micro):
        for occ, zero in ((1, 0), (2, 0), (1, 1)):
            time.sleep(0.5)
            smp.window()
            out = subprocess.run([a.micro, str(a.seconds), str(occ), str(zero)], capture_output=True, text=True, timeout=120)
            w = smp.close()
            # the window includes the micro's start-up (hipMalloc, first launch): p95 / max are the loop's level
            print(f"{'bare bf16 MFMA stream occ%d %s' % (occ, 'zero' if zero else 'live'):30s} {'':>9s} {w.get('power_mean', 0):8.1f} {w.get('power_p95', 0):7.1f} "
                  f"{w.get('power_p95', 0) / cap:7.2f} {w.get('sclk_mean', 0):9.0f} {w.get('sclk_min', 0):6.0f}   (of cap: p95; window includes process start-up)")
            for ln in out.stdout.strip().splitlines()[-1:]:
                print("#   " + ln)
    smp.stop_flag = True


if __name__ == "__main__":
    main()
