#!/usr/bin/env python3
"""
Streaming-row experiments, all in ONE process on ONE box, alternating so that only differences are read:

  (a) output bases SKEWED: one slab, pos at 0, vel at size + s, act at 2 size + 2 s for byte skews s -- the three write
      streams of a wave otherwise sit at the same offset of three arrays 700 MiB (= 175 x 4 MiB) apart, i.e. on the same
      channel phase of the HBM interleave;
  (c) trajectory only (2 output arrays) vs trajectory + actions (3 arrays);
  (t) a sysfs timeline (sclk / power / mem busy, every ~5 ms) while the streaming kernel runs and while a plain fill of
      the same arrays runs.

    python tools/skew_probe.py [B] [rounds]
"""
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fancy_gym_amd import RolloutSpec, TrajectoryEngine  # noqa: E402


def timed(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / n


def timeline(fn, seconds=0.6):
    """run fn back to back for `seconds`, sampling sysfs from a thread meanwhile"""
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            s = bench.gpu_state_sysfs(0)
            samples.append((s.get("sclk_mhz"), s.get("package_power_w"), s.get("gpu_busy_percent"), s.get("mem_busy_percent"),
                            s.get("mclk_mhz"), s.get("fclk_mhz")))
            time.sleep(0.005)
    th = threading.Thread(target=sampler)
    t0 = time.perf_counter()
    th.start()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        n += 5
    el = time.perf_counter() - t0
    stop.set(); th.join()
    tail = samples[len(samples) // 2:]
    col = lambda i: [s[i] for s in tail if s[i] is not None]  # noqa: E731
    avg = lambda v: (sum(v) / len(v)) if v else float("nan")  # noqa: E731
    return el / n, dict(sclk=avg(col(0)), sclk_min=min(col(0) or [0]), sclk_max=max(col(0) or [0]), power=avg(col(1)),
                        gpu_busy=avg(col(2)), mem_busy=avg(col(3)), mclk=avg(col(4)), fclk=avg(col(5)), n=len(tail))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    eng = TrajectoryEngine("prodmp", "exp", "prodmp", device=0, **bench.CFG)
    spec = RolloutSpec("motor", 7, bench.P_GAINS, bench.D_GAINS, -1.0, 1.0, plant="static")
    params, ip, iv = (t.to(dev) for t in bench.synth_inputs(B, 1))
    cp, cv = ip.double().contiguous(), iv.double().contiguous()
    T, D = eng.num_steps, eng.num_dof
    n_el = B * T * D
    size = n_el * 4
    nb3 = B * (eng.num_params * 4 + 2 * D * 4 + 3 * T * D * 4)
    nb2 = B * (eng.num_params * 4 + 2 * D * 4 + 2 * T * D * 4)
    sep = tuple(torch.empty((B, T, D), device=dev) for _ in range(3))
    skews = [0, 256, 1024, 4096, 3 * 4096, 17 * 4096, 257 * 4096, (1 << 21) + 4096, 64, 4096 + 256]
    slab = torch.empty(3 * n_el + 2 * max(skews) // 4 * 2 + 1024, device=dev)

    def views(s):
        e = s // 4
        return tuple(slab[i * (n_el + e): i * (n_el + e) + n_el].view(B, T, D) for i in range(3))

    def run3(out):
        return lambda: eng.trajectory_actions(params, ip, iv, spec, cp, cv, out=out)

    def run2(out):
        return lambda: eng.trajectory(params, ip, iv, 0.0, out=out[:2])
    print(f"B = {B}; array size {size / 2**20:.1f} MiB; data_ptr of three allocations: "
          + ", ".join(hex(t.data_ptr()) for t in sep) + f"; slab {hex(slab.data_ptr())}")
    from fancy_gym_amd import _lib as _l
    _l.set_option("flat", 0)    # the skew sweep characterises k_traj_stream (round 2's kernel)
    timed(run3(sep), n=80)      # clocks
    res = {}
    for r in range(rounds):
        order = [("sep", None)] + [("skew", s) for s in skews]
        if r % 2:
            order.reverse()
        for kind, s in order:
            out = sep if kind == "sep" else views(s)
            t3 = timed(run3(out))
            t2 = timed(run2(out))
            res.setdefault((kind, s), []).append((t3, t2))
    fill = timed(lambda: [o.fill_(1.0) for o in sep], n=10)
    print(f"fill of the three arrays: {3 * size / fill / 1e9:.0f} GB/s")
    print("| outputs | skew (bytes) | fused +actions us (per round) | GB/s alg. | of 8 TB/s | trajectory only us | GB/s alg. | of 8 TB/s |")
    print("|---|---|---|---|---|---|---|---|")
    for (kind, s), v in res.items():
        t3 = min(x[0] for x in v); t2 = min(x[1] for x in v)
        print(f"| {'three allocations' if kind == 'sep' else 'one slab'} | {'-' if s is None else s} | "
              f"{' / '.join(f'{x[0] * 1e6:.0f}' for x in v)} | {nb3 / t3 / 1e9:.0f} | {nb3 / t3 / 8e12 * 100:.1f} % | "
              f"{' / '.join(f'{x[1] * 1e6:.0f}' for x in v)} | {nb2 / t2 / 1e9:.0f} | {nb2 / t2 / 8e12 * 100:.1f} % |")
    # occupancy: EXTRA dynamic LDS per workgroup (mpk_set_option "lds_pad", KiB) -- 0: three 4-wave workgroups per CU
    # (45.5 KB each), 10: two, i.e. 12 -> 8 waves per CU
    from fancy_gym_amd import _lib
    print()
    print("| kernel | lds_pad KiB | workgroups per CU | fused +actions us | trajectory only us |")
    print("|---|---|---|---|---|")
    for flat, pad, wg in ((0, 0, 3), (0, 10, 2), (1, 0, 2), (0, 0, 3), (0, 10, 2), (1, 0, 2), (0, 40, 1)):
        _lib.reset_options()
        _lib.set_option("flat", flat)
        if pad:
            _lib.set_option("lds_pad", pad)
        t3, k3 = timed(run3(sep)), eng.last_kernel()
        t2 = timed(run2(sep))
        print(f"| `{k3}` | {pad} | {wg} | {t3 * 1e6:.0f} | {t2 * 1e6:.0f} |")
    _lib.reset_options()
    print()
    print("| what runs | s per launch | sclk avg (min-max) MHz | power W | gpu busy % | mem busy % | mclk | fclk | samples |")
    print("|---|---|---|---|---|---|---|---|---|")
    for name, fn in (("streaming kernel (+actions)", run3(sep)), ("fill of the three arrays", lambda: [o.fill_(1.0) for o in sep]),
                     ("trajectory only", run2(sep)), ("streaming kernel (+actions) again", run3(sep))):
        t, st = timeline(fn)
        print(f"| {name} | {t * 1e6:.0f} us | {st['sclk']:.0f} ({st['sclk_min']}-{st['sclk_max']}) | {st['power']:.0f} | "
              f"{st['gpu_busy']:.0f} | {st['mem_busy']:.0f} | {st['mclk']:.0f} | {st['fclk']:.0f} | {st['n']} |")


if __name__ == "__main__":
    main()
