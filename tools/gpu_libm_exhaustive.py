#!/usr/bin/env python3
"""The device instance of yuki_amd/csrc/yk_libm.h against oracle/olibm.h for ALL 2^32 binary32 arguments of
sinf, cosf, tanf, logf, acosf (and atan2f: every argument as y against x = rotated copy, and as x against the same),
plus the oracle against this host's platform libm on the same sweep (the `hostlibm` build of the oracle), and the device's f32
sqrt and division against IEEE (numpy on the host) on the same arguments.
GPU box:  python tools/gpu_libm_exhaustive.py [--stride 1] [--out gpurun_out/device_libm_exhaustive.txt]
"""
import argparse
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yuki_amd import core as yk  # noqa: E402
from oracle import binding as oracle  # noqa: E402

NAMES = ["sinf", "cosf", "tanf", "logf", "acosf", "atan2f", "expf"]  # expf: host side only (the pbrt loader), oracle vs platform libm


def same(a, b):
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


def oracle_parallel(pool, workers, fn, x, y):
    parts = np.array_split(np.arange(x.size), workers)
    out = np.empty_like(x)

    def job(idx):
        out[idx[0]:idx[-1] + 1] = oracle.libm_array(fn, x[idx[0]:idx[-1] + 1], None if y is None else y[idx[0]:idx[-1] + 1])

    list(pool.map(job, [p for p in parts if p.size]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--chunk", type=int, default=1 << 25)
    ap.add_argument("--workers", type=int, default=min(16, os.cpu_count() or 4))
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    ctx = yk.Context(0)
    pool = ThreadPoolExecutor(a.workers)
    dev_bad = [0] * 7
    host_bad = [0] * 7
    count = [0] * 7
    sqrt_bad = rcp_bad = pair_bad = 0
    first = {}
    t0 = time.time()
    total = (1 << 32) // a.stride
    for start in range(0, total, a.chunk):
        n = min(a.chunk, total - start)
        u = (np.arange(start, start + n, dtype=np.uint64) * a.stride).astype(np.uint32)
        x = u.view(np.float32)
        partner = ((u * np.uint32(2654435761)) ^ np.uint32(0x9E3779B9)).view(np.float32)  # atan2f's other argument: a bijective scramble
        for fn in range(7):
            y = partner if fn == 5 else None
            want = oracle_parallel(pool, a.workers, fn, x, y)
            got = yk.device_math(ctx, fn, x, y) if fn < 6 else want
            ok = same(got, want)
            dev_bad[fn] += int((~ok).sum())
            if not ok.all() and fn not in first:
                i = int(np.argmin(ok))
                first[fn] = (hex(int(u[i])), hex(int(got.view(np.uint32)[i])), hex(int(want.view(np.uint32)[i])))
            with oracle.flavour("hostlibm"):
                host = oracle_parallel(pool, a.workers, fn, x, y)
            host_bad[fn] += int((~same(host, want)).sum())
            count[fn] += n
        # det_sincosf, the shared-reduction pair of the shading code, against the oracle's sinf / cosf
        pair_bad += int((~same(yk.device_math(ctx, 28, x), oracle_parallel(pool, a.workers, 0, x, None))).sum())
        pair_bad += int((~same(yk.device_math(ctx, 29, x), oracle_parallel(pool, a.workers, 1, x, None))).sum())
        # the two IEEE operations everything else leans on: f32 sqrt and division on gfx950 against the host's (correctly rounded)
        with np.errstate(all="ignore"):
            sqrt_bad += int((~same(yk.device_math(ctx, 6, x), np.sqrt(x))).sum())
            rcp_bad += int((~same(yk.device_math(ctx, 7, partner, x), partner / x)).sum())
        print(f"  {start + n:>11d} / {total} arguments, {time.time() - t0:6.0f} s, device differs {sum(dev_bad)}, host libm differs {sum(host_bad)}", flush=True)
    lines = [f"# device yk_libm.h vs oracle/olibm.h vs this host's libm ({os.confstr('CS_GNU_LIBC_VERSION')}), stride {a.stride}; NaNs as a class",
             f"# {time.time() - t0:.0f} s with {a.workers} host threads"]
    for fn in range(7):
        dev = f"device != oracle on {dev_bad[fn]}" if fn < 6 else "(no device instance in use)"
        lines.append(f"{NAMES[fn]:7s} {count[fn]:>11d} arguments: {dev}, oracle != platform libm on {host_bad[fn]}"
                     + (f"  first {first[fn]}" if fn in first else ""))
    lines.append(f"sincos pair (one reduction, both results) {count[0]:>11d} arguments: device != oracle's sinf / cosf on {pair_bad}")
    lines.append(f"sqrt    {count[0]:>11d} arguments: device f32 sqrt != IEEE on {sqrt_bad}")
    lines.append(f"div     {count[0]:>11d} pairs (scrambled partner / argument): device f32 division != IEEE on {rcp_bad}")
    print("\n".join(lines))
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")
    return 1 if sum(dev_bad) or sum(host_bad) or sqrt_bad or rcp_bad or pair_bad else 0


if __name__ == "__main__":
    sys.exit(main())
