#!/usr/bin/env python3
"""timeline_summary.py FILE [slots] — per-workgroup wall-clock timeline of the fused table kernel (DEV build,
VET_LUT_TIMELINE=FILE): when the workgroups of each dispatch wave start and what their stages take."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 6)
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
ok = a[:, 4] > 0
t = a[:, :5].astype(np.float64) * 0.01           # 100 MHz wall clock -> us
t0 = t[ok, 0].min()
t -= t0
print(f"{ok.sum()} workgroups, kernel span {t[ok, 4].max():.1f} us (first entry -> last exit)")
names = ["samples->set", "lists", "walk", "entropy"]
n = len(a)
for lo in range(0, n, slots):
    hi = min(n, lo + slots)
    s = slice(lo, hi)
    d = np.diff(t[s], axis=1)
    print(f"workgroups [{lo:5d},{hi:5d}): start {t[s, 0].mean():7.1f} us (min {t[s, 0].min():6.1f}, max {t[s, 0].max():6.1f})  end {t[s, 4].mean():7.1f} (max {t[s, 4].max():6.1f})"
          f"  lifetime {(t[s, 4] - t[s, 0]).mean():6.1f} us = " + " + ".join(f"{names[i]} {d[:, i].mean():5.1f}" for i in range(4)))
# concurrency over time: workgroups resident per 5 us bucket
edges = np.arange(0, t[ok, 4].max() + 5, 5.0)
res = [(int(((t[ok, 0] < e + 5) & (t[ok, 4] > e)).sum())) for e in edges[:-1]]
print("resident workgroups per 5 us:", res)
hw = a[:, 5]
cu = ((hw >> 8) & 0xF).astype(int); se = ((hw >> 13) & 0x7).astype(int); xcc = ((hw >> 32) & 0xF).astype(int)
print("first 16 workgroups: xcc", xcc[:16].tolist(), "se", se[:16].tolist(), "cu", cu[:16].tolist())
# per-XCD and per-CU view: how many workgroups each ran, how long they lived there, when the last one ended
life = t[:, 4] - t[:, 0]
key = xcc * 64 + se * 16 + cu
print("per XCD: workgroups, mean lifetime, last end (us)")
for x in np.unique(xcc[ok]):
    m = ok & (xcc == x)
    print(f"  xcc {x}: {m.sum():5d}  {life[m].mean():6.1f}  {t[m, 4].max():6.1f}   CUs seen {len(np.unique(key[m]))}")
per_cu = np.array([[k, (key == k).sum(), life[key == k].mean(), t[key == k, 4].max()] for k in np.unique(key[ok])])
print(f"per CU ({len(per_cu)} seen): workgroups min/mean/max {per_cu[:, 1].min():.0f}/{per_cu[:, 1].mean():.1f}/{per_cu[:, 1].max():.0f};"
      f" mean lifetime min/max {per_cu[:, 2].min():.1f}/{per_cu[:, 2].max():.1f} us; last end min/max {per_cu[:, 3].min():.1f}/{per_cu[:, 3].max():.1f} us")
q = np.percentile(life[ok], [0, 10, 50, 90, 100])
print("lifetime percentiles 0/10/50/90/100:", np.round(q, 1).tolist())
# lifetime against the frame index (early frames of a random walk have fewer distinct directions)
nb = len(a)
for lo in range(0, nb, max(nb // 10, 1)):
    hi = min(nb, lo + max(nb // 10, 1))
    print(f"  workgroups [{lo:5d},{hi:5d}): lifetime {life[lo:hi].mean():6.1f} us, walk {(t[lo:hi, 3] - t[lo:hi, 2]).mean():6.1f}")
