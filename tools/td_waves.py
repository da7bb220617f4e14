#!/usr/bin/env python3
"""Where and when the waves of one k_fused4_td launch ran (dump of a -DSPECINV_TD_STAMPS=1 build, SPECINV_TD_STAMP_DUMP=file):
duration against XCC / SE / CU / SIMD, start spread, how many waves a SIMD held at a time."""
import sys, collections
import numpy as np
rows = [tuple(int(v) for v in ln.split()) for ln in open(sys.argv[1])]
for it in sorted({r[0] for r in rows}):
    r = np.array([x for x in rows if x[0] == it], dtype=np.int64)
    wave, xcc, hw, beg, end = r[:, 1], r[:, 2] & 0xf, r[:, 3], r[:, 4], r[:, 5]
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
    dur = end - beg
    print(f"iteration {it}: {len(r)} waves; begin spread {beg.max()} ticks, end min {end.min()} max {end.max()}; duration mean {dur.mean():.0f} "
          f"min {dur.min()} max {dur.max()}")
    key = xcc * 100000 + se * 10000 + sh * 1000 + cu * 10 + simd
    per = collections.defaultdict(list)
    for k, b, e, d in zip(key, beg, end, dur):
        per[k].append((b, e, d))
    occ = collections.Counter(len(v) for v in per.values())
    print(f"  distinct (xcc, se, sh, cu, simd): {len(per)}; waves per SIMD over the launch: {dict(sorted(occ.items()))}")
    for n in sorted(occ):
        ds = [d for v in per.values() if len(v) == n for (_, _, d) in v]
        print(f"    SIMDs that ran {n} wave(s): mean duration {np.mean(ds):.0f} (min {np.min(ds)}, max {np.max(ds)})")
    for name, arr in (("xcc", xcc), ("se", se), ("cu", cu), ("simd", simd)):
        print("  by", name, {int(v): int(dur[arr == v].mean()) for v in np.unique(arr)})
    late = beg > 0.05 * end.max()
    print(f"  waves that began after 5 % of the launch: {int(late.sum())}; their mean duration {dur[late].mean() if late.any() else 0:.0f}")
