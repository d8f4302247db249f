"""Reads the raw per-block phase stamps of bottleneck_tail_kernel (FAV_CONV_DBG=1 FAV_CONV_DBG_DUMP=file, 100 MHz clock)
and prints how many blocks sit in the conv_b phase (no HBM traffic) and in the expand phase (HBM-bound) at the same time:
if the CUs walk through the phases in lock step, the chip alternates between an MFMA-only and an HBM-only state."""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64)
pos = 0
while pos < raw.size:
    nb, tag = int(raw[pos]), int(raw[pos + 1])
    t = raw[pos + 2: pos + 2 + nb * 16].reshape(nb, 16).astype(np.int64)
    pos += 2 + nb * 16
    t0 = t[:, 0].min()
    start, p1s, p1e, p2s, p2e, end = [(t[:, i] - t0) / 100.0 for i in range(6)]      # us
    span = end.max()
    grid = np.arange(0.0, span, 0.5)
    in_p1 = ((grid[:, None] >= p1s[None, :]) & (grid[:, None] < p1e[None, :])).sum(1)
    in_p2 = ((grid[:, None] >= p2s[None, :]) & (grid[:, None] < p2e[None, :])).sum(1)
    live = ((grid[:, None] >= start[None, :]) & (grid[:, None] < end[None, :])).sum(1)
    mid = (grid > 0.1 * span) & (grid < 0.9 * span)
    f2 = in_p2[mid] / np.maximum(live[mid], 1)
    print(f"launch Cmid/Nred tag {tag}: {nb} blocks, span {span:.0f} us; per block P1 {np.mean(p1e - p1s):.1f} us, P2 {np.mean(p2e - p2s):.1f} us")
    print(f"  share of live blocks in P2, middle 80 % of the launch: mean {f2.mean():.2f}, std {f2.std():.2f}, "
          f"p5 {np.percentile(f2, 5):.2f}, p95 {np.percentile(f2, 95):.2f}  (lock step: p5 ~ 0, p95 ~ 1; spread out: std ~ 0)")
    # coarse strip chart: one character per 1/100 of the span
    cols = 100
    chart = ""
    for c in range(cols):
        sel = (grid >= span * c / cols) & (grid < span * (c + 1) / cols)
        v = in_p2[sel].sum() / max(1, live[sel].sum())
        chart += " .:-=+*#%@"[min(9, int(v * 10))]
    print("  P2 share over time: |" + chart + "|")
