"""Offline analysis of gpurun_out/mfma_probe.npz: which summation model reproduces
the bf16 MFMA bit for bit?"""
import numpy as np, itertools, sys
d = np.load("gpurun_out/mfma_probe.npz")
f32 = np.float32

def tof(b):  # uint16 bf16 bits -> float32
    return (b.astype(np.uint32) << 16).view(np.float32)

def model_seq(A, B, Cm, order):
    """acc = fl32(acc + a_k*b_k) in the given k order (products exact in fp32)."""
    acc = Cm.copy()
    for k in order:
        acc = (acc + (A[:, :, None, k] * B[:, None, :, k]).astype(f32)).astype(f32)
    return acc

def model_grouped(A, B, Cm, groups, inner="exact"):
    """acc = fl32(acc + S_g) with S_g the sum of the group's products: exact (float64) or fp32 sequential."""
    acc = Cm.copy()
    for g in groups:
        if inner == "exact":
            s = np.zeros(acc.shape, np.float64)
            for k in g:
                s += A[:, :, None, k].astype(np.float64) * B[:, None, :, k].astype(np.float64)
            acc = (acc.astype(np.float64) + s).astype(f32)
        elif inner == "f32seq":
            s = np.zeros(acc.shape, f32)
            for k in g:
                s = (s + A[:, :, None, k] * B[:, None, :, k]).astype(f32)
            acc = (acc + s).astype(f32)
    return acc

for tag, K in (("16", 32), ("32", 16)):
    print("==== mfma", "16x16x32" if tag == "16" else "32x32x16")
    for name in ("normal", "wide", "pos", "bigc"):
        A, B, Cm, D = tof(d[f"{name}_{tag}_A"]), tof(d[f"{name}_{tag}_B"]), d[f"{name}_{tag}_C"], d[f"{name}_{tag}_D"]
        A, B, Cm, D = A[:96], B[:96], Cm[:96], D[:96]
        # D[p][m][n] = sum_k A[p][m][k] * Bt[p][n][k] + C[p][m][n]
        res = {}
        res["seq k"] = model_seq(A, B, Cm, range(K))
        res["seq rev"] = model_seq(A, B, Cm, range(K - 1, -1, -1))
        res["all exact"] = model_grouped(A, B, Cm, [list(range(K))])
        for G in (2, 4, 8, 16):
            res[f"groups of {G} exact, seq"] = model_grouped(A, B, Cm, [list(range(i, i + G)) for i in range(0, K, G)])
        # lane-group interleave: j-major (k = 8g + j)
        res["j-major seq"] = model_seq(A, B, Cm, [8 * g + j for j in range(8) for g in range(K // 8)])
        res["j-major groups(4 lanes) exact"] = model_grouped(A, B, Cm, [[8 * g + j for g in range(K // 8)] for j in range(8)])
        res["products first (exact), then +C"] = None
        s = np.zeros(Cm.shape, np.float64)
        for k in range(K):
            s += A[:, :, None, k].astype(np.float64) * B[:, None, :, k].astype(np.float64)
        res["products first (exact), then +C"] = (s.astype(f32) + Cm).astype(f32)
        for k_, v in res.items():
            print(f"  {name:7s} {k_:36s} match {np.mean(v == D):.5f}  maxrel {np.max(np.abs(v - D) / (np.abs(D) + 1e-30)):.2e}")
