import numpy as np
d = np.load("gpurun_out/mfma_probe.npz")
f32 = np.float32
def tof(b): return (b.astype(np.uint32) << 16).view(np.float32)
def r32(x): return x.astype(f32).astype(np.float64)

def run(A, B, Cm, K, fn):
    acc = Cm.astype(np.float64)
    for g0 in range(0, K, 8):
        P = [np.broadcast_to(A[:, :, None, k].astype(np.float64) * B[:, None, :, k].astype(np.float64), acc.shape) for k in range(g0, g0 + 8)]
        acc = fn(P, acc)
    return acc.astype(f32)

models = {
 "tree f32 nodes, then +acc": lambda P, a: r32(r32(r32(r32(P[0]+P[1])+r32(P[2]+P[3])) + r32(r32(P[4]+P[5])+r32(P[6]+P[7]))) + a),
 "exact8 -> f32, then +acc": lambda P, a: r32(r32(sum(P)) + a),
 "exact4+exact4 each f32, + , +acc": lambda P, a: r32(r32(r32(sum(P[:4])) + r32(sum(P[4:]))) + a),
 "acc + exact4 (rne), + exact4 (rne)": lambda P, a: r32(r32(a + sum(P[:4])) + sum(P[4:])),
 "acc + exact pairs seq": lambda P, a: r32(r32(r32(r32(a + P[0]+P[1]) + P[2]+P[3]) + P[4]+P[5]) + P[6]+P[7]),
 "even/odd: acc + (0,2,4,6) then (1,3,5,7)": lambda P, a: r32(r32(a + P[0]+P[2]+P[4]+P[6]) + P[1]+P[3]+P[5]+P[7]),
 "exact 8 + acc (single rne)": lambda P, a: r32(a + sum(P)),
}
for tag, K in (("16", 32), ("32", 16)):
    for name in ("wide", "normal", "pos", "bigc"):
        A, B, Cm, D = tof(d[f"{name}_{tag}_A"])[:64], tof(d[f"{name}_{tag}_B"])[:64], d[f"{name}_{tag}_C"][:64], d[f"{name}_{tag}_D"][:64]
        print(tag, name, {k: round(float(np.mean(run(A, B, Cm, K, fn) == D)), 4) for k, fn in models.items()})
# look at the mismatches of the best model on 'wide': relation between error and exponent spread
A, B, Cm, D = tof(d["wide_16_A"])[:64], tof(d["wide_16_B"])[:64], d["wide_16_C"][:64], d["wide_16_D"][:64]
ref = run(A, B, Cm, 32, models["exact 8 + acc (single rne)"])
bad = ref != D
ulp = np.spacing(np.abs(D)).astype(np.float64)
err = (D.astype(np.float64) - ref.astype(np.float64)) / ulp
print("mismatch frac", bad.mean(), "err in ulps: min/max", err[bad].min(), err[bad].max(), "hist", np.histogram(err[bad], bins=[-1e9,-3.5,-2.5,-1.5,-0.5,0.5,1.5,2.5,3.5,1e9])[0])
