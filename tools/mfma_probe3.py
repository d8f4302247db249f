"""Probe the carry-out regime of the bf16 MFMA adder (accumulator just below a power of two)."""
import ctypes as C, os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "mfma_probe.so"))
lib.run_probe16.argtypes = [C.c_void_p] * 4 + [C.c_int]; lib.run_probe16.restype = C.c_int
rng = np.random.default_rng(7)
def bf16r(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32).reshape(np.shape(x))
NP = 768
cases = {}
def run(name, A, Bt, Cm):
    a = torch.from_numpy((A.view(np.uint32) >> 16).astype(np.uint16).view(np.int16)).cuda()
    b = torch.from_numpy((Bt.view(np.uint32) >> 16).astype(np.uint16).view(np.int16)).cuda()
    c = torch.from_numpy(Cm).cuda(); d = torch.empty_like(c)
    assert lib.run_probe16(a.data_ptr(), b.data_ptr(), c.data_ptr(), d.data_ptr(), A.shape[0]) == 0
    cases[name + "_A"] = (A.view(np.uint32) >> 16).astype(np.uint16); cases[name + "_B"] = (Bt.view(np.uint32) >> 16).astype(np.uint16)
    cases[name + "_C"] = Cm; cases[name + "_D"] = d.cpu().numpy()
for name, kexp, pscale in (("carry6", 6, 1.0), ("carry8", 8, 1.0), ("carry10", 10, 1.0), ("carry12", 12, 1.0), ("carry4", 4, 1.0), ("carry2", 2, 1.0), ("carry0", 0, 0.25)):
    A = bf16r((rng.standard_normal((NP, 16, 32)) * pscale).astype(np.float32))
    Bt = bf16r((rng.standard_normal((NP, 16, 32)) * pscale).astype(np.float32))
    sgn = rng.choice([-1.0, 1.0], (NP, 16, 16))
    mant = 1.0 - rng.random((NP, 16, 16)) * 2.0 ** -9          # just below a power of two
    Cm = (sgn * mant * 2.0 ** kexp).astype(np.float32)
    run(name, A, Bt, Cm)
# heavy cancellation: c ~ -sum of products
A = bf16r(rng.standard_normal((NP, 16, 32)).astype(np.float32)); Bt = bf16r(rng.standard_normal((NP, 16, 32)).astype(np.float32))
S = np.einsum("pmk,pnk->pmn", A.astype(np.float64), Bt.astype(np.float64))
run("cancel", A, Bt, (-S * (1 + rng.standard_normal(S.shape) * 1e-3)).astype(np.float32))
# ordinary activations-like: positive x, mixed w, running accumulators of various sizes
A = bf16r(np.abs(rng.standard_normal((NP, 16, 32))).astype(np.float32)); Bt = bf16r((rng.standard_normal((NP, 16, 32)) * 0.1).astype(np.float32))
run("relu_like", A, Bt, (rng.standard_normal((NP, 16, 16)) * 3).astype(np.float32))
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/mfma_probe3.npz", **cases)
print("saved", sum(v.nbytes for v in cases.values()) / 1e6, "MB")
