"""Single-group probes around a power of two of the accumulator (both directions)."""
import ctypes as C, os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "mfma_probe.so"))
lib.run_probe16.argtypes = [C.c_void_p] * 4 + [C.c_int]; lib.run_probe16.restype = C.c_int
rng = np.random.default_rng(11)
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
for kexp in (8, 10, 12, 14, 16):
    for side in ("below", "above"):
        for nnz in (1, 8):
            A = np.zeros((NP, 16, 32), np.float32); Bt = np.zeros((NP, 16, 32), np.float32)
            A[:, :, :nnz] = bf16r(rng.standard_normal((NP, 16, nnz)).astype(np.float32) * 1.5)
            Bt[:, :, :nnz] = bf16r(rng.standard_normal((NP, 16, nnz)).astype(np.float32) * 1.5)
            sgn = rng.choice([-1.0, 1.0], (NP, 16, 16))
            if side == "below":
                mant = 1.0 - rng.random((NP, 16, 16)) * 2.0 ** -10
            else:
                mant = 1.0 + rng.random((NP, 16, 16)) * 2.0 ** -10
            run(f"k{kexp}_{side}_{nnz}", A, Bt, (sgn * mant * 2.0 ** kexp).astype(np.float32))
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/mfma_probe4.npz", **cases)
print("saved", sum(v.nbytes for v in cases.values()) / 1e6, "MB")
