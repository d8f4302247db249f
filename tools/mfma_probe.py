"""Runs tools/mfma_probe.so on crafted operands and saves inputs + raw outputs for
offline analysis of the bf16 MFMA's internal summation order."""
import ctypes as C, os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "mfma_probe.so"))
for f in (lib.run_probe16, lib.run_probe32):
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    f.restype = C.c_int
rng = np.random.default_rng(0)

def bf16r(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32).reshape(x.shape)

def gen(kind, shape):
    if kind == "normal":
        return bf16r(rng.standard_normal(shape).astype(np.float32))
    if kind == "wide":
        return bf16r((rng.standard_normal(shape) * np.exp2(rng.integers(-6, 7, shape))).astype(np.float32))
    if kind == "pos":
        return bf16r(np.abs(rng.standard_normal(shape)).astype(np.float32))

out = {}
NP = 384
for name, (ka, kb, cs) in {"normal": ("normal", "normal", 4.0), "wide": ("wide", "wide", 64.0), "pos": ("pos", "pos", 0.0),
                           "bigc": ("normal", "normal", 4096.0)}.items():
    for tag, (m, k, fn) in {"16": (16, 32, lib.run_probe16), "32": (32, 16, lib.run_probe32)}.items():
        A = gen(ka, (NP, m, k)); Bt = gen(kb, (NP, m, k))
        Cm = (rng.standard_normal((NP, m, m)) * cs).astype(np.float32)
        a = torch.from_numpy((A.view(np.uint32) >> 16).astype(np.uint16).view(np.int16)).cuda()
        b = torch.from_numpy((Bt.view(np.uint32) >> 16).astype(np.uint16).view(np.int16)).cuda()
        c = torch.from_numpy(Cm).cuda()
        d = torch.empty_like(c)
        rc = fn(a.data_ptr(), b.data_ptr(), c.data_ptr(), d.data_ptr(), NP)
        assert rc == 0, rc
        out[f"{name}_{tag}_A"] = (A.view(np.uint32) >> 16).astype(np.uint16)
        out[f"{name}_{tag}_B"] = (Bt.view(np.uint32) >> 16).astype(np.uint16)
        out[f"{name}_{tag}_C"] = Cm
        out[f"{name}_{tag}_D"] = d.cpu().numpy()
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/mfma_probe.npz", **out)
print("saved", sum(v.nbytes for v in out.values()) / 1e6, "MB")
