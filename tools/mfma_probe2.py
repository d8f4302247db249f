"""Structured probes of v_mfma_f32_16x16x32_bf16 for offline modelling of its adder datapath."""
import ctypes as C, os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "mfma_probe.so"))
lib.run_probe16.argtypes = [C.c_void_p] * 4 + [C.c_int]; lib.run_probe16.restype = C.c_int
rng = np.random.default_rng(1)

def bf16r(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32).reshape(np.shape(x))

def rand_bf16(shape, lo=-1, hi=1):
    m = 1.0 + rng.integers(0, 128, shape) / 128.0
    e = rng.integers(lo, hi + 1, shape)
    s = rng.choice([-1.0, 1.0], shape)
    return (s * m * np.exp2(e)).astype(np.float32)

NP = 512
cases = {}
def run(name, A, Bt, Cm):
    a = torch.from_numpy((A.view(np.uint32) >> 16).astype(np.uint16).view(np.int16)).cuda()
    b = torch.from_numpy((Bt.view(np.uint32) >> 16).astype(np.uint16).view(np.int16)).cuda()
    c = torch.from_numpy(Cm).cuda(); d = torch.empty_like(c)
    assert lib.run_probe16(a.data_ptr(), b.data_ptr(), c.data_ptr(), d.data_ptr(), A.shape[0]) == 0
    cases[name + "_A"] = (A.view(np.uint32) >> 16).astype(np.uint16); cases[name + "_B"] = (Bt.view(np.uint32) >> 16).astype(np.uint16)
    cases[name + "_C"] = Cm; cases[name + "_D"] = d.cpu().numpy()

def sparse(ks, c_scale_exp=None, gap=None):
    A = np.zeros((NP, 16, 32), np.float32); Bt = np.zeros((NP, 16, 32), np.float32)
    for i, k in enumerate(ks):
        lo, hi = (-1, 1) if gap is None or i == 0 else (-1 - gap, 1)
        A[:, :, k] = rand_bf16((NP, 16), lo // 2, hi // 2 if hi else 0)
        Bt[:, :, k] = rand_bf16((NP, 16), lo - lo // 2, 1)
    if c_scale_exp is None:
        Cm = np.zeros((NP, 16, 16), np.float32)
    else:
        Cm = (rng.standard_normal((NP, 16, 16)) * np.exp2(rng.integers(-c_scale_exp, c_scale_exp + 1, (NP, 16, 16)))).astype(np.float32)
    return A.astype(np.float32), Bt.astype(np.float32), Cm

run("one_c", *sparse([0], 30))                # single product + accumulator, wide exponent gaps
run("two_same", *sparse([0, 1], None, 30))     # two products in one lane group, no accumulator
run("two_same_far", *sparse([0, 7], None, 30))
run("two_diff", *sparse([0, 8], None, 30))     # two products in different lane groups
run("two_c", *sparse([0, 1], 20, 20))
run("three_same", *sparse([0, 1, 2], None, 24))
run("four_same", *sparse([0, 1, 2, 3], None, 24))
run("eight_same", *sparse(list(range(8)), None, 16))
run("eight_c", *sparse(list(range(8)), 12, 12))
run("g1_only", *sparse(list(range(8, 16)), 12, 12))
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/mfma_probe2.npz", **cases)
print("saved", sum(v.nbytes for v in cases.values()) / 1e6, "MB")
