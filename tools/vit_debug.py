"""Teacher-forced stage-by-stage comparison of the ViT kernels with the oracle (production mode)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from failure_aware_vision_amd import _lib, synth, weights
from oracle import fav_oracle as O
from test_gpu_ops import run_conv, dev_bf16, host_f32
lib = _lib.load()
arch = sys.argv[1] if len(sys.argv) > 1 else "vit_b16"
hw = (224, 224) if arch == "vit_b16" else (64, 64)
blob, info = weights.make_synthetic_vit(arch, seed=3, in_hw=hw)
m = O.parse_blob(blob); c = O.VIT_CFG[m.arch]; Ls = m.layers
u8 = synth.synthetic_frames_u8(2, hw[0], hw[1], seed=21)
frames = synth.gaussian_noise_f32(u8, 3, seed=3)
cfg = O.ClassifyConfig()
xn = O.normalize_input(frames, cfg.mean, O.inv_std32(cfg.std))
b = xn.shape[0]; p, d = c["patch"], c["dim"]; gh, gw = hw[0] // p, hw[1] // p
EX = "mfma"
def lin(x, L, res=None, act=0):
    x4 = x.reshape(b, -1, 1, x.shape[-1])
    r4 = None if res is None else res.reshape(b, -1, 1, L.cout)
    got = run_conv(lib, x4, L.w.reshape(L.cout, 1, 1, -1), L.b, r4, 1, 0, relu=act, math_mode=0).reshape(b, -1, L.cout)
    y = O.gemm_acc(x.reshape(-1, x.shape[-1]), L.w.reshape(L.cout, -1), EX) + L.b
    if res is not None: y = y + res.reshape(-1, L.cout)
    if act == 2: y = O.gelu_exact(y)
    exp = O.bf16_round(y.reshape(b, -1, L.cout))
    return got, exp
def report(name, got, exp):
    bad = got != exp
    print(f"{name:28s} mismatches {bad.mean():.6f}  max|d| {np.abs(got-exp).max():.4g}", flush=True)
def ln(x, L):
    rows = x.shape[0] * x.shape[1]
    xd = dev_bf16(x); y = torch.empty((rows, d), dtype=torch.bfloat16, device="cuda")
    g, be = torch.from_numpy(L.w).cuda(), torch.from_numpy(L.b).cuda()
    _lib.check(lib.fav_op_layernorm(xd.data_ptr(), d, g.data_ptr(), be.data_ptr(), y.data_ptr(), rows, d, C.c_float(1e-6), None))
    torch.cuda.synchronize()
    return host_f32(y).reshape(x.shape), O.bf16_round(O.layernorm_exact(x, L.w, L.b))
patches = xn.reshape(b, gh, p, gw, p, 3).transpose(0, 1, 3, 2, 4, 5).reshape(b, gh * gw, p * p * 3)
got, emb = lin(patches, Ls[0]); report("patch gemm", got, emb)
pos = Ls[1].w.reshape(-1, d)
x = np.empty((b, gh * gw + 1, d), np.float32); x[:, 0] = pos[0]; x[:, 1:] = emb + pos[1:]; x = O.bf16_round(x)
li = 2
for blk in range(c["depth"]):
    ln1, qkv, proj, ln2, fc1, fc2 = Ls[li:li + 6]; li += 6
    got, y = ln(x, ln1); report(f"blk{blk} ln1", got, y)
    got, q = lin(y, qkv); report(f"blk{blk} qkv", got, q)
    qd = dev_bf16(q); out = torch.empty((b, q.shape[1], d), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.fav_op_attention(qd.data_ptr(), out.data_ptr(), b, q.shape[1], d, c["heads"], 0, None)); torch.cuda.synchronize()
    a = O.attention(q, c["heads"], EX); report(f"blk{blk} attention", host_f32(out), a)
    got, x2 = lin(a, proj, res=x); report(f"blk{blk} proj+res", got, x2)
    got, y2 = ln(x2, ln2); report(f"blk{blk} ln2", got, y2)
    got, hdn = lin(y2, fc1, act=2); report(f"blk{blk} fc1+gelu", got, hdn)
    got, x = lin(hdn, fc2, res=x2); report(f"blk{blk} fc2+res", got, x)
    if blk >= int(os.environ.get("NBLK", "2")) - 1: break
