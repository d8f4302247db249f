"""CPU-side checks: the C-ABI library loads and exports every symbol include/fav.h
declares, the ctypes mirror of fav_config matches the C layout, the product path
fails loudly without a GPU, and the shard/gather logic of the N>1 path is correct
under a 2-rank gloo group (the per-rank classifier is stood in for by the oracle,
used here as the checker's input, never shipped)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from failure_aware_vision_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "fav.h")).read()
    names = set(re.findall(r"\b(fav_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 18
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in fav.h but not exported"


def test_config_struct_layout_and_defaults(lib):
    from failure_aware_vision_amd import _lib
    c = _lib.FavConfig()
    lib.fav_default_config(C.byref(c), _lib.ARCH_RESNET50)
    assert c.struct_size == C.sizeof(_lib.FavConfig)
    assert (c.arch, c.num_classes, c.in_h, c.in_w, c.n_samples) == (1, 1000, 224, 224, 1)
    assert abs(c.tau - 0.5) < 1e-7 and abs(c.temperature - 1.0) < 1e-7 and c.regroup_block == -1
    assert np.allclose(list(c.mean), [0.485, 0.456, 0.406]) and np.allclose(list(c.stdev), [0.229, 0.224, 0.225])
    lib.fav_default_config(C.byref(c), _lib.ARCH_RESNET18_CIFAR)
    assert (c.num_classes, c.in_h) == (10, 32)
    assert lib.fav_abi_version() == 2


def test_no_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from failure_aware_vision_amd import Backend, _lib
    with pytest.raises(RuntimeError):
        Backend("resnet18_cifar", b"", max_batch=1)
    c = _lib.FavConfig()
    lib.fav_default_config(C.byref(c), 0)
    h = C.c_void_p()
    st = lib.fav_create(C.byref(c), C.byref(h))
    assert st == 5 and b"no CPU fallback" in lib.fav_last_error(None)
    c.struct_size = 12
    assert lib.fav_create(C.byref(c), C.byref(h)) == 1


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "failure_aware_vision_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert not re.search(r"(CDLL|dlopen|open)\([^)]*oracle", src), f


def test_shard_range_covers_batch():
    from failure_aware_vision_amd import shard_range
    for n in (1, 7, 8, 255, 256, 10000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_site_masks():
    from failure_aware_vision_amd import weights
    assert weights.site_mask_for(1, "all_blocks") == 0xFFFF
    assert weights.site_mask_for(1, "last_layer") == 1 << 16
    assert weights.site_mask_for(1, "layer4+fc") == (1 << 12) | (1 << 13) | (1 << 14) | (1 << 16)
    assert weights.site_mask_for(0, "all_blocks") == 0xFF
    assert weights.site_mask_for(1, "none") == 0


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from failure_aware_vision_amd import classify_sharded, shard_range, synth, weights
from oracle import fav_oracle as O
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
blob, _ = weights.make_synthetic("resnet18_cifar", seed=1)
model = O.parse_blob(blob)
cfg = O.ClassifyConfig(n_samples=3, site_mask=weights.site_mask_for(0, "all_blocks"), p=0.1, seed=4,
                       exact=True)   # fixed summation order: a frame's result cannot depend on its shard
n = 11
frames = synth.synthetic_frames_u8(n, 32, 32, seed=5)
def stand_in(local, first_index=0):   # plays the per-rank Backend.classify
    ids = np.arange(first_index, first_index + local.shape[0])
    return O.classify(model, local, cfg, img_ids=ids)
s, e = shard_range(n, rank, world)
labels, conf = classify_sharded(stand_in, frames[s:e], n, rank, world)
full_l, full_c = O.classify(model, frames, cfg)
assert labels.dtype == torch.int32 and conf.dtype == torch.float32
assert np.array_equal(labels.numpy(), full_l), (labels, full_l)
assert np.array_equal(conf.numpy(), full_c)          # bit-for-bit: the gather moves bits, not values
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharded_classify_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def _build_c_example(tmp_path):
    exe = str(tmp_path / "classify_host")
    libdir = os.path.join(ROOT, "failure_aware_vision_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "classify_host.c"), "-o", exe, "-L" + libdir, "-lfav_hip",
                           "-Wl,-rpath," + libdir])
    return exe


def test_plain_c_caller_links_and_reports_missing_gpu(lib, tmp_path):
    """include/fav.h is C99-clean and a C program links against the library; without a GPU it gets
    FAV_ERR_NO_DEVICE (5) and the 'no CPU fallback' message, not a crash."""
    import torch
    exe = _build_c_example(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 64 and "ABI version 2" in r.stderr
    if torch.cuda.is_available():
        pytest.skip("GPU present: the classify run is covered by the gpu-marked test")
    (tmp_path / "w.favw").write_bytes(b"\0" * 64)
    (tmp_path / "f.u8").write_bytes(b"\0" * (32 * 32 * 3))
    r = subprocess.run([exe, "0", str(tmp_path / "w.favw"), str(tmp_path / "f.u8"), "1", "32", "32"], capture_output=True, text=True)
    assert r.returncode == 5 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_plain_c_caller_matches_python_backend(lib, tmp_path):
    """The same frames through a plain-C host and through the Python Backend: identical labels and confidences."""
    import torch
    from failure_aware_vision_amd import Backend, synth, weights
    exe = _build_c_example(tmp_path)
    blob, _ = weights.make_synthetic("resnet18_cifar", seed=1)
    frames = synth.synthetic_frames_u8(6, 32, 32, seed=11)
    (tmp_path / "w.favw").write_bytes(blob)
    (tmp_path / "f.u8").write_bytes(frames.tobytes())
    r = subprocess.run([exe, "0", str(tmp_path / "w.favw"), str(tmp_path / "f.u8"), "6", "32", "32"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = [ln.split() for ln in r.stdout.strip().splitlines()]
    be = Backend("resnet18_cifar", blob, max_batch=6)
    labels, conf = be.classify(torch.from_numpy(frames).cuda())
    be.close()
    assert [int(x[0]) for x in rows] == labels.cpu().tolist()
    assert np.allclose([float(x[1]) for x in rows], conf.cpu().numpy(), rtol=0, atol=1e-7)


# ---- checkpoint blob validation (file-supplied, hence hostile) ----------------------------------------------------
def _check(lib, blob: bytes):
    err = C.create_string_buffer(200)
    st = lib.fav_check_blob(blob, len(blob), err, 200)
    return st, err.value.decode()


def test_check_blob_accepts_real_checkpoints_and_rejects_hostile_tables(lib, r18_blob):
    import struct
    from failure_aware_vision_amd import weights
    blob, _ = r18_blob
    assert _check(lib, blob) == (0, "")
    vblob, _ = weights.make_synthetic_vit("vit_tiny", seed=3)
    assert _check(lib, vblob)[0] == 0
    BAD = 2   # FAV_ERR_BAD_BLOB
    assert _check(lib, blob[:16])[0] == BAD and _check(lib, b"")[0] == BAD
    assert _check(lib, blob[:32 + 48 * 3])[0] == BAD                       # truncated layer table
    assert _check(lib, blob[:len(blob) - 64])[0] == BAD                     # truncated data
    nl = struct.unpack_from("<I", blob, 16)[0]

    def patched(layer, field_off, fmt, value):
        b = bytearray(blob)
        struct.pack_into(fmt, b, 32 + 48 * layer + field_off, value)
        return bytes(b)

    for layer in (0, nl // 2, nl - 1):
        w_off, b_off = struct.unpack_from("<2Q", blob, 32 + 48 * layer + 32)
        # an offset near 2^64 must not wrap past the range check (off + bytes overflow)
        for off in (2 ** 64 - 8, 2 ** 64 - 64, 2 ** 63, len(blob), len(blob) - 2):
            st, msg = _check(lib, patched(layer, 32, "<Q", off))
            assert st == BAD and "out of range" in msg, (layer, off, msg)
            assert _check(lib, patched(layer, 40, "<Q", off))[0] == BAD
        assert _check(lib, patched(layer, 32, "<Q", w_off + 1))[0] == BAD   # bf16 data must be 2-byte aligned
        assert _check(lib, patched(layer, 40, "<Q", b_off + 2))[0] == BAD   # fp32 data must be 4-byte aligned
        assert _check(lib, patched(layer, 32, "<Q", 8))[0] == BAD           # data inside the header
        assert _check(lib, patched(layer, 0, "<I", 0))[0] == BAD            # cout = 0
        assert _check(lib, patched(layer, 0, "<I", 0xFFFFFFFF))[0] == BAD   # cout * k overflows
    b = bytearray(blob)
    struct.pack_into("<I", b, 16, 0x7FFFFFFF)                               # absurd layer count
    assert _check(lib, bytes(b))[0] == BAD
    b = bytearray(blob); b[0] ^= 0xFF
    assert _check(lib, bytes(b))[0] == BAD
    # a non-finite weight or bias is refused: it would defeat the pixel sanitiser (NaN in, NaN out whatever the frame)
    w_off, b_off = struct.unpack_from("<2Q", blob, 32 + 48 * 2 + 32)
    for off, fmt_, val in ((w_off + 6, "<H", 0x7FC0), (w_off, "<H", 0xFF80), (b_off + 4, "<I", 0x7F800000), (b_off, "<I", 0xFFC00000)):
        b = bytearray(blob)
        struct.pack_into(fmt_, b, off, val)
        st, msg = _check(lib, bytes(b))
        assert st == BAD and "non-finite" in msg, (off, msg)


def test_oracle_sanitises_fp32_pixels():
    """fp32 frames: NaN -> 0, everything else clamped to [-64, 64], pixels in [0, 1] untouched (the GPU side: fav_sanitize_px;
    reference behaviour mirrored: a garbage frame yields a status, never an exception, signal_analyzer.py:145-171)."""
    import numpy as np
    from oracle import fav_oracle as O
    x = np.array([0.0, 1.0, 0.25, np.nan, np.inf, -np.inf, 1e30, -3e38, 64.0, -64.5], np.float32)
    y = O.sanitize_pixels(x)
    assert np.array_equal(y, np.array([0.0, 1.0, 0.25, 0.0, 64.0, -64.0, 64.0, -64.0, 64.0, -64.0], np.float32))
    frames = np.random.default_rng(0).random((2, 8, 8, 3), dtype=np.float32)
    assert np.array_equal(O.sanitize_pixels(frames), frames)
    bad = frames.copy(); bad[0, 1, 2, 0] = np.nan; bad[1, 3, 3, 1] = np.inf
    n = O.normalize_input(bad, (0.485, 0.456, 0.406), O.inv_std32((0.229, 0.224, 0.225)))
    assert np.isfinite(n).all()


# ---- bench.py launch logic (no GPU, no torch import in the parent) -------------------------------------------------
def test_bench_self_launch_command_and_world_size_check():
    import bench
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "3"], port=29555)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-5].endswith("bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    # a launcher that started a different number of ranks than --gpus is an error, not silently ignored
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    # the parent of a self-launch must not have imported torch (it would initialise HIP before the exec of the children)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert not re.search(r"^(import|from)\s+torch", src, re.M)                      # no module-level torch import
    body = src[src.index("def main()"):]
    assert body.index("subprocess.call(launch_command") < body.index("import torch")   # spawn first, torch later


def test_pmc_traffic_is_tied_to_the_kernel_sources():
    import json
    import bench
    tr = bench.pmc_traffic()
    d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    if d.get("kernel_source_sha256") == bench.kernel_source_sha():
        assert tr["bytes_per_launch"] == d["conv_bytes_per_launch"]
    else:
        assert tr.get("stale") is True and "bytes_per_launch" not in tr


_NCCL_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from failure_aware_vision_amd import Backend, classify_sharded, shard_range, synth, weights
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group("nccl", device_id=torch.device("cuda", rank))
blob, _ = weights.make_synthetic("resnet18_cifar", seed=1)
kw = dict(n_samples=3, dropout_policy="all_blocks", dropout_p=0.1, seed=4)
n = 11
frames = synth.synthetic_frames_u8(n, 32, 32, seed=5)
be = Backend("resnet18_cifar", blob, device=rank, max_batch=n, **kw)
s, e = shard_range(n, rank, world)
# the Backend itself: its confidence head writes the packed records into the all-gather send slot
labels, conf = classify_sharded(be, torch.from_numpy(frames[s:e]).cuda(), n, rank, world)
full_l, full_c = be.classify(torch.from_numpy(frames).cuda())          # the 1-GPU result, on this rank's GPU
assert torch.equal(labels, full_l) and torch.equal(conf, full_c), (labels, full_l)   # bit for bit
l2, c2 = classify_sharded(be.classify, torch.from_numpy(frames[s:e]).cuda(), n, rank, world)   # the generic (function) form
assert torch.equal(l2, full_l) and torch.equal(c2, full_c)
be.close()
# a non-headline config over the same path: the ViT miniature with entropy confidence (BASELINE configs[4]'s arithmetic),
# equal shards (the receive buffer is the result)
vblob, _ = weights.make_synthetic_vit("vit_tiny", seed=3)
vit = Backend("vit_tiny", vblob, device=rank, max_batch=8, temperature=1.5, conf_kind="entropy")
vframes = torch.from_numpy(synth.synthetic_frames_u8(8, 64, 64, seed=11)).cuda()
s, e = shard_range(8, rank, world)
lv, cv = classify_sharded(vit, vframes[s:e], 8, rank, world)
fl, fc = vit.classify(vframes)
assert torch.equal(lv, fl) and torch.equal(cv, fc)
vit.close()
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.gpu
def test_sharded_classify_nccl_world2(tmp_path):
    """The N>1 path over RCCL on two real GPUs: gathered result == 1-GPU result bit for bit.
    Skipped on a one-GPU box (two ranks cannot share one device under RCCL)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    script = tmp_path / "worker.py"
    script.write_text(_NCCL_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


# ---- the static launch schedule, replayed symbolically (no GPU) ---------------------------------------------------
def _plan(lib, arch, flags=0, **kw):
    from failure_aware_vision_amd import _lib
    cfg = _lib.FavConfig()
    lib.fav_default_config(C.byref(cfg), arch)
    for k, v in kw.items():
        setattr(cfg, k, v)
    buf = C.create_string_buffer(1 << 17)
    st = lib.fav_plan_schedule(C.byref(cfg), flags, buf, len(buf))
    assert st == 0, buf.value
    ops = []
    for line in buf.value.decode().splitlines():
        f = dict(kv.split("=") for kv in line.split()[2:])
        ops.append({k: int(v) for k, v in f.items()})
    return ops


def _replay(ops):
    """Every buffer holds a symbolic tensor (a nested tuple naming the ops that made it).  Rotating buffers do not
    survive a phase boundary (only the phase output does), a read of a buffer nothing wrote is an error, and an op may
    not write a buffer it reads.  Returns the symbol of the last phase's output (the logits)."""
    bufs, phase, phase_in, phase_out = {}, 0, None, None
    NONE, FRAMES, PIN, POUT = -4, -1, -2, -3

    def rd(b):
        if b == NONE:
            return None
        if b == FRAMES:
            return "frames"
        if b == PIN:
            assert phase_in is not None
            return phase_in
        return bufs[b]          # KeyError = read of a buffer nothing wrote in this phase

    def wr(b, sym):
        nonlocal phase_out
        if b == POUT:
            phase_out = sym
        elif b != NONE:
            bufs[b] = sym

    for o in ops:
        if o["phase"] != phase:
            assert phase_out is not None, "phase ended without an output"
            phase, phase_in, phase_out, bufs = o["phase"], phase_out, None, {}
        # a tail behind the entry dropout may take its residual from the cached phase input, dropped in its own epilogue
        x = rd(o["in"])
        r = ("entry_dropout", o["esite"], rd(PIN)) if o.get("rese") else rd(o["res"])
        writes = [b for b in (o["out"], o["out2"]) if b >= 0]
        assert len(set(writes)) == len(writes) and not (set(writes) & {o["in"], o["res"]}), o
        k = o["kind"]
        if k == 0:
            wr(o["out"], ("im2col", x))
        elif k == 1:
            wr(o["out"], ("conv", o["layer"], x, r, o["relu"], o["site"]))
        elif k == 2:
            wr(o["out"], ("maxpool", x))
        elif k == 3:
            wr(o["out"], ("avgpool", x, o["site"]))
        elif k == 4:
            wr(o["out"], ("entry_dropout", o["site"], x))
        elif k == 7:                  # the fused ImageNet stem: im2col + GEMM + max pool in one launch
            wr(o["out"], ("maxpool", ("conv", o["layer"], ("im2col", x), None, 1, -1)))
        elif k == 6:                  # entry dropout + the 1x1 reduce behind it
            y = ("entry_dropout", o["site"], x)
            if not o.get("skipy"):    # the dropped copies themselves are stored only if somebody reads them
                wr(o["out"], y)
            wr(o["out2"], ("conv", o["la"], y, None, 1, -1))
        elif k == 5:
            t2 = ("conv", o["layer"], x, None, 1, -1) if o["layer"] >= 0 else x
            y = ("conv", o["lc"], t2, r, 1, o["site"])
            wr(o["out"], y)
            if o["la"] >= 0:
                wr(o["out2"], ("conv", o["la"], y, None, 1, -1))
            else:
                assert o["out2"] == NONE
        else:
            raise AssertionError(o)
    return phase_out


def test_fused_schedule_computes_the_layer_by_layer_dataflow(lib):
    """fav_plan_schedule (no device): for every architecture, dropout policy, regrouping point and frame size the schedule
    with fused bottleneck tails must be the same computation as the layer-by-layer one - same convolutions, same inputs,
    same residuals, same dropout sites - and may never read a rotating buffer across a phase boundary."""
    from failure_aware_vision_amd import weights
    n_tail = 0
    for arch in (1, 0):
        nb = weights.n_blocks(arch)
        masks = [0, weights.site_mask_for(arch, "all_blocks"), weights.site_mask_for(arch, "last_layer"),
                 weights.site_mask_for(arch, "layer4+fc"), 0b101000, 1 << 3, (1 << nb) | 1]
        for mask in masks:
            for regroup in (-1, 0, 1, 3, 7, 8, nb):
                for hw in ((224, 224), (64, 64), (240, 320), (1024, 2048)):
                    kw = dict(in_h=hw[0], in_w=hw[1], site_mask=mask, n_samples=3 if mask else 1, dropout_p=0.1 if mask else 0.0,
                              regroup_block=regroup)
                    fused, plain = _plan(lib, arch, 0, **kw), _plan(lib, arch, 1, **kw)
                    assert not any(o["kind"] == 5 for o in plain)
                    n_tail += sum(o["kind"] == 5 for o in fused)
                    a, b = _replay(fused), _replay(plain)
                    assert a == b, (arch, mask, regroup, hw)
                    assert a[0] == "conv" and "avgpool" in (a[2][0], a[2][2][0] if a[2][0] == "entry_dropout" else "")   # fc over the pooled features
    assert n_tail > 1000                                                  # the sweep really exercised fused schedules
    # the headline configuration: which ops it consists of
    ops = _plan(lib, 1, 0, site_mask=weights.site_mask_for(1, "all_blocks"), n_samples=30, dropout_p=0.1)
    tails = [o for o in ops if o["kind"] == 5]
    assert len(tails) == 16 and sum(o["la"] >= 0 for o in tails) == 5      # every bottleneck of the suffix ends in a fused tail
    assert sum(o["layer"] >= 0 for o in tails) == 11     # conv_b inside the tail: layers 1-2 (6) and layer 3's five identity blocks
    assert sum(o["kind"] == 6 for o in ops) == 1 and not any(o["kind"] == 4 for o in ops) and len({o["phase"] for o in ops}) == 3
    # the T dropped copies of the prefix output are not stored: the tail behind the entry takes them from the cached tensor
    assert [o["skipy"] for o in ops if o["kind"] == 6] == [1] and sum(o["rese"] for o in tails) == 1
    # the validation mode keeps the separate launches
    assert not any(o["kind"] == 5 for o in _plan(lib, 1, 0, math_mode=1))


def test_bench_config_table_and_roofline_objects():
    """bench.py without a GPU: every --config entry names its BASELINE config, scaling mode and roofline bound, and the
    roofline builder turns a per-class profile into the objects the driver's line carries."""
    import bench
    args = bench.parse_args([])
    table = bench.config_table(args)
    assert set(table) == {"mc30", "single", "ens5", "vit"}
    assert table["mc30"]["metric"] == "frames/sec, ResNet-50 MC-Dropout T=30 224x224 batch=256" and table["mc30"]["scaling"] == "weak"
    assert abs(table["mc30"]["gflop"] - 225.098) < 0.01 and table["mc30"]["bound"] == "hbm"
    assert table["ens5"]["scaling"] == table["vit"]["scaling"] == "strong"
    assert table["ens5"]["global_batch"] == 256 and table["vit"]["global_batch"] == 512 and table["single"]["per_gpu"] == 256
    for c in table.values():
        assert "BASELINE configs[" in c["workload"]
    prof = {k: dict(ms=0.0, flops=0.0, bytes=0.0, launches=0) for k in ("stem_im2col", "conv_igemm", "maxpool", "avgpool", "entry_dropout", "head")}
    prof["conv_igemm"] = dict(ms=100.0, flops=8.0e13, bytes=3.5e11, launches=50)
    prof["head"]["ms"] = 1.0
    hbm, mfma = bench.rooflines(prof, "hbm")
    assert hbm["bound"] == "hbm" and mfma["bound"] == "mfma"
    assert abs(hbm["achieved"] - 3500.0) < 1e-6 and abs(hbm["frac"] - 3500.0 / 8000.0) < 1e-9 and hbm["unit"] == "GB/s"
    assert abs(mfma["achieved"] - 800.0) < 1e-6 and abs(mfma["frac"] - 0.32) < 1e-9 and mfma["unit"] == "TFLOP/s"
    assert abs(hbm["avg_launch_us"] - 2000.0) < 1e-6 and set(("peak", "traffic", "launches")) <= set(hbm)
    m2, h2 = bench.rooflines(prof, "mfma", "note", wall_s=0.05)          # overlapping streams: work over wall time
    assert m2["bound"] == "mfma" and abs(m2["achieved"] - 1600.0) < 1e-6 and abs(m2["achieved_per_launch_events"] - 800.0) < 1e-6
    assert h2["traffic"] is None and m2["timing"] == "note"
