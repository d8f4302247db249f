"""End-to-end parity of Backend.classify (HIP, through the C ABI) against the CPU
oracle, plus size-independent properties at BASELINE.json's full sizes.

Tolerances.  The reference pins nothing on this path (parity
unpinned), so the oracle is the spec.  Two bf16 pipelines that sum in different
orders differ by one bf16 ulp on a growing fraction of activations (a flipped
rounding perturbs every downstream sum; tests/test_oracle.py shows the same between
the NumPy oracle and torch-CPU), so end-to-end:
  * labels must be equal wherever the oracle's top-2 probability gap exceeds GAP_TOL,
  * confidences within CONF_TOL,
  * logits within LOGIT_RMS_TOL of the oracle's in rms relative to their spread.
Per-kernel parity (tests/test_gpu_ops.py) is tight; the exact math mode
(test_gpu_exact.py) is bit-for-bit.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from failure_aware_vision_amd import Backend, synth, weights  # noqa: E402
from failure_aware_vision_amd._lib import FavError  # noqa: E402
from oracle import fav_oracle as O  # noqa: E402

# measured on MI355X (gpurun_out/e2e_measured.txt, round 2): logit rms 0.006-0.015, no label differs, max |dconf| 0.023
GAP_TOL = 0.05
CONF_TOL = 0.04
LOGIT_RMS_TOL = 0.03


from conftest import note as _note  # noqa: E402


def check_against_oracle(be, model, frames, ocfg, first_index=0):
    if isinstance(frames, np.ndarray):
        dev_frames = torch.from_numpy(frames).cuda()
    labels, conf = be.classify(dev_frames, first_index=first_index)
    lg = be.logits()
    torch.cuda.synchronize()
    ids = np.arange(first_index, first_index + frames.shape[0])
    ol, oc, olg, opb = O.classify(model, frames, ocfg, img_ids=ids, return_logits=True)
    l, c, g = labels.cpu().numpy(), conf.cpu().numpy(), lg.cpu().numpy()
    assert g.shape == olg.shape
    rms = np.sqrt(((g - olg) ** 2).mean()) / olg.std()
    assert rms < LOGIT_RMS_TOL, f"logit rms error {rms:.4f}"
    srt = np.sort(opb, axis=1)
    gap = srt[:, -1] - srt[:, -2]
    bad = (l != ol) & (gap >= GAP_TOL)
    assert not bad.any(), f"labels differ beyond tolerance at {np.nonzero(bad)[0]}: {l[bad]} vs {ol[bad]}, gap {gap[bad]}"
    assert np.abs(c - oc).max() < CONF_TOL, np.abs(c - oc).max()
    _note(f"{be.arch} T={g.shape[0]} n={g.shape[1]}: logit rms {rms:.4f}, labels equal {(l == ol).mean():.3f}, largest oracle gap "
          f"among disagreements {gap[l != ol].max() if (l != ol).any() else 0:.4f}, max |dconf| {np.abs(c - oc).max():.4f}")
    return (l == ol).mean(), rms


def test_resnet18_config1_single_pass(r18_blob):
    """BASELINE config 1 shape (ResNet-18, 32x32, batch 32, max-softmax), on the GPU."""
    blob, _ = r18_blob
    model = O.parse_blob(blob)
    frames = synth.synthetic_frames_u8(32, 32, 32, seed=7)
    be = Backend("resnet18_cifar", blob, max_batch=32)
    agree, rms = check_against_oracle(be, model, frames, O.ClassifyConfig())
    assert agree >= 0.9
    # fp32 [0,1] frames take the other layout branch and must give the same answer as u8
    l1, c1 = be.classify(torch.from_numpy(frames).cuda())
    l2, c2 = be.classify(torch.from_numpy(frames.astype(np.float32) * np.float32(1 / 255.0)).cuda())
    assert torch.equal(l1, l2) and torch.equal(c1, c2)
    # numpy in -> numpy out through fav_classify_host
    l3, c3 = be.classify(frames)
    assert np.array_equal(l3, l1.cpu().numpy()) and np.array_equal(c3, c1.cpu().numpy())
    be.close()


@pytest.mark.parametrize("policy", ["last_layer", "layer4+fc", "all_blocks"])
def test_resnet18_mc_dropout(r18_blob, policy):
    blob, _ = r18_blob
    model = O.parse_blob(blob)
    frames = synth.synthetic_frames_u8(12, 32, 32, seed=8)
    mask = weights.site_mask_for(0, policy)
    be = Backend("resnet18_cifar", blob, max_batch=12, n_samples=5, dropout_policy=policy, dropout_p=0.1, seed=4,
                 conf_kind="entropy", chunk_a=7, chunk_b=16)  # ragged chunks on purpose
    ocfg = O.ClassifyConfig(n_samples=5, site_mask=mask, p=0.1, seed=4, conf_kind=O.CONF_ENTROPY)
    check_against_oracle(be, model, frames, ocfg, first_index=1000)
    assert be.logits().shape == (5, 12, 10)
    be.close()


def test_resnet50_small_frames(r50_blob):
    """ResNet-50 on 64x64 frames: every layer shape class (7x7/2 stem, 3x3/2, 1x1/2
    downsample, bottleneck residuals) at a size the oracle finishes in seconds."""
    blob, _ = r50_blob
    model = O.parse_blob(blob)
    frames = synth.synthetic_frames_u8(6, 64, 64, seed=9)
    be = Backend("resnet50", blob, in_hw=(64, 64), max_batch=6)
    check_against_oracle(be, model, frames, O.ClassifyConfig())
    be.close()
    be = Backend("resnet50", blob, in_hw=(64, 64), max_batch=6, n_samples=3, dropout_policy="all_blocks",
                 dropout_p=0.1, seed=4)
    ocfg = O.ClassifyConfig(n_samples=3, site_mask=weights.site_mask_for(1, "all_blocks"), p=0.1, seed=4)
    check_against_oracle(be, model, frames, ocfg, first_index=17)
    be.close()


def test_resnet50_224_vs_oracle(r50_blob):
    """BASELINE config 2/3 frame size, small batch: clean and Gaussian-noise severity 3."""
    blob, _ = r50_blob
    model = O.parse_blob(blob)
    frames = synth.synthetic_frames_u8(8, 224, 224, seed=7)
    be = Backend("resnet50", blob, max_batch=8)
    check_against_oracle(be, model, frames, O.ClassifyConfig())
    noisy = synth.gaussian_noise_f32(frames, 3, seed=3)
    labels, conf = be.classify(torch.from_numpy(noisy).cuda())
    ol, oc, olg, opb = O.classify(model, noisy, O.ClassifyConfig(), return_logits=True)
    srt = np.sort(opb, axis=1)
    gap = srt[:, -1] - srt[:, -2]
    l = labels.cpu().numpy()
    assert np.all((l == ol) | (gap < GAP_TOL))
    assert np.abs(conf.cpu().numpy() - oc).max() < CONF_TOL
    be.close()


def test_resnet50_every_layer_teacher_forced(r50_blob):
    """Production (bf16 MFMA) kernels on the REAL network, layer by layer: every one of the 53
    convolutions of ResNet-50 at 224x224 is launched on the oracle's own input for that layer
    (so errors cannot compound) and must match the oracle's output to one bf16 ulp on < 1 % of
    elements.  This isolates kernel correctness from the bf16 chaos of the end-to-end check."""
    import ctypes as C
    from failure_aware_vision_amd import _lib
    from test_gpu_ops import run_conv, assert_one_ulp
    blob, _ = r50_blob
    model = O.parse_blob(blob)
    net = O.OracleNet(model, exact=True)
    net.trace = []
    frames = synth.gaussian_noise_f32(synth.synthetic_frames_u8(5, 224, 224, seed=21), 3, seed=3)
    xn = O.normalize_input(frames, (0.485, 0.456, 0.406), O.inv_std32((0.229, 0.224, 0.225)))
    net.forward_logits(xn)
    lib = _lib.load()
    assert len(net.trace) == 53
    checked = 0
    for L, x, res, relu, y in net.trace:
        if L.cin % 64 != 0:
            continue                      # the 3-channel stem goes through im2col: covered by test_gpu_ops
        got = run_conv(lib, x, L.w, L.b, res, L.stride, L.pad, relu=1 if relu else 0)
        assert got.shape == y.shape
        assert_one_ulp(got, y)
        checked += 1
    assert checked == 52


def test_full_size_properties(r50_blob):
    """BASELINE headline shape (ResNet-50, 224x224, batch 256, MC-Dropout T = 30): properties
    that need no oracle run.  (a) determinism; (b) shard invariance: two half batches
    with first_image_index give bit-identical results (what the 8-GPU path relies on,
    SURVEY.md §8e); (c) chunking invariance: pass sizes change the schedule, not the
    result; (d) a frame's result does not depend on its neighbours."""
    blob, _ = r50_blob
    n, T = 256, 30        # the headline shape exactly: 256 frames x T = 30, all_blocks
    frames = torch.from_numpy(synth.synthetic_frames_u8(n, 224, 224, seed=21)).cuda()
    kw = dict(n_samples=T, dropout_policy="all_blocks", dropout_p=0.1, seed=4)
    be = Backend("resnet50", blob, max_batch=n, **kw)
    l0, c0 = be.classify(frames)
    lg0 = be.logits()
    l1, c1 = be.classify(frames)
    assert torch.equal(l0, l1) and torch.equal(c0, c1)
    la, ca = be.classify(frames[:128], first_index=0)
    lb, cb = be.classify(frames[128:], first_index=128)
    assert torch.equal(torch.cat([la, lb]), l0) and torch.equal(torch.cat([ca, cb]), c0)
    assert len(set(l0.cpu().tolist())) > 20          # not a degenerate model (mean over 30 samples on clean frames)
    assert lg0.shape == (T, n, 1000) and torch.isfinite(lg0).all()
    assert not torch.equal(lg0[0], lg0[1])            # samples really differ
    be.close()
    be2 = Backend("resnet50", blob, max_batch=n, chunk_a=24, chunk_b=200, regroup_block=10, **kw)
    l2, c2 = be2.classify(frames)
    assert torch.equal(l2, l0) and torch.equal(c2, c0)
    perm = torch.arange(n - 1, -1, -1, device="cuda")
    # reversed batch with the same global indices is not expressible (indices are
    # positional), so check neighbour-independence with dropout off instead
    be2.close()
    be3 = Backend("resnet50", blob, max_batch=n)
    l3, c3 = be3.classify(frames)
    l4, c4 = be3.classify(frames[perm])
    assert torch.equal(l4, l3[perm]) and torch.equal(c4, c3[perm])
    be3.close()


def test_error_behaviour(r18_blob):
    blob, _ = r18_blob
    with pytest.raises(FavError) as e:
        Backend("resnet18_cifar", blob[:1000], max_batch=4)
    assert "BAD_BLOB" in str(e.value)
    bad = bytearray(blob); bad[0] ^= 0xFF
    with pytest.raises(FavError):
        Backend("resnet18_cifar", bytes(bad), max_batch=4)
    with pytest.raises(FavError):                       # resnet50 handle, resnet18 blob
        Backend("resnet50", blob, max_batch=4, in_hw=(64, 64))
    be = Backend("resnet18_cifar", blob, max_batch=4)
    with pytest.raises(FavError) as e:
        be.classify(torch.zeros((5, 32, 32, 3), dtype=torch.uint8, device="cuda"))
    assert "max_batch" in str(e.value)
    with pytest.raises(ValueError):
        be.classify(torch.zeros((2, 16, 32, 3), dtype=torch.uint8, device="cuda"))
    with pytest.raises(TypeError):
        be.classify(torch.zeros((2, 32, 32, 3), dtype=torch.float16, device="cuda"))
    with pytest.raises(FavError):
        Backend("resnet18_cifar", blob, max_batch=4, temperature=0.0)
    # the seam adapter: the dict shape of SignalAnalyzer.analyze_frame
    frame = synth.synthetic_frame_u8(32, 32, 1, 0)
    out = be.analyze_frame(frame)
    assert set(out) == {"anomaly_score", "vision_status", "metrics"}
    assert 0.0 <= out["anomaly_score"] <= 1.0 and out["vision_status"].startswith("VISION_")
    assert abs(out["anomaly_score"] - (1 - out["metrics"]["classifier"]["confidence"])) < 1e-3
    assert {"blur", "brightness", "freeze", "entropy", "raw"} <= set(out["metrics"])      # main.py:171-177 reads these
    assert be.analyze_frame(frame, status_provider=lambda f: "VISION_FROZEN")["vision_status"] == "VISION_FROZEN"
    with pytest.raises(ValueError):                               # wrong size: a caller bug, not a quiet "VISION_OK"
        be.analyze_frame(np.zeros((16, 16, 3), np.uint8))
    be.close()


@pytest.mark.parametrize("arch,hw", [("resnet18_cifar", 32), ("resnet50", 64)])
def test_garbage_fp32_frames_stay_finite_and_match_the_sanitised_oracle(r18_blob, r50_blob, arch, hw):
    """Caller-supplied fp32 frames with NaN / +-Inf / 1e30 pixels: the library treats NaN as 0 and clamps to [-64, 64]
    (fav_sanitize_px, in all three stem kernels: CIFAR im2col, ImageNet fused stem, ViT patches share the im2col), so labels and
    confidences stay finite and equal the oracle's on the sanitised frames bit for bit in the validation mode - the reference's seam
    answers a garbage frame with a status, never an exception (signal_analyzer.py:145-171, video_source.py:76-78)."""
    blob, _ = r18_blob if arch == "resnet18_cifar" else r50_blob
    rng = np.random.default_rng(5)
    x = rng.random((4, hw, hw, 3), dtype=np.float32)
    x[0, 3, 4, 1] = np.nan; x[0, 0, 0, 0] = np.inf; x[1, hw - 1, hw - 1, 2] = -np.inf; x[2, 5, 5, :] = 1e30; x[2, 6, 6, 0] = -1e30
    x[3, :, :, :] = np.nan                                          # a whole frame of NaNs = a black frame
    be = Backend(arch, blob, max_batch=4, in_hw=(hw, hw), math_mode="f32_exact")
    labels, conf = be.classify(torch.from_numpy(x).cuda())
    logits = be.logits().cpu().numpy()
    assert np.isfinite(logits).all() and torch.isfinite(conf).all()
    model = O.parse_blob(blob)
    ol, oc, olg, _ = O.classify(model, O.sanitize_pixels(x), O.ClassifyConfig(exact=True), return_logits=True)
    assert np.array_equal(logits, olg)
    ol2, oc2, olg2, _ = O.classify(model, x, O.ClassifyConfig(exact=True), return_logits=True)     # the oracle sanitises by itself, too
    assert np.array_equal(olg2, olg)
    zero = np.zeros((1, hw, hw, 3), np.float32)
    assert np.array_equal(logits[0, 3], O.classify(model, zero, O.ClassifyConfig(exact=True), return_logits=True)[2][0, 0])
    be.close()
    fast = Backend(arch, blob, max_batch=4, in_hw=(hw, hw))          # production mode: finite as well
    fl, fc = fast.classify(torch.from_numpy(x).cuda())
    assert torch.isfinite(fast.logits()).all() and torch.isfinite(fc).all()
    fast.close()


def test_one_call_scorer_at_the_seam(r50_blob):
    """main.py:160-168 with the drop-in: ONE Backend.analyze_frame call per 240x320 frame returns the rule status of
    the reference's scorer (checked against its CPU restatement) AND the classifier's anomaly score, in the
    reference's dict shape (signal_analyzer.py:128-143); the result drives TrustEngine.update."""
    from failure_aware_vision_amd.trust import TrustEngine
    from oracle.signal_oracle import SignalOracle
    from test_gpu_signal import stream_of_frames
    blob, _ = r50_blob
    frames = stream_of_frames()
    be = Backend("resnet50", blob, in_hw=(240, 320), max_batch=1, n_samples=4, dropout_policy="all_blocks", dropout_p=0.1, seed=4)
    ref_be = Backend("resnet50", blob, in_hw=(240, 320), max_batch=1, n_samples=4, dropout_policy="all_blocks", dropout_p=0.1, seed=4)
    orc, eng, seen = SignalOracle(), TrustEngine(), set()
    for i, fr in enumerate(frames):
        out = be.analyze_frame(fr)
        ref = orc.analyze_frame(fr)
        assert set(out) == {"anomaly_score", "vision_status", "metrics"}
        assert out["vision_status"] == ref["vision_status"], i
        # the keys the reference's loop reads right after the call (main.py:163-177)
        assert abs(out["metrics"]["rule_anomaly_score"] - ref["anomaly_score"]) <= 2e-6
        assert out["metrics"]["raw"]["mean_brightness"] == ref["metrics"]["raw"]["mean_brightness"]
        for key in ("blur", "brightness", "freeze", "entropy"):
            assert abs(out["metrics"][key] - ref["metrics"][key]) <= 1e-4, key
        labels, conf, fail, score = ref_be.classify_detect(fr[None])          # the classifier alone, same frame
        assert out["metrics"]["classifier"]["label"] == int(labels[0]) and out["anomaly_score"] == round(float(score[0]), 6)
        state = eng.update(out["vision_status"], out["anomaly_score"], 1 / 30)
        state["anomaly_score"] = round(out["anomaly_score"], 6)               # main.py:169 (a number, never None)
        seen.add(out["vision_status"])
    assert seen == {"VISION_OK", "VISION_FROZEN", "VISION_BLANK", "VISION_CORRUPTED"}
    assert 0.0 <= state["reliability"] <= 1.0
    be.reset()
    assert be.analyze_frame(frames[0])["metrics"]["raw"]["frame_diff"] == 10.0   # reset cleared the previous frame
    # caller bugs are raised, not swallowed
    with pytest.raises(ValueError):
        be.analyze_frame(frames[0][:100])
    with pytest.raises(TypeError):
        be.analyze_frame(frames[0].astype(np.float32))
    # a failure of the device path does not read as a healthy frame, and leaves the rule scorer's state alone
    be.analyze_frame(frames[1])
    before = be._rules.save_state()
    real = be.classify_detect
    def boom(*a, **k):
        raise FavError(5, "injected")
    be.classify_detect = boom
    bad = be.analyze_frame(frames[2])
    be.classify_detect = real
    assert bad["vision_status"] == "VISION_CORRUPTED" and bad["anomaly_score"] == 1.0 and "injected" in bad["metrics"]["error"]
    assert round(bad["anomaly_score"], 6) == 1.0 and "blur" in bad["metrics"]
    after = be._rules.save_state()
    assert after[0] is before[0] and after[1:] == before[1:]
    eng.update(bad["vision_status"], bad["anomaly_score"], 1 / 30)
    be.close(); ref_be.close()


def test_records_entry_and_calls_on_different_streams(r18_blob):
    """fav_classify_records: the head writes packed (label, confidence) records - the unit the multi-GPU all-gather moves.
    And the handle orders calls issued on DIFFERENT streams on the device (ev_last): a host-buffer classify (the handle's
    own stream) or a call on a second torch stream, issued while a call on the first stream is still running, neither
    corrupts that call's result nor reads its activations half written."""
    from failure_aware_vision_amd import classify_sharded
    blob, _ = r18_blob
    n = 64
    frames_np = synth.synthetic_frames_u8(n, 32, 32, seed=13)
    frames = torch.from_numpy(frames_np).cuda()
    kw = dict(n_samples=8, dropout_policy="all_blocks", dropout_p=0.1, seed=4)
    be = Backend("resnet18_cifar", blob, max_batch=n, **kw)
    l0, c0 = be.classify(frames)
    rec = be.classify_records(frames)
    assert rec.dtype == torch.int32 and tuple(rec.shape) == (n, 2)
    assert torch.equal(rec[:, 0], l0) and torch.equal(rec[:, 1].contiguous().view(torch.float32), c0)
    slot = torch.full((n + 4, 2), -7, dtype=torch.int32, device="cuda")            # a slot inside a larger send buffer
    be.classify_records(frames[:10], first_index=0, out=slot[2:12])
    assert torch.equal(slot[2:12, 0], l0[:10]) and int((slot[:2] != -7).sum()) == 0 and int((slot[12:] != -7).sum()) == 0
    ls, cs = classify_sharded(be, frames, n, 0, 1)                                  # world 1: views of the send buffer
    assert torch.equal(ls, l0) and torch.equal(cs, c0)
    with pytest.raises(ValueError):
        be.classify_records(frames, out=torch.empty((n, 2), dtype=torch.int64, device="cuda"))
    # two streams, no host synchronisation in between
    half = frames_np[:32]
    lh_ref, ch_ref = be.classify(half)                       # numpy in -> numpy out
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for _ in range(5):
        la, ca = be.classify(frames)                          # current stream, asynchronous
        lb, cb = be.classify(half)                            # numpy in: the handle's own stream, synchronous
        with torch.cuda.stream(side):
            lc, cc = be.classify(frames)                      # a second torch stream
        torch.cuda.synchronize()
        assert np.array_equal(lb, lh_ref) and np.array_equal(cb, ch_ref)
        assert torch.equal(la, l0) and torch.equal(ca, c0) and torch.equal(lc, l0) and torch.equal(cc, c0)
    be.close()


def test_full_size_properties_vit_and_ensemble(r50_members):
    """The two 8-GPU configs of BASELINE.json at their real sizes, production mode, no oracle: determinism, shard
    invariance (what the multi-GPU path relies on) and neighbour independence.  configs[4]: ViT-B/16 at its global batch
    512 and at the per-GPU share 64 (32 + 32 shards; the two-stream split inside a call is on); configs[3]: the 5-member
    ensemble at its global batch 256 (128 + 128 and a ragged 100 + 156 split)."""
    vblob, _ = weights.make_synthetic_vit("vit_b16", seed=1)
    x = torch.from_numpy(synth.synthetic_frames_u8(512, 224, 224, seed=21)).cuda()
    vit = Backend("vit_b16", vblob, max_batch=512, temperature=1.5, conf_kind="entropy")
    l0, c0 = vit.classify(x)
    lg0 = vit.logits()
    l1, c1 = vit.classify(x)
    assert torch.equal(l0, l1) and torch.equal(c0, c1) and torch.isfinite(lg0).all()
    for cut in (256, 200):
        la, ca = vit.classify(x[:cut], first_index=0)
        lb, cb = vit.classify(x[cut:], first_index=cut)
        assert torch.equal(torch.cat([la, lb]), l0) and torch.equal(torch.cat([ca, cb]), c0), cut
    la, ca = vit.classify(x[:64])
    lb, cb = vit.classify(x[:32]); lc, cc = vit.classify(x[32:64])
    assert torch.equal(la, l0[:64]) and torch.equal(torch.cat([lb, lc]), la) and torch.equal(torch.cat([cb, cc]), ca)
    perm = torch.randperm(512, generator=torch.Generator().manual_seed(1)).cuda()
    lp, cp = vit.classify(x[perm])
    assert torch.equal(lp, l0[perm]) and torch.equal(cp, c0[perm])
    assert len(set(l0.cpu().tolist())) > 20
    vit.close()
    ens = Backend("resnet50", [b for b, _ in r50_members], max_batch=256)
    x = x[:256]
    l0, c0 = ens.classify(x)
    l1, c1 = ens.classify(x)
    assert torch.equal(l0, l1) and torch.equal(c0, c1)
    for cut in (128, 100):
        la, ca = ens.classify(x[:cut], first_index=0)
        lb, cb = ens.classify(x[cut:], first_index=cut)
        assert torch.equal(torch.cat([la, lb]), l0) and torch.equal(torch.cat([ca, cb]), c0), cut
    perm = torch.randperm(256, generator=torch.Generator().manual_seed(2)).cuda()
    lp, cp = ens.classify(x[perm])
    assert torch.equal(lp, l0[perm]) and torch.equal(cp, c0[perm])
    assert len(set(l0.cpu().tolist())) > 20
    ens.close()
