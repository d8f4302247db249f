"""BASELINE configs[3] at its real shape: a deep ensemble of FIVE independently seeded ResNet-50 members on
224x224 Gaussian-noise severity-3 frames, head = mean over members of softmax (the reference has no
counterpart; the slot it feeds is platform/backend/main.py:160).

* production mode (bf16 MFMA) and validation mode (fp32 chain): the logits [5][n][1000] of every member are
  bit-equal to the CPU oracle's, run live here on the same seeded frames;
* production mode against the committed fixture tests/golden/r50_ens5_mfma_256.npz - the config's global batch of 256 frames - (per member and frame CRC-32
  of the 1000 logits; generator: tests/golden/make_classifier_fixtures.py ens5);
* at the per-GPU share of the 8-GPU configuration (32 frames per call): member independence (member m's logits
  equal a single-model handle loaded with checkpoint m), shard invariance and determinism.
"""
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from failure_aware_vision_amd import Backend, synth  # noqa: E402
from oracle import fav_oracle as O  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FRAME_SEED, NOISE_SEED, SEVERITY = 21, 3, 3


def frames_np(start, n):
    u8 = synth.synthetic_frames_u8(n, 224, 224, seed=FRAME_SEED, start_id=start)
    return synth.gaussian_noise_f32(u8, SEVERITY, seed=NOISE_SEED, start_id=start)


@pytest.mark.parametrize("mode,exact", [("bf16", "mfma"), ("f32_exact", True)])
def test_ens5_resnet50_224_logits_bitwise_vs_oracle(r50_members, mode, exact):
    n = 8
    x = frames_np(0, n)
    be = Backend("resnet50", [b for b, _ in r50_members], max_batch=n, math_mode=mode)
    labels, conf = be.classify(torch.from_numpy(x).cuda())
    lg = be.logits().cpu().numpy()
    be.close()
    assert lg.shape == (5, n, 1000)
    cfg = O.ClassifyConfig(exact=exact)
    olg = np.stack([O.classify(O.parse_blob(b), x, cfg, return_logits=True)[2][0] for b, _ in r50_members])
    for m in range(5):
        assert np.array_equal(lg[m], olg[m]), f"member {m}: {np.mean(lg[m] != olg[m]):.4f} of logits differ ({mode})"
    ol, oc, pb = O.confidence_head(olg)
    srt = np.sort(pb, axis=1)
    tie = (srt[:, -1] - srt[:, -2]) < 1e-6
    assert np.array_equal(labels.cpu().numpy()[~tie], ol[~tie])
    np.testing.assert_allclose(conf.cpu().numpy(), oc, rtol=0, atol=3e-6)
    assert not np.array_equal(olg[0], olg[1])          # the members really are different networks


def test_ens5_production_fixture(r50_members):
    path = os.path.join(GOLD, "r50_ens5_mfma_256.npz")
    if not os.path.exists(path):
        pytest.skip("r50_ens5_mfma_256.npz not generated yet")
    d = np.load(path)
    assert [i["sha256"] for _, i in r50_members] == [str(s) for s in d["member_blob_sha256"]], "different checkpoints"
    n = len(d["labels"])
    be = Backend("resnet50", [b for b, _ in r50_members], max_batch=n)
    labels, conf = be.classify(torch.from_numpy(frames_np(0, n)).cuda())
    lg = be.logits().cpu().numpy()
    be.close()
    crc = np.array([[zlib.crc32(np.ascontiguousarray(lg[m, i]).tobytes()) for i in range(n)] for m in range(5)], np.uint32)
    assert np.array_equal(crc, d["member_logit_crc32"])
    tie = d["gap"] < 1e-6
    assert np.array_equal(labels.cpu().numpy()[~tie], d["labels"].astype(np.int32)[~tie])
    np.testing.assert_allclose(conf.cpu().numpy(), d["conf"], rtol=0, atol=3e-6)


def test_ens5_per_gpu_share_properties(r50_members):
    """32 frames per call = BASELINE configs[3]'s batch 256 sharded over 8 GPUs."""
    n = 32
    x = torch.from_numpy(frames_np(100, n)).cuda()
    blobs = [b for b, _ in r50_members]
    be = Backend("resnet50", blobs, max_batch=n)
    l0, c0 = be.classify(x, first_index=100)
    lg = be.logits().clone()
    l1, c1 = be.classify(x, first_index=100)
    assert torch.equal(l0, l1) and torch.equal(c0, c1)                       # determinism
    la, ca = be.classify(x[:13], first_index=100)
    lb, cb = be.classify(x[13:], first_index=113)
    assert torch.equal(torch.cat([la, lb]), l0) and torch.equal(torch.cat([ca, cb]), c0)   # shard invariance (ragged split)
    be.close()
    for m in (0, 3):                                                         # member independence
        single = Backend("resnet50", blobs[m], max_batch=n)
        single.classify(x)
        assert torch.equal(single.logits()[0], lg[m]), m
        single.close()
    assert lg.shape == (5, n, 1000) and torch.isfinite(lg).all()


@pytest.mark.parametrize("n", [5, 32, 70])
def test_ens5_grouped_launches_equal_member_streams(r50_members, n, monkeypatch):
    """The two ways a 5-member call runs - every op ONE launch over all members (block row = member; the default for every
    call) and one stream per member (fav_config.ens_grouped_max = -1, or calls beyond a positive limit) - give the same logits
    bit for bit, labels and confidences included."""
    x = torch.from_numpy(frames_np(300, n)).cuda()
    blobs = [b for b, _ in r50_members]
    out = {}
    for limit in ("0", "4096"):
        be = Backend("resnet50", blobs, max_batch=n, ens_grouped_max=-1 if limit == "0" else int(limit))
        labels, conf = be.classify(x, first_index=300)
        out[limit] = (labels.clone(), conf.clone(), be.logits().clone())
        if limit == "4096":            # a second, smaller call on the same handle (strides follow the call's n)
            l2, c2 = be.classify(x[:3], first_index=300)
            assert torch.equal(l2, labels[:3]) and torch.equal(c2, conf[:3])
        be.close()
    for a, b in zip(out["0"], out["4096"]):
        assert torch.equal(a, b)
    assert not torch.equal(out["0"][2][0], out["0"][2][1])
