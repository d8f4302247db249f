"""On-device corruption generator vs its CPU restatement (oracle/corrupt_oracle.py): the uint8
modes are order-free fp32 arithmetic on Philox draws and must be bit-exact; Gaussian noise goes
through logf/sqrtf/cosf and is held to 2e-6."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from failure_aware_vision_amd import synth  # noqa: E402
from failure_aware_vision_amd.corrupt import Corruptor  # noqa: E402
from failure_aware_vision_amd.signal import SignalAnalyzerHIP  # noqa: E402
from oracle import corrupt_oracle as CO  # noqa: E402


def test_modes_bit_exact_and_statuses_detected():
    frames = synth.synthetic_frames_u8(6, 240, 320, seed=2)
    dev = torch.from_numpy(frames).cuda()
    c = Corruptor(seed=0xABCDEF0123)
    c.set_noise(0.4); c.set_brightness(0.6)
    out = c.apply(dev[:3]).cpu().numpy()
    assert np.array_equal(out, CO.corrupt(frames[:3], 0, 0.4, np.float32(0.6 / 0.5), 0, c.seed, 0))
    assert c.get_vision_status() == "VISION_OK"
    c.set_mode("corrupted")
    glitch = c.apply(dev[3:5]).cpu().numpy()
    assert np.array_equal(glitch, CO.corrupt(frames[3:5], 2, 0.4, np.float32(0.6 / 0.5), 0, c.seed, 3))
    assert 0.15 < (glitch[..., 1] == 0).mean() < 0.35          # ~20 % glitched pixels + bars
    c.set_mode("frozen")
    frozen = c.apply(dev[5:6]).cpu().numpy()
    assert np.array_equal(frozen[0], glitch[-1]) and c.get_vision_status() == "VISION_FROZEN"
    c.set_mode("blank")
    blank = c.apply(dev[5:6]).cpu().numpy()
    assert np.array_equal(blank, CO.corrupt(frames[5:6], 1, 0, 1, 0, c.seed, 6)) and blank.max() == 4
    # the rule-based scorer sees what the modes are meant to provoke
    an = SignalAnalyzerHIP()
    assert an.analyze_frame(blank[0])["vision_status"] == "VISION_BLANK"
    for _ in range(6):
        r = an.analyze_frame(glitch[-1])
    assert r["vision_status"] == "VISION_FROZEN"


def test_gaussian_noise_matches_restatement_and_has_the_right_sigma():
    frames = synth.synthetic_frames_u8(2, 224, 224, seed=4)
    c = Corruptor(seed=77)
    out = c.gaussian(torch.from_numpy(frames).cuda(), severity=3, first_index=1000).cpu().numpy()
    ref = CO.corrupt(frames, 3, 0, 1, 0.18, 77, 1000)
    assert np.abs(out - ref).max() < 2e-6
    mid = (frames > 60) & (frames < 195)                        # unclipped pixels
    resid = (out - frames.astype(np.float32) / 255.0)[mid]
    assert abs(resid.std() - 0.18) < 0.005 and abs(resid.mean()) < 0.002
