"""The fused signal-statistics kernel and the SignalAnalyzer mirror against the CPU
restatement of the reference's scorer (oracle/signal_oracle.py; parity unpinned, cv2 absent)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from failure_aware_vision_amd import synth  # noqa: E402
from failure_aware_vision_amd.signal import SignalAnalyzerHIP  # noqa: E402
from failure_aware_vision_amd.trust import TrustEngine  # noqa: E402
from oracle.signal_oracle import SignalOracle  # noqa: E402


def stream_of_frames():
    """A 320x240 stream with every status the rules can produce: normal, frozen x7, dark, bright, noise."""
    rng = np.random.default_rng(0)
    base = [synth.synthetic_frame_u8(240, 320, 5, i) for i in range(6)]
    noisy = [np.clip(f.astype(np.int16) + rng.integers(-20, 21, f.shape), 0, 255).astype(np.uint8) for f in base]
    frames = noisy[:3] + [noisy[2]] * 7 + noisy[3:5]
    frames += [np.full((240, 320, 3), 4, np.uint8), np.full((240, 320, 3), 252, np.uint8)]
    frames += [rng.integers(0, 256, (240, 320, 3), dtype=np.uint8) for _ in range(2)] + noisy[5:]
    return np.stack(frames)


def test_stats_integers_exact_and_floats_close():
    frames = stream_of_frames()
    an, orc = SignalAnalyzerHIP(), SignalOracle()
    stats = an.stats(frames[:9]) + an.stats(frames[9:])      # two calls: prev gray carried across calls
    for i, (st, fr) in enumerate(zip(stats, frames)):
        gray, lap_var, mean, mean_diff, entropy, hist = orc.raw(fr)
        assert np.array_equal(np.array(st.hist[:]), hist), i
        assert st.sum_gray == int(gray.astype(np.int64).sum())
        assert st.has_prev == (1 if i > 0 else 0)
        if i > 0:
            assert abs(st.mean_diff - mean_diff) < 1e-12
        assert abs(st.mean - mean) < 1e-12
        assert abs(st.lap_var - lap_var) <= 1e-9 * max(1.0, lap_var)
        assert abs(st.entropy - entropy) < 2e-5


def test_analyze_frame_dicts_match_the_restated_scorer():
    frames = stream_of_frames()
    an, orc = SignalAnalyzerHIP(), SignalOracle()
    got = an.analyze_frames(frames[:5]) + [an.analyze_frame(f) for f in frames[5:]]
    seen = set()
    for i, (g, fr) in enumerate(zip(got, frames)):
        ref = orc.analyze_frame(fr)
        assert g["vision_status"] == ref["vision_status"], i
        assert abs(g["anomaly_score"] - ref["anomaly_score"]) <= 2e-6
        for k in ("blur", "brightness", "freeze", "entropy"):
            assert abs(g["metrics"][k] - ref["metrics"][k]) <= 1e-4
        for k in ("laplacian_var", "mean_brightness", "frame_diff"):
            assert g["metrics"]["raw"][k] == ref["metrics"]["raw"][k], (i, k)
        seen.add(g["vision_status"])
    assert seen == {"VISION_OK", "VISION_FROZEN", "VISION_BLANK", "VISION_CORRUPTED"}
    # the scorer feeds the trust engine exactly like main.py:160-168
    e = TrustEngine()
    for g in got:
        s = e.update(g["vision_status"], g["anomaly_score"], 1 / 30)
    assert 0.0 <= s["reliability"] <= 1.0
    an.reset()
    assert an.analyze_frame(frames[0])["metrics"]["raw"]["frame_diff"] == 10.0   # first-frame placeholder
