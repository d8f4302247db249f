"""Trust-engine consumer of the classifier's anomaly score.

Host-side mirror of the reference's ``TrustEngine`` so the (label, confidence) path
can be wired to the same downstream contract end to end:
``engine.update(vision_status, anomaly_score, dt) -> state dict``
(platform/backend/trust_engine.py:139-243, consumed at main.py:145,168,348).

Behaviour is pinned by the reference's own smoke script (test_trust.py:1-33) and by
per-tick fixtures generated from the reference (tests/golden/trust_engine.json).
It is O(1) scalar arithmetic per tick — host code, not a kernel.  Quirks of the
reference that callers rely on are kept: the first call and
every status change return WITHOUT integrating a tick, and ``anomaly_score`` in the
returned state is always 0.0 (callers overwrite it, main.py:146,169).
"""
from __future__ import annotations

import statistics
import time
from collections import deque

OK, FROZEN, BLANK, CORRUPTED = "VISION_OK", "VISION_FROZEN", "VISION_BLANK", "VISION_CORRUPTED"

# reliability change per second while a failure status holds (trust_engine.py:21-26,207,215,223);
# VISION_OK's entry is the nominal recovery rate shown to the UI (sign convention of the reference)
_RATE = {OK: -0.10, FROZEN: 0.30, BLANK: 0.60, CORRUPTED: 1.00}

DEBT_CAP, DEBT_GAIN, COEFF_FLOOR, DEBT_DRAIN = 10.0, 0.008, 0.03, 0.10   # trust_engine.py:28-32
ML_GAIN, ML_LEAK = 0.15, 0.5                                             # trust_engine.py:47-48
VELOCITY_ALPHA = 0.12                                                    # trust_engine.py:53
WINDOW, WINDOW_MIN, STATUS_MIN, Z_LIMIT, STD_FLOOR = 60, 30, 10, 3.0, 0.001  # trust_engine.py:61,108,115,126,129


def policy_for(reliability: float, velocity: float) -> str:
    """Policy gate (trust_engine.py:79-86)."""
    if reliability >= 0.7:
        return "VISION_DECLINING" if velocity < -0.15 else "VISION_ALLOWED"
    return "VISION_DEGRADED" if reliability >= 0.3 else "VISION_BLOCKED"


class TrustEngine:
    def __init__(self):
        self.reset()

    def reset(self):
        self.reliability = 1.0
        self.policy_state = "VISION_ALLOWED"
        self.anomaly_integral = 0.0
        self.current_status = None
        self.trust_velocity = 0.0
        self.recovery_debt = 0.0
        self.recovery_coeff = 0.10
        self.contradiction_detected = False
        self.contradiction_count = 0
        self._last_reliability = 1.0
        self._window = deque(maxlen=WINDOW)
        self._ticks = 0

    # -- one tick ---------------------------------------------------------------
    def update(self, vision_status: str, anomaly_score, dt: float) -> dict:
        self._ticks += 1
        previous = self.current_status
        if previous is None or vision_status != previous:
            # first sample / status edge: latch the status, no dynamics this tick
            self.current_status = vision_status
            if previous == OK and vision_status != OK:
                self.anomaly_integral = 0.0
            self.policy_state = policy_for(self.reliability, self.trust_velocity)
            return self.get_state()

        if vision_status == OK:
            self._recover(anomaly_score, dt)
        elif vision_status in _RATE:
            self._decay(_RATE[vision_status], dt)

        self.reliability = min(1.0, max(0.0, self.reliability))
        slope = (self.reliability - self._last_reliability) / max(dt, 0.001)
        self.trust_velocity = VELOCITY_ALPHA * slope + (1 - VELOCITY_ALPHA) * self.trust_velocity
        self._last_reliability = self.reliability
        self._check_contradiction(vision_status, anomaly_score)
        self.policy_state = policy_for(self.reliability, self.trust_velocity)
        return self.get_state()

    def _recover(self, score, dt):
        self.recovery_debt = max(0.0, self.recovery_debt - DEBT_DRAIN * dt)
        self.recovery_coeff = max(COEFF_FLOOR, 0.10 - DEBT_GAIN * self.recovery_debt)
        self.reliability += self.recovery_coeff * dt
        if score is not None:  # ML is a penalty-only sensor, active only under VISION_OK
            self.anomaly_integral += score * dt
            self.anomaly_integral -= ML_LEAK * self.anomaly_integral * dt
            self.anomaly_integral = max(0.0, self.anomaly_integral)
            self.reliability -= ML_GAIN * self.anomaly_integral * dt

    def _decay(self, rate, dt):
        self.recovery_debt = min(DEBT_CAP, self.recovery_debt + max(0.0, 0.7 - self.reliability) * dt)
        self.reliability -= rate * dt
        self.anomaly_integral = 0.0

    def _check_contradiction(self, status, score):
        if score is None:
            self.contradiction_detected = False
            return
        self._window.append((status, score))
        same = [s for st, s in self._window if st == status] if len(self._window) >= WINDOW_MIN else []
        if len(same) < STATUS_MIN:
            self.contradiction_detected = False
            return
        spread = max(statistics.stdev(same), STD_FLOOR)
        outlier = status == OK and (score - statistics.mean(same)) / spread > Z_LIMIT
        if outlier and not self.contradiction_detected:
            self.contradiction_count += 1
        self.contradiction_detected = outlier

    # -- snapshot (trust_engine.py:245-263) -----------------------------------------
    def get_state(self) -> dict:
        status = self.current_status
        return {
            "timestamp": time.time(),
            "reliability": round(self.reliability, 6),
            "policy_state": self.policy_state,
            "vision_status": status or "UNKNOWN",
            "anomaly_score": 0.0,
            "anomaly_integral": round(self.anomaly_integral, 6),
            "trust_velocity": round(self.trust_velocity, 6),
            "recovery_debt": round(self.recovery_debt, 4),
            "recovery_coeff": round(self.recovery_coeff, 4),
            "contradiction_detected": self.contradiction_detected,
            "contradiction_count": self.contradiction_count,
            "ml_influence_active": status == OK,
            "decay_coefficient": _RATE.get(status or OK, 0),
            "recovery_coefficient": round(self.recovery_coeff, 4),
            "tick_count": self._ticks,
        }


def drive(engine: TrustEngine, scorer, frames, dt: float, status_provider=None):
    """The reference's live-mode tick (main.py:153-170) with the classifier as scorer:
    each frame -> scorer.analyze_frame -> engine.update; returns the per-tick states
    with ``anomaly_score`` filled in the way main.py:169 does."""
    out = []
    for frame in frames:
        a = scorer.analyze_frame(frame, status_provider) if status_provider else scorer.analyze_frame(frame)
        state = engine.update(a["vision_status"], a["anomaly_score"], dt)
        if a["anomaly_score"] is not None:
            state["anomaly_score"] = round(a["anomaly_score"], 6)
        out.append(state)
    return out
