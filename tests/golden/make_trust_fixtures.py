"""Generates tests/golden/trust_engine.json by RUNNING the reference's stdlib-only
modules from /root/reference (read-only; available in the build container only).
The JSON holds inputs and the reference's outputs — data, not source.

  python tests/golden/make_trust_fixtures.py
"""
import json
import os
import sys

REF = "/root/reference/platform/backend"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
from trust_engine import TrustEngine  # noqa: E402
from anomaly_simulator import AnomalySimulator  # noqa: E402

DET = ["reliability", "policy_state", "vision_status", "anomaly_integral", "trust_velocity", "recovery_debt",
       "recovery_coeff", "contradiction_detected", "contradiction_count", "ml_influence_active",
       "decay_coefficient", "recovery_coefficient", "tick_count"]


def det(state):
    return {k: state[k] for k in DET}


out = {}

# 1. the reference's own smoke script, platform/backend/test_trust.py:1-33
e = TrustEngine()
dt = 0.033
seq = [("VISION_OK", 0.019, 1), ("VISION_FROZEN", 0.019, 50), ("VISION_BLANK", None, 30),
       ("VISION_CORRUPTED", None, 100), ("VISION_OK", 0.019, 200)]
steps, after = [], []
for status, score, n in seq:
    for _ in range(n):
        s = e.update(status, score, dt)
        steps.append(det(s))
    after.append(det(s))
out["test_trust"] = {"dt": dt, "segments": [[a, b, c] for a, b, c in seq], "after_each_segment": after, "every_tick": steps}

# 2. playground preset full_cycle replayed as main.py:338-352 does (engine + AnomalySimulator(seed=99), dt=1/30)
events = [("VISION_OK", 0.0, 0.5, 60), ("VISION_FROZEN", 0.0, 0.5, 60), ("VISION_OK", 0.0, 0.5, 60),
          ("VISION_BLANK", 0.0, 0.5, 60), ("VISION_OK", 0.0, 0.5, 60), ("VISION_CORRUPTED", 0.6, 0.5, 60),
          ("VISION_OK", 0.0, 0.5, 120)]
e = TrustEngine()
sim = AnomalySimulator(seed=99)
ticks, scores = [], []
for status, noise, bright, frames in events:
    for _ in range(frames):
        sc = sim.compute_anomaly(noise, bright, status)
        s = e.update(status, sc, 1.0 / 30.0)
        ticks.append(det(s))
        scores.append(sc)
out["full_cycle"] = {"dt": 1.0 / 30.0, "events": [list(ev) for ev in events], "scores": scores, "every_tick": ticks}

# 3. contradiction detector: an outlier score under VISION_OK after a quiet baseline
e = TrustEngine()
ticks, scores = [], []
for i in range(80):
    sc = 0.019 + 0.0005 * ((i * 7919) % 11 - 5) / 5.0
    if i in (60, 61, 70):
        sc = 0.2
    scores.append(sc)
    ticks.append(det(e.update("VISION_OK", sc, 1.0 / 30.0)))
out["contradiction"] = {"dt": 1.0 / 30.0, "scores": scores, "every_tick": ticks}

# 4. None scores and a long outage (debt cap) 
e = TrustEngine()
ticks = []
plan = [("VISION_OK", None, 5), ("VISION_CORRUPTED", None, 700), ("VISION_OK", 0.5, 400)]
for status, sc, n in plan:
    for _ in range(n):
        ticks.append(det(e.update(status, sc, 0.05)))
out["outage"] = {"dt": 0.05, "plan": [list(p) for p in plan], "every_tick": ticks[::10], "last": ticks[-1]}

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "trust_engine.json")
json.dump(out, open(path, "w"))
print("wrote", path, os.path.getsize(path), "bytes")
for a in out["test_trust"]["after_each_segment"]:
    print(a["reliability"], a["policy_state"], a["anomaly_integral"])
