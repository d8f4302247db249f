"""Generates tests/golden/reporting.json by running the reference's stdlib-only
SessionLogger / FailureAttributor / TrustEngine (read-only, build container only)."""
import json, os, sys
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/platform/backend")
from trust_engine import TrustEngine
from session_logger import SessionLogger
from failure_attributor import FailureAttributor

plan = [("VISION_OK", 0.02, 40), ("VISION_FROZEN", 0.02, 70), ("VISION_OK", 0.02, 300), ("VISION_CORRUPTED", None, 30),
        ("VISION_BLANK", None, 20), ("VISION_OK", 0.9, 200), ("VISION_OK", 0.01, 900), ("VISION_OK", 4.5, 160),
        ("VISION_OK", 0.01, 500)]
e, log, att = TrustEngine(), SessionLogger(), FailureAttributor()
dt, t = 1 / 30, 1000.0
states, summaries = [], []
for status, score, n in plan:
    for _ in range(n):
        s = e.update(status, score, dt)
        t += dt
        s["timestamp"] = round(t, 6)
        s["anomaly_score"] = round(score, 6) if score is not None else 0.0
        att.update(s, s["timestamp"])
        log.log(s, s.get("anomaly_score", 0))
        states.append({k: s[k] for k in ("timestamp", "reliability", "policy_state", "vision_status", "anomaly_score",
                                         "anomaly_integral", "trust_velocity", "recovery_debt", "recovery_coeff",
                                         "contradiction_detected", "contradiction_count", "ml_influence_active")})
        summaries.append(att.get_summary())
out = {"plan": plan, "dt": dt, "t0": 1000.0, "states": states, "csv": log.get_csv(), "entries": log.entry_count,
       "events": att.get_events(), "events_csv": att.get_events_csv(), "summary": att.get_summary(),
       "summary_every_100": summaries[::100]}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reporting.json")
json.dump(out, open(path, "w"))
print(len(states), "ticks;", att.get_summary())
