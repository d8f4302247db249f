"""Wire / CSV formats of the reference's reporting side, so
the existing dashboard can display this backend unchanged.  Host code; mirrors

* ``SessionLogger``  — platform/backend/session_logger.py:12-55 (11-column CSV, one row per tick)
* ``FailureAttributor`` — platform/backend/failure_attributor.py:13-121 (excursions below 0.7)
* the per-tick message assembled at main.py:146-149,171-182,192-193

and is pinned by fixtures generated from the reference (tests/golden/reporting.json).
"""
from __future__ import annotations

import csv
import io
import time

CSV_COLUMNS = ("timestamp", "reliability", "policy_state", "anomaly", "anomaly_integral", "vision_status",
               "trust_velocity", "recovery_debt", "recovery_coeff", "contradiction_detected", "contradiction_count")
EVENT_COLUMNS = ("start_time", "duration_s", "min_reliability", "cause", "recovery_time_s")
EXCURSION_LEVEL = 0.7
_RANK = {"NONE": 0, "ML_ANOMALY": 1, "FROZEN": 2, "BLANK": 3, "CORRUPTED": 4}   # higher dominates


class SessionLogger:
    HEADER = list(CSV_COLUMNS)

    def __init__(self):
        self.reset()

    def reset(self):
        self._rows = io.StringIO()
        self._csv = csv.writer(self._rows)
        self._csv.writerow(self.HEADER)
        self.entry_count = 0

    def log(self, state: dict, anomaly_score: float):
        g = state.get
        self._csv.writerow([
            "%.6f" % g("timestamp", time.time()), "%.6f" % g("reliability", 0), g("policy_state", ""),
            "%.6f" % anomaly_score, "%.6f" % g("anomaly_integral", 0), g("vision_status", ""),
            "%.6f" % g("trust_velocity", 0), "%.4f" % g("recovery_debt", 0), "%.4f" % g("recovery_coeff", 0.10),
            g("contradiction_detected", False), g("contradiction_count", 0)])
        self.entry_count += 1

    def get_csv(self) -> str:
        return self._rows.getvalue()


def tick_cause(state: dict) -> str:
    status = state["vision_status"]
    if status.startswith("VISION_") and status[7:] in ("FROZEN", "BLANK", "CORRUPTED"):
        return status[7:]
    if state["ml_influence_active"] and state.get("anomaly_integral", 0) > 0.5:
        return "ML_ANOMALY"
    return "NONE"


class FailureAttributor:
    CAUSE_PRIORITY = dict(_RANK)

    def __init__(self):
        self.reset()

    def reset(self):
        self._events = []
        self._open = None          # [start, min_reliability, cause] while below the excursion level

    def update(self, state: dict, timestamp: float):
        r, cause = state["reliability"], tick_cause(state)
        if r < EXCURSION_LEVEL:
            if self._open is None:
                self._open = [timestamp, r, cause]
            else:
                self._open[1] = min(self._open[1], r)
                if _RANK.get(cause, 0) > _RANK.get(self._open[2], 0):
                    self._open[2] = cause
        elif self._open is not None:
            start, lowest, why = self._open
            span = round(timestamp - start, 3)       # the reference reports the same span twice
            self._events.append({"start_time": round(start, 3), "duration_s": span, "min_reliability": round(lowest, 4),
                                 "cause": why, "recovery_time_s": span})
            self._open = None

    def get_events(self) -> list:
        return list(self._events)

    def get_summary(self) -> dict:
        if not self._events:
            return {"total_excursions": 0}
        causes = [e["cause"] for e in self._events]
        return {"total_excursions": len(causes), "by_cause": {c: causes.count(c) for c in set(causes)},
                "mean_recovery_s": round(sum(e["recovery_time_s"] for e in self._events) / len(causes), 3),
                "worst_reliability": round(min(e["min_reliability"] for e in self._events), 4)}

    def get_events_csv(self) -> str:
        buf = io.StringIO()
        w = csv.writer(buf)
        w.writerow(EVENT_COLUMNS)
        for e in self._events:
            w.writerow([e[c] for c in EVENT_COLUMNS])
        return buf.getvalue()


def tick_message(state: dict, analysis: dict, dt: float, source_mode: str, attributor: FailureAttributor) -> dict:
    """The live-mode per-tick message of main.py:165-193 for a scorer result ``analysis``
    (the dict Backend.analyze_frame / SignalAnalyzerHIP.analyze_frame return)."""
    msg = dict(state)
    score = analysis["anomaly_score"]
    msg["anomaly_score"] = round(score, 6) if score is not None else 0.0
    msg["dt"] = round(dt, 6)
    m = analysis.get("metrics", {})
    msg["frame"] = {"mode": source_mode, "noise_level": m.get("blur", 0.0), "brightness": 1.0 - m.get("brightness", 0.0),
                    "vision_status": analysis["vision_status"]}
    msg["source_mode"] = source_mode
    msg["signal_metrics"] = m
    attributor.update(msg, msg["timestamp"])
    msg["failure_events"] = attributor.get_summary()
    return msg
