"""Tracker-compatible experiment output: the `_data.json` schema of the reference's
ExperimentTracker.save_data_for_plotting (src/experiments/experiment_tracker.py:82-158), so that the reference's
plot_mccfr.py works on results produced by this engine.  Pure host code; no pickle, no plotting."""
import json
from dataclasses import dataclass, field
from pathlib import Path
from typing import List, Optional

import numpy as np


@dataclass
class ExperimentMetrics:
    """Fields of the reference's ExperimentMetrics that reach the JSON (experiment_tracker.py:13-56)."""
    iterations: List[int] = field(default_factory=list)
    algorithm: str = "Unknown"
    eval_iterations: List[int] = field(default_factory=list)
    eval_rewards: List[float] = field(default_factory=list)
    eval_scopas_trained: List[float] = field(default_factory=list)
    eval_scopas_random: List[float] = field(default_factory=list)
    eval_scopa_diff: List[float] = field(default_factory=list)
    final_reward: float = 0.0
    final_scopa_trained: float = 0.0
    final_scopa_random: float = 0.0
    final_scopa_diff: float = 0.0
    num_info_sets: int = 0
    exploitability_iterations: Optional[List[int]] = None
    exploitability_values: Optional[List[float]] = None


class ExperimentTracker:
    def __init__(self, experiment_name, save_dir="experiments/results"):
        self.experiment_name = experiment_name
        self.save_dir = Path(save_dir)
        self.runs: List[ExperimentMetrics] = []

    def add_run(self, metrics: ExperimentMetrics):
        self.runs.append(metrics)

    def plot_data(self):
        data = {"experiment_name": self.experiment_name, "algorithm": self.runs[0].algorithm if self.runs else "Unknown",
                "num_runs": len(self.runs), "runs": []}
        for i, run in enumerate(self.runs):
            rd = {"run_id": i + 1, "eval_iterations": run.eval_iterations, "eval_rewards": run.eval_rewards,
                  "eval_scopas_trained": run.eval_scopas_trained, "eval_scopas_random": run.eval_scopas_random,
                  "eval_scopa_diff": run.eval_scopa_diff, "final_reward": run.final_reward,
                  "final_scopa_trained": run.final_scopa_trained, "final_scopa_random": run.final_scopa_random,
                  "final_scopa_diff": run.final_scopa_diff, "num_info_sets": run.num_info_sets}
            if run.exploitability_iterations:
                rd["exploitability_iterations"] = run.exploitability_iterations
                rd["exploitability_values"] = run.exploitability_values
            data["runs"].append(rd)
        if len(self.runs) > 1:
            arr = lambda name: np.array([getattr(r, name) for r in self.runs])
            rw, st, sr, sd = arr("eval_rewards"), arr("eval_scopas_trained"), arr("eval_scopas_random"), arr("eval_scopa_diff")
            fin = lambda name: [getattr(r, name) for r in self.runs]
            data["statistics"] = {
                "eval_iterations": self.runs[0].eval_iterations,
                "rewards": {"mean": rw.mean(0).tolist(), "std": rw.std(0).tolist(), "min": rw.min(0).tolist(), "max": rw.max(0).tolist()},
                "scopas_trained": {"mean": st.mean(0).tolist(), "std": st.std(0).tolist()},
                "scopas_random": {"mean": sr.mean(0).tolist(), "std": sr.std(0).tolist()},
                "scopa_diff": {"mean": sd.mean(0).tolist(), "std": sd.std(0).tolist()},
                "final_metrics": {"reward_mean": float(np.mean(fin("final_reward"))), "reward_std": float(np.std(fin("final_reward"))),
                                  "scopa_trained_mean": float(np.mean(fin("final_scopa_trained"))),
                                  "scopa_trained_std": float(np.std(fin("final_scopa_trained"))),
                                  "scopa_random_mean": float(np.mean(fin("final_scopa_random"))),
                                  "scopa_random_std": float(np.std(fin("final_scopa_random")))}}
        return data

    def save_data_for_plotting(self, path=None):
        path = Path(path) if path else self.save_dir / f"{self.experiment_name}_data.json"
        path.parent.mkdir(parents=True, exist_ok=True)
        with open(path, "w") as f:
            json.dump(self.plot_data(), f, indent=2)
        return path
