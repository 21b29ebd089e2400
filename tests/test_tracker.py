"""scopa_amd.experiment_tracker against the reference's own committed experiment output.

tests/golden/MiniScopa_MCCFR_data.reference.json is the `_data.json` the reference's ExperimentTracker wrote for its published
10-run MCCFR experiment (experiment_tracker.py:82-158).  Feeding that document's `runs` through the build's tracker must give the
whole document back -- key structure, key ORDER (plot_mccfr.py and spreadsheet users see it), and every statistic bit for bit."""
import json

from scopa_amd.experiment_tracker import ExperimentMetrics, ExperimentTracker


def _structure(x):
    if isinstance(x, dict):
        return {k: _structure(v) for k, v in x.items()}
    if isinstance(x, list):
        return [type(x[0]).__name__ if x else None, len(x)] if not x or not isinstance(x[0], (dict, list)) else [_structure(x[0]), len(x)]
    return type(x).__name__


def _tracker_from(ref):
    tr = ExperimentTracker(ref["experiment_name"], save_dir="unused")
    for run in ref["runs"]:
        m = ExperimentMetrics(algorithm=ref["algorithm"])
        for k, v in run.items():
            if k != "run_id":
                setattr(m, k, v)
        tr.add_run(m)
    return tr


def test_tracker_reproduces_the_references_document(golden, tmp_path):
    ref = golden.json("MiniScopa_MCCFR_data.reference.json")
    tr = _tracker_from(ref)
    mine = tr.plot_data()
    assert _structure(mine) == _structure(ref)
    assert list(mine) == list(ref) and list(mine["runs"][0]) == list(ref["runs"][0])
    assert list(mine["statistics"]) == list(ref["statistics"]) and list(mine["statistics"]["final_metrics"]) == list(ref["statistics"]["final_metrics"])
    assert mine == ref                                               # every mean / std / min / max, exactly
    path = tr.save_data_for_plotting(tmp_path / "out" / "MiniScopa_MCCFR_data.json")
    assert json.load(open(path)) == ref                              # and through the file the plotting script reads


def test_tracker_single_run_and_exploitability_fields(golden):
    ref = golden.json("MiniScopa_MCCFR_data.reference.json")
    tr = ExperimentTracker("one", save_dir="unused")
    m = ExperimentMetrics(algorithm="CFR", eval_iterations=[5, 10], eval_rewards=[0.1, 0.2], eval_scopas_trained=[0.3, 0.4],
                          eval_scopas_random=[0.1, 0.1], eval_scopa_diff=[0.2, 0.3], final_reward=0.5, num_info_sets=738,
                          exploitability_iterations=[10], exploitability_values=[0.25])
    tr.add_run(m)
    d = tr.plot_data()
    assert "statistics" not in d and d["num_runs"] == 1 and d["algorithm"] == "CFR"          # statistics need > 1 run (:118)
    assert list(d["runs"][0])[:len(ref["runs"][0])] == list(ref["runs"][0])
    assert d["runs"][0]["exploitability_iterations"] == [10] and d["runs"][0]["exploitability_values"] == [0.25]
    assert ExperimentTracker("none", save_dir="unused").plot_data() == {"experiment_name": "none", "algorithm": "Unknown", "num_runs": 0, "runs": []}
