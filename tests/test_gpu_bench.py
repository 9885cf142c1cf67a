"""GPU: the bench's replayed training step end to end (the path `bench.py --workload train_real --graph` takes): buckets captured
ahead of the timed region, no overflow, a finite loss next to the eager run's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, workload="train_real", steps=6):
    """(full object from --detail-out, the compact last stdout line)"""
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        detail = os.path.join(td, "detail.json")
        run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", str(steps), "--warmup", "2",
                              "--no-kernel-timers", "--detail-out", detail, *flags], cwd=ROOT, capture_output=True, text=True, timeout=600)
        lines = run.stdout.rstrip().splitlines()
        assert run.returncode == 0 and lines and lines[-1].startswith("{"), (run.stderr or run.stdout)[-1500:]
        assert len(lines[-1]) < 8192
        line = json.loads(lines[-1])
        full = json.load(open(detail))
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]
    return full


def test_replayed_training_step_of_the_bench():
    g = _bench("--graph")
    e = _bench()
    hg = g["config"]["hip_graph"]
    assert hg["captures_inside_the_timed_region"] == 0 and hg["overflowed_batches"] == 0 and hg["graphs_captured"] >= 3
    assert hg["samples_of_last_step"] <= hg["capacity_of_last_step"] and hg["nodes_per_replayed_step"] > 100
    assert g["unit"] == "rays/s" and g["value"] > 0 and g["vs_baseline"] is None and g["n_gpus"] == 1
    lg, le = g["config"]["loss_mean_of_timed_steps"], e["config"]["loss_mean_of_timed_steps"]
    assert lg == lg and le == le and abs(lg - le) <= 0.1 * abs(le), (lg, le)      # same model, same occupancy, other random batches


def test_optimisation_loop_with_replayed_real_view_steps():
    """`bench.py --workload train_loop --graph`: per iteration one eager virtual-view step, one eager real-view step that adds its
    gradient to the virtual view's, nine real-view steps replayed from HIP graphs -- nothing captured inside the timed region, a
    finite loss next to the all-eager loop's.

    The two runs draw their own random batches and the table gradients are atomics, so two of the conditions are statistical: a batch
    whose sample count leaves the buckets captured ahead of the timed region is legal (it is captured on the spot and reported), and the
    mean of three single-step losses scatters.  One of ~12 runs of the suite on fresh boxes tripped here in round 6 and six isolated
    repetitions did not: those two conditions get ONE retry, with the figures in the assertion message; a bench process that dies
    (`_bench`) or overflowed batches fail at once."""
    tried = []
    for attempt in range(2):
        g = _bench("--graph", workload="train_loop", steps=3)
        e = _bench(workload="train_loop", steps=3)
        hg = g["config"]["hip_graph"]
        assert hg["overflowed_batches"] == 0, hg
        assert g["iters_per_s"] > 0 and e["iters_per_s"] > 0 and abs(g["train_steps_per_s"] / g["iters_per_s"] - 11.0) < 0.05
        lg, le = g["config"]["loss_mean_of_timed_steps"], e["config"]["loss_mean_of_timed_steps"]
        assert lg == lg and le == le, (lg, le)
        tried.append(dict(captures_inside_the_timed_region=hg["captures_inside_the_timed_region"], loss_graph=lg, loss_eager=le))
        # same model and occupancy, other random batches (3 single-step losses each)
        if hg["captures_inside_the_timed_region"] == 0 and abs(lg - le) <= 0.3 * abs(le):
            return
    assert False, tried
