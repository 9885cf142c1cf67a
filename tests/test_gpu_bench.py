"""GPU: the bench's replayed training step end to end (the path `bench.py --workload train_real --graph` takes): buckets captured
ahead of the timed region, no overflow, a finite loss next to the eager run's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags):
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "train_real", "--steps", "6", "--warmup", "2",
                          "--no-kernel-timers", *flags], cwd=ROOT, capture_output=True, text=True, timeout=600)
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert run.returncode == 0 and lines, (run.stderr or run.stdout)[-1500:]
    return json.loads(lines[-1])


def test_replayed_training_step_of_the_bench():
    g = _bench("--graph")
    e = _bench()
    hg = g["config"]["hip_graph"]
    assert hg["captures_inside_the_timed_region"] == 0 and hg["overflowed_batches"] == 0 and hg["graphs_captured"] >= 3
    assert hg["samples_of_last_step"] <= hg["capacity_of_last_step"] and hg["nodes_per_replayed_step"] > 100
    assert g["unit"] == "rays/s" and g["value"] > 0 and g["vs_baseline"] is None and g["n_gpus"] == 1
    lg, le = g["config"]["loss_mean_of_timed_steps"], e["config"]["loss_mean_of_timed_steps"]
    assert lg == lg and le == le and abs(lg - le) <= 0.1 * abs(le), (lg, le)      # same model, same occupancy, other random batches
