"""bench.py's own launcher and N>1 plumbing on the CPU: `python bench.py --gpus 2` (no torchrun environment) must spawn
two ranks through torch.distributed.run, rendezvous on 127.0.0.1, run warm-up + timed steps bracketed by barriers,
take the MAX over ranks and print ONE JSON line from rank 0 that says what world it saw.  The step itself is the
MORPHEUS_BENCH_STUB toy (the real step needs an MI355X; tests/test_gpu_dist.py covers it on the GPU box)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MORPHEUS_BENCH_STUB"] = "1"
    env.update(extra_env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and out.stdout.rstrip().splitlines()[-1] == lines[0], out.stdout
    assert len(lines[0]) < 8192, len(lines[0])            # the driver keeps an 8 KB tail of stdout: the line must fit in it
    return json.loads(lines[0])


def test_self_launch_two_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--rays", "64"])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1
    assert r["config"]["world_size"] == 2 and r["config"]["backend"] == "gloo"
    assert r["value"] > 0 and abs(r["value"] - 2 * 64 * 3 / (r["ms_per_step"] * 3e-3)) / r["value"] < 1e-3   # whole-job rays/s
    assert r["scaling"] == "weak" and r["higher_is_better"] is True and r["cpu_baseline"] is None


def test_self_launch_eight_ranks():
    r = _run(["--gpus", "8", "--steps", "2", "--warmup", "1", "--rays", "32"])
    assert r["n_gpus"] == 8 and r["config"]["world_size"] == 8 and r["value"] > 0


def test_single_rank_needs_no_launcher_and_overlap_switch():
    r = _run(["--steps", "2", "--warmup", "1", "--rays", "32", "--no-overlap"])
    assert r["n_gpus"] == 1 and r["config"]["world_size"] == 1 and r["config"]["backend"] is None


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, MORPHEUS_BENCH_STUB="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)


def test_driver_line_is_compact_and_complete(tmp_path):
    """Round 4's line was 34 KB and came back from the driver unparsed.  The LAST stdout line is a compact object (< 6 KB) carrying
    the contract's keys plus `roofline` and `cpu_baseline`; the full object goes to --detail-out.  Checked (a) end to end on the stub
    and (b) by compacting the largest full object ever produced (round 4's committed one)."""
    import bench
    detail = tmp_path / "d.json"
    r = _run(["--steps", "2", "--warmup", "1", "--rays", "32", "--detail-out", str(detail)])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    full = json.load(open(detail))
    assert full["value"] == r["value"] and "kernels" in full and "kernels" not in r
    big = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_cfg3.json")))
    line = bench.compact_line(big, os.path.join(ROOT, "bench_detail.json"))
    assert len(line) < bench.COMPACT_LIMIT < 8192 and "\n" not in line
    c = json.loads(line)
    assert c["value"] == big["value"] and c["ms_per_step"] == big["ms_per_step"] and c["config"]["workload"].endswith("(cfg3)")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert c["roofline"][k] == big["roofline"][k], k
    assert set(c["roofline"]["per_kernel"]) == set(big["roofline"]["per_kernel"])
    assert c["cpu_baseline"]["value"] == big["cpu_baseline"]["value"] and c["cpu_baseline"]["cores"] == 32 and c["cpu_baseline"]["kind"] == "port"
    assert c["train_real_ms"]["reference_glue"] > c["train_real_ms"]["graph"] > 0 and c["train_loop_iters_per_s"] > 0
    assert set(c["modes_ms_per_step"]) >= {"b3", "f32"}
    # nothing named "frac" in the driver's line may read > 1 (round 5's hash-grid entry did: algorithmic bytes over the HBM peak):
    # re-price the newest committed kernel table with today's bench.py and walk the compact line
    newest = json.load(open(next(p for p in (os.path.join(ROOT, "profiles", n) for n in ("r06_bench_cfg3_detail.json", "r05_bench_cfg3_detail.json"))
                                 if os.path.exists(p))))
    M = float(newest["config"]["sample_points_per_step_per_gpu"])
    newest["roofline_hashgrid"] = bench.build_hash_roofline(newest["kernels"], M, "cfg3", True, "b3")
    newest["roofline"] = bench.build_roofline(newest["kernels"], "b3", M, "cfg3", True)
    cl = json.loads(bench.compact_line(newest))
    rh = cl["roofline_hashgrid"]
    assert 0 < rh["frac"] < 1 and 0 < rh["bwd_frac"] < 1 and rh["algorithmic_gbs"] > rh["achieved"] and rh["compulsory_bytes"] == 140 * int(M)
    assert rh["traffic_source"].startswith("profiles/") and cl["roofline"]["traffic_source"].startswith("profiles/")

    def fracs(o, path=""):
        if isinstance(o, dict):
            for k, v in o.items():
                if isinstance(v, (int, float)) and (k == "frac" or k.endswith("_frac") or k.startswith("frac_")):
                    yield path + k, v
                else:
                    yield from fracs(v, path + k + ".")
    found = dict(fracs(cl))
    assert len(found) >= 8 and all(0 <= v <= 1 for v in found.values()), found
    # a pathological object (huge strings everywhere) still yields a parseable line under the tail size
    big["config"]["workload"] = "x" * 5000
    big["cpu_baseline"]["sample"] = "y" * 5000
    assert len(bench.compact_line(big)) < bench.COMPACT_LIMIT


def test_committed_bench_line_recomputes_from_committed_profiles():
    """The judged bench line (profiles/r0N_bench_cfg3.json, newest round) must be recomputable from what is committed next to it:
    per mode, `roofline` = the step's largest time item of that mode's kernel table, priced by bench.build_roofline per SURVEY
    8(d) -- top-level frac = algorithmic FLOPs / time over (the pipe's dense peak / slice products per MAC); the parked-bytes view
    rides beside it under `hbm` -- with `traffic` from the PMC summary of the same mode (profiles/r0N_pmc_summary[_mode].csv);
    every per-kernel frac likewise; the headline is the faster fp32-faithful mode."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    path = next(p for p in (os.path.join(root, "profiles", n) for n in ("r06_bench_cfg3_detail.json", "r05_bench_cfg3_detail.json", "r04_bench_cfg3.json",
                                                                      "r03_bench_cfg3.json")) if os.path.exists(p))
    d = json.load(open(path))
    assert set(d["modes"]) >= {"b3", "f32"} and d["headline_mode"] in ("b3", "f32")      # (rounds 3-5 also carried the deleted h2 mode)
    faithful = {m: d["modes"][m]["ms_per_step"] for m in ("b3", "f32")}
    assert d["headline_mode"] == min(faithful, key=faithful.get) and d["ms_per_step"] == faithful[d["headline_mode"]]
    assert d["dtype"].startswith("f32")
    M = float(d["config"]["sample_points_per_step_per_gpu"])
    for m, r in d["modes"].items():
        if m not in bench.PRODUCTS:
            continue
        ro = bench.build_roofline(r["kernels"], m, M, "cfg3", True)
        stored = r["roofline"]
        largest = max((v["ms_per_step"], k) for k, v in r["kernels"].items() if k in bench.IO_BYTES)[1]
        assert ro["kernel"] == stored["kernel"] == largest, (m, ro["kernel"], stored["kernel"], largest)
        assert ro["bound"] == "mfma" and ro["unit"] == "TFLOP/s"
        # round 3's line carried the algorithmic-FLOP view under "mfma" and put the nearer of two roofs on top; from round 4
        # on the top-level numbers ARE the algorithmic-FLOP view
        smf = stored.get("mfma", stored)
        for k in ("achieved", "peak", "frac"):
            assert ro[k] == smf[k], (m, k, ro[k], smf[k])
        assert ro["hbm"]["frac"] == stored["hbm"]["frac"]
        assert ro["algorithmic_bytes"] == stored["algorithmic_bytes"] == 32 * int(M)
        if stored["traffic"] and "per_kernel" in stored:
            assert abs(ro["traffic"] - stored["traffic"]) <= 0.02 * stored["traffic"], (m, ro["traffic"], stored["traffic"])
            assert stored["traffic"] > 100 * stored["algorithmic_bytes"]        # the parking waste is visible in the line
        for k, v in stored.get("per_kernel", {}).items():
            assert ro["per_kernel"][k]["frac"] == v["frac"] and 0 < v["frac"] < 1, (m, k)
    if "per_kernel" in d["roofline"]:
        assert d["roofline"]["frac"] == d["modes"][d["headline_mode"]]["roofline"]["frac"]
        for k in ("train_real", "train_virtual"):                  # the training-step workloads ride in the driver's line
            assert k in d and all(v.get("ms_per_step", 0) > 0 for v in d[k].values() if isinstance(v, dict) and "error" not in v), k
