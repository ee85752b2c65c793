"""bench.py's N > 1 control flow without GPUs: `python bench.py --gpus 2` must launch its own ranks, exchange the
gradient of every step with the overlapped double-buffered all-reduce, and print ONE JSON line from rank 0.
FR_BENCH_STUB=1 swaps the rasterizer for a deterministic CPU gradient (rank- and step-dependent), FR_DIST_BACKEND=gloo
the collective backend; the launch, exchange, barrier and timing code is the code the GPU run executes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = dict(os.environ, FR_BENCH_STUB="1", FR_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True,
                       timeout=300)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    return json.loads(lines[0])


@pytest.mark.parametrize("overlap,K,R", [(True, 3, 1), (False, 3, 2), (True, 1, 1), (True, 2, 3), (True, 1, 4)])
def test_bench_spawns_two_ranks_and_reports_both_modes(overlap, K, R):
    """N > 1: the line's `value` is the LITERAL BASELINE configs[3] shape (one view per rank per exchange, the exchange
    finished before the next step), and the amortised mode (R rounds of K views per exchange, overlapped) sits beside it in
    dp.modes; every step of both modes averages its gradient over the ranks."""
    steps, warm = 7, 3
    j = _run(["--gpus", "2", "--steps", str(steps), "--warmup", str(warm), "--in-flight", str(K), "--rounds", str(R)]
             + ([] if overlap else ["--no-overlap"]))
    assert j["n_gpus"] == 2 and j["data"] == "stub" and j["steps"] == steps and j["scaling"] == "weak"
    d = j["dp"]
    assert d["ranks_seen"] == 2 and d["backend"] == "gloo" and d["overlap"] is overlap
    assert d["num_rendered_per_rank"] == [1000, 1001]
    assert d["allreduce_payload_bytes"] == 4 * 4096 and d["allreduce_us"] > 0
    assert d["allreduce_table"][0]["payload_bytes"] == 4 * 4096 and d["allreduce_table"][0]["us"] > 0
    lit, amo = d["modes"]["literal"], d["modes"]["amortised"]
    # ---- the literal mode is the headline
    assert (lit["frames_per_step_per_gpu"], lit["rounds_per_step"], lit["frames_in_flight_per_gpu"], lit["overlap"]) == (1, 1, 1, False)
    assert j["value"] == lit["value"] and j["ms_per_step"] == lit["ms_per_step"]
    assert j["config"]["frames_per_step_per_gpu"] == 1 and j["config"]["rounds_per_step"] == 1
    assert abs(lit["value"] - 2 * steps / (lit["ms_per_step"] * steps * 1e-3)) <= 0.02 * lit["value"]
    # the stub gradient of (rank r, view v of its K, round k counted from the mode's start) is
    # i * 1e-3 + (r * K + v + 1) * (k + 1); literal mode: K = 1, one round per step -> mean over the two ranks 1.5 * (k + 1)
    want = sum(i * 1e-3 + 1.5 * (warm + steps) for i in range(4096))
    assert abs(lit["grad_checksum"] - want) <= 1e-4 * want, (lit["grad_checksum"], want)
    # ---- the amortised mode beside it
    assert (amo["frames_per_step_per_gpu"], amo["rounds_per_step"], amo["frames_in_flight_per_gpu"], amo["overlap"]) == (K * R, R, K, overlap)
    assert abs(amo["value"] - 2 * K * R * steps / (amo["ms_per_step"] * steps * 1e-3)) <= 0.02 * amo["value"]
    # A step is R rounds; the exchange buffer must hold the mean over the R rounds of the LAST step, the K views of each
    # rank and the two ranks: the mean of 1 .. 2K is (2K + 1) / 2, the mean of (k + 1) over the last step's rounds is
    # (warm + steps - 1) * R + (R + 1) / 2
    kmean = (warm + steps - 1) * R + (R + 1) / 2
    want = sum(i * 1e-3 + (2 * K + 1) / 2 * kmean for i in range(4096))
    assert abs(d["grad_checksum"] - want) <= 1e-4 * want, (d["grad_checksum"], want)
    # ---- the third mode: the reference's own step (FateAvatar's 'gs' group, 12 floats per Gaussian, one frame per rank)
    av = d["modes"]["avatar"]
    assert (av["frames_per_step_per_gpu"], av["rounds_per_step"], av["frames_in_flight_per_gpu"]) == (1, 1, 1)
    assert av["allreduce_payload_bytes"] % 48 == 0 and av["value"] > 0 and av["sh_degree"] == 0
    assert abs(av["mean_check"] - ((steps - 1) + 0.5)) < 1e-6      # (stub: the last step's buffer is the mean over the two ranks)
    # ---- a scaling curve over `value` has a like-for-like first point and this run's efficiency against it
    assert j["scaling_reference_status"] == "ok" and j["scaling_reference"] == j["dp_reference_at_1"]["literal"]["value"] > 0
    assert abs(j["efficiency"] - j["value"] / (2 * j["scaling_reference"])) <= 1e-3
    assert "efficiency" in amo and "efficiency" in av
    # ... and the generic literal step's ceiling from the run's own all-reduce table
    c = d["literal_ceiling"]
    assert c["allreduce_us"] == d["allreduce_table"][0]["us"] and 0.0 <= c["efficiency_ceiling"] <= 1.0
    assert abs(c["efficiency_ceiling"] - c["frame_us"] / (lit["ms_per_step"] * 1e3)) <= 2e-3


def test_bench_single_rank_stub_line_is_well_formed():
    j = _run(["--steps", "3", "--warmup", "1", "--cpu-seconds", "0"])
    assert j["n_gpus"] == 1 and j["dp"] is None and j["data"] == "stub"
    for k in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "config",
              "roofline", "cpu_baseline", "stage_us", "stage_frac", "stage_frac_required", "scaling_reference", "efficiency",
              "opaque", "roofline_config5", "coherent_layout", "one_frame_at_a_time"):
        assert k in j


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, FR_BENCH_STUB="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
