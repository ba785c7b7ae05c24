"""bench.py's launcher and multi-rank control flow (`--gpus N` must really run N ranks
or fail loudly).  CPU part: the refusal, the torchrun-environment check and a 2-rank gloo run of the
whole control flow (spawn -> rendezvous -> barriers -> max-over-ranks timing -> gather -> one JSON
line) with a sleep in place of the GPU step.  GPU part: the real bench on tiny shapes, one rank and
two gloo ranks sharing cuda:0."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=600):
    import tempfile

    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    fd, full = tempfile.mkstemp(prefix="bjx_bench_full_", suffix=".json")
    os.close(fd)
    os.unlink(full)
    env["BJX_BENCH_FULL"] = full  # the FULL record (the stdout line is the compact one, <= 6 KB)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env,
                       timeout=timeout, cwd=ROOT)
    r.full_path = full
    return r


def _json_line(stdout):
    lines = [ln for ln in stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    assert len(lines[0]) <= 6000, len(lines[0])  # the driver keeps an 8 KB tail of stdout
    return json.loads(lines[0])


def _full(r):
    """The full record bench.py writes next to the compact stdout line (also on stderr)."""
    try:
        with open(r.full_path) as fh:
            j = json.loads(fh.read())
    finally:
        if os.path.exists(r.full_path):
            os.unlink(r.full_path)
    on_stderr = [ln for ln in r.stderr.splitlines() if ln.startswith("bench.py FULL RECORD: ")]
    assert len(on_stderr) == 1 and json.loads(on_stderr[0][len("bench.py FULL RECORD: "):]) == j
    return j


def test_gpus_flag_refuses_instead_of_running_fewer_ranks():
    n = (torch.cuda.device_count() if torch.cuda.is_available() else 0) + 1
    r = _run(["--gpus", str(max(n, 2))])
    assert r.returncode != 0
    assert "refusing to run" in r.stderr and "{" not in r.stdout


def test_gpus_flag_must_match_the_torchrun_world():
    r = _run(["--gpus", "2", "--selftest-control-flow"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_two_rank_control_flow_over_gloo():
    r = _run(["--gpus", "2", "--selftest-control-flow", "--steps", "4", "--warmup", "1"])
    assert r.returncode == 0, r.stderr
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["value"] is None and "NOT a measurement" in j["metric"]
    assert [x["rank"] for x in j["ranks_seen"]] == [0, 1]
    assert len({x["pid"] for x in j["ranks_seen"]}) == 2  # two processes
    assert len(j["per_rank_ms_per_step"]) == 2
    assert abs(j["ms_per_step"] - max(j["per_rank_ms_per_step"])) < 1e-9  # max over ranks
    assert j["ms_per_step"] >= 4.0  # rank 1 sleeps 4 ms per step: the slowest rank sets the time
    assert j["gathered_rows"] == [4, 3] and j["backend"] == "gloo"


TINY = ["--chains", "4096", "--dim", "256", "--leapfrogs", "6", "--steps", "4", "--warmup", "1"]


@pytest.mark.gpu
def test_bench_c2_tiny_single_rank_json_contract():
    r = _run(TINY)
    assert r.returncode == 0, r.stderr[-2000:]
    c = _json_line(r.stdout)  # the compact line of the contract
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity",
              "ess_nonresonant", "full_record"):
        assert k in c, k
    assert c["n_gpus"] == 1 and c["steps"] == 4 and c["dtype"] == "f32" and c["vs_baseline"] is None
    assert c["roofline"]["bound"] == "hbm" and abs(c["roofline"]["frac"] - c["roofline"]["achieved"] / c["roofline"]["peak"]) < 1e-5
    assert c["parity"]["accept_mismatches"] == 0 and c["parity"]["chains_checked"] > 0
    assert c["cpu_baseline"]["kind"] in ("port", "reference") and c["cpu_baseline"]["cores"] >= 1
    j = _full(r)
    assert abs(j["value"] - c["value"]) <= 1e-5 * j["value"] and j["config"] == c["config"]
    roof = j["roofline"]
    assert roof["bound"] == "hbm" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    assert roof["chains_per_launch"] == 4096 and "streaming" in roof["mode"]
    assert roof["traffic_source"] is None or "NOT measured in this run" in roof["traffic_source"]
    assert j["cpu_baseline"]["kind"] in ("port", "reference") and j["cpu_baseline"]["value"] > 0
    tc = j["torch_callable_mode"]  # a plain PyTorch log-density, as the user writes it: the DEFAULT path (traced)
    assert tc["value"] > 0 and "elementwise" in tc["path"]
    ta = j["torch_autograd_mode"]  # the same function kept on eager autograd (blackjax_amd.no_trace)
    assert ta["value"] > 0 and "autograd" in ta["logdensity"] and "autograd" in ta["path"]
    # the three labelled user-callable lines (autograd, autograd under a HIP graph, plain-torch pair)
    assert j["torch_pair_mode"]["value"] > 0 and "no autograd" in j["torch_pair_mode"]["logdensity"]
    tg = j["torch_callable_graph_mode"]
    assert tg["value"] is None or (tg["value"] > 0 and tg["hip_graph"] is True)
    assert j["ranks"] == 1 and j["devices_distinct"] == 1
    assert j["rng_pin"].split(":")[0].split(" ")[0] in ("verified", "mismatch", "unavailable")
    assert j["ess"] is not None and j["ess_nonresonant"]["eps"] == 0.21
    assert 0.3 < j["mean_acceptance"] <= 1.0


@pytest.mark.gpu
def test_bench_two_gloo_ranks_share_one_gpu():
    """The real bench under its own launcher with 2 ranks (gloo, both on cuda:0 of a 1-GPU box):
    n_gpus = 2, twice the chains, both ranks seen."""
    r = _run(["--gpus", "2", "--no-cpu-baseline"] + TINY, {"BJX_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    c = _json_line(r.stdout)
    assert c["n_gpus"] == 1 and c["ranks"] == 2 and "cpu_baseline" not in c
    j = _full(r)
    # two gloo ranks on ONE device: n_gpus counts distinct devices, `ranks` the processes
    assert j["n_gpus"] == 1 and j["ranks"] == 2 and j["devices_distinct"] == 1 and "n_gpus_note" in j
    assert j["config"]["global_chains"] == 8192
    assert [x["rank"] for x in j["ranks_seen"]] == [0, 1] and len(j["per_rank_ms_per_step"]) == 2
    assert j["final_draws_gathered"] == [512, 256]
    assert "cpu_baseline" not in j


@pytest.mark.gpu
def test_bench_c4_tiny():
    r = _run(["--config", "c4", "--chains", "1024", "--dim", "512", "--leapfrogs", "5", "--steps", "24",
              "--warmup", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["roofline"]["bound"] == "hbm"
    j = _full(r)
    assert "window_adaptation" in j["metric"] and j["config"]["workload"].startswith("C4")
    assert j["value"] > 0 and j["roofline"]["algorithmic_bytes_per_launch"] == 24.0 * 512 * 1024


@pytest.mark.gpu
def test_bench_c3_tiny():
    r = _run(["--config", "c3", "--chains", "512", "--dim", "64", "--steps", "6", "--warmup", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["config"]["workload"].startswith("C3: NUTS")
    j = _full(r)
    assert j["config"]["workload"].startswith("C3: NUTS") and j["value"] > 0 and j["steps"] == 6
    assert j["roofline"]["bound"] == "hbm" and j["roofline"]["algorithmic_bytes_per_chain_leapfrog"] == 52.0 * 64
    assert abs(j["roofline"]["frac"] - j["roofline"]["achieved"] / j["roofline"]["peak"]) < 1e-12
    assert j["lockstep_step"]["value"] > 0 and 1.0 <= j["mean_leapfrogs_per_chain_transition"] <= 1023.0


@pytest.mark.gpu
def test_bench_c5_tiny():
    r = _run(["--config", "c5", "--chains", "1024", "--dim", "128", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["roofline"]["bound"] == "mfma"
    j = _full(r)
    assert j["config"]["workload"].startswith("C5: dense") and j["value"] > 0
    roof = j["roofline"]
    assert roof["bound"] == "mfma" and roof["algorithmic_flops_per_launch"] == 2.0 * 1024 * 128 * 128
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12


@pytest.mark.gpu
def test_bench_default_line_carries_c3_c5_c4_sub_objects():
    """The default run (no --chains / --dim): C2 headline + BASELINE.json configs[2..4] as sub-objects.
    Shortened with --steps; the sub-objects keep their own step counts."""
    r = _run(["--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-torch-callable", "--no-ess-nonresonant"],
             timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    c = _json_line(r.stdout)  # the compact line carries every config's value, roofline fraction and parity counts
    for k, start in (("c3_nuts", "C3: NUTS"), ("c5_dense", "C5: dense"), ("c4_shard", "C4: window_adaptation")):
        assert c[k]["value"] > 0 and c[k]["ms_per_step"] > 0 and c[k]["workload"].startswith(start), c[k]
        assert c[k]["roofline"]["frac"] > 0 and c[k]["roofline"]["bound"] in ("hbm", "mfma")
    assert c["c5_dense"]["parity"]["accept_mismatches"] == 0 and c["c5_dense"]["parity"]["chains_checked"] > 0
    assert c["c3_nuts"]["parity"]["tree_size_mismatches"] == 0 and c["c3_nuts"]["parity"]["chains_checked"] > 0
    assert c["parity"]["accept_mismatches"] == 0
    j = _full(r)
    for k, start in (("c3_nuts", "C3: NUTS"), ("c5_dense", "C5: dense"), ("c4_shard", "C4: window_adaptation")):
        assert k in j, k
        assert j[k].get("value"), j[k]
        assert j[k]["config"]["workload"].startswith(start)
        assert j[k]["roofline"]["frac"] > 0
    assert j["c3_nuts"]["steps"] == 100 and j["c4_shard"]["steps"] == 200


def test_compact_line_of_a_committed_full_record_fits_the_drivers_tail():
    """the driver keeps an 8 KB tail of stdout.  `bench.compact_line` on the committed full record of
    round 6 (profiles/r06/bench_final_r06.json, ~26 KB): at most 6 000 bytes, one JSON object, and it carries the
    contract keys, the roofline / cpu_baseline objects, every config's value, roofline fraction and parity counts."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("_bjx_bench", BENCH)
    bench = importlib.util.module_from_spec(spec)
    argv = sys.argv
    sys.argv = [BENCH]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    full = json.load(open(os.path.join(ROOT, "profiles", "r06", "bench_final_r06.json")))
    assert len(json.dumps(full)) > 15000
    line = bench.compact_line(full, "gpurun_out/bench_full_latest.json")
    assert len(line) <= 6000 and "\n" not in line
    c = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "ess_nonresonant"):
        assert k in c, k
    assert c["value"] == pytest.approx(full["value"], rel=1e-5) and c["vs_baseline"] is None
    assert c["roofline"]["frac"] == pytest.approx(c["roofline"]["achieved"] / c["roofline"]["peak"], rel=1e-5)
    assert c["roofline"]["algorithmic_bytes_per_launch"] == 20 * 1024 * c["roofline"]["chains_per_launch"]  # integers stay exact
    for k in ("c3_nuts", "c5_dense", "c4_shard"):
        assert c[k]["value"] > 0 and c[k]["ms_per_step"] > 0 and 0 < c[k]["roofline"]["frac"] < 1
    assert c["c5_dense"]["parity"]["accept_mismatches"] == 0 and c["c3_nuts"]["parity"]["tree_size_mismatches"] == 0
    assert c["c3_nuts"]["lockstep_step"]["value"] > 0 and c["c3_nuts"]["free_running_T400"]["value"] > 0
