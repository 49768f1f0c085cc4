"""bench.py as a command: the single-GPU line, the multi-rank line started WITHOUT torch.distributed.run (bench.py
launches its own ranks as child processes), and BASELINE config 5 as a command (--scene leapfrog --grid .. --dump):
two z-slab ranks write one file per slab and frame, which stitched together equal the single-GPU dump byte for byte.
The ranks share GPU 0 (host-staged transport), so the figures mean nothing here -- the plumbing is what is tested."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run_bench(*args, timeout=900):
    env = dict(os.environ, OMP_NUM_THREADS="4")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):                      # a clean single-process environment
        env.pop(k, None)
    r = subprocess.run([sys.executable, BENCH, *map(str, args)], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    line = run_bench("--size", 64, "--steps", 3, "--warmup", 1, "--jacobi-iters", 40, "--cpu-n", 16, "--cpu-steps", 1)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["dtype"] == "f32" and line["vs_baseline"] is None
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["frac"] < 1.0 and rf["compulsory_bytes_per_launch"] == 12 * 64 ** 3 and "algorithmic_equiv" in rf
    # the launch's HBM traffic is measured IN THIS RUN (two rocprofv3 --pmc child passes), not looked up: at least the bytes a
    # launch has to move, at most a few times that on this tiny grid
    assert rf["traffic_source"].startswith("MEASURED IN THIS RUN"), rf["traffic_source"]
    assert 0.9 * rf["compulsory_bytes_per_launch"] < rf["traffic"] < 6 * rf["compulsory_bytes_per_launch"], rf["traffic"]
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0
    # the second kernel family against ITS bounds: phase times of the step and the gather family's object
    ph, rg = line["phase_ms_per_step"], line["roofline_gather"]
    assert set(ph) == {"maps", "advect_compensate", "forces", "projection", "accumulate_reinit"} and ph["advect_compensate"] > 0
    assert rg["frac"] < 0.5 and "NOT this run" in rg["counters"]["source"] and rg["ms_per_step"] == ph["advect_compensate"]
    # ... and its counters measured in THIS run (three rocprofv3 --pmc child passes): VALU issue and texture-addresser busy fractions
    cl = rg["counters_live"]
    assert cl["source"].startswith("MEASURED IN THIS RUN"), cl["source"]
    assert 0.05 < cl["valu_issue_busy"]["single_field"][0] <= cl["valu_issue_busy"]["single_field"][1] < 1.3
    assert 0.05 < cl["ta_busy"]["single_field"][1] < 1.3 and cl["l1_to_l2_bytes_over_algorithmic"]["single_field"][1] > 0.5


def test_multi_rank_bench_launches_itself():
    """`python bench.py --gpus 2` with no launcher around it: rc 0, ONE JSON line, strong scaling of ONE grid (the
    shape of BASELINE config 4), two ranks in the communicator, the reference-faithful map border"""
    line = run_bench("--gpus", 2, "--transport", "host", "--size", 64, "--steps", 2, "--warmup", 1, "--jacobi-iters", 30, "--no-extra")
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["global_grid"] == [64, 64, 64] and line["config"]["grid_per_gpu"] == [64, 64, 32]
    assert line["config"]["comm_size"] == 2
    assert "BQ_OPT_KEEP_DMC_BORDER = 0" in line["config"]["parallelism"]
    assert line["value"] > 0
    # N > 1 lines diagnose themselves (here: host-staged transport, whose blocking calls count as exposed communication)
    d = line["diagnostics"]
    assert len(d["per_rank"]) == 2 and d["comm_exposed_ms_per_step"] > 0
    assert d["phase_ms_per_step_slowest_rank"]["projection"] > 0
    assert {"shallow_exchange_2", "ends_first_off", "jacobi_triples_off", "reserve_cus_8", "reserve_cus_16"} <= set(line["extra"])


def test_emulated_rank_line_carries_phases_and_knob_legs():
    """--emulate-slab: one process plays a middle rank with a transport that moves nothing; the diagnostic leg gives the
    per-phase times of the slab code path and the compute-side price of each knob (the CU-masked stream included)"""
    line = run_bench("--size", 64, "--emulate-slab", 2, "--steps", 3, "--warmup", 1, "--jacobi-iters", 40, "--no-cpu-baseline",
                     "--no-extra", "--diag-steps", 2)
    assert "EMULATED" in line["metric"]
    d = line["diagnostics"]
    assert d["steps"] == 2 and d["comm_exposed_ms_per_step"] == 0.0 and len(d["per_rank"]) == 1
    assert d["phase_ms_per_step_slowest_rank"]["advect_compensate"] > 0
    assert line["extra"]["reserve_cus_8"]["ms_per_step"] > 0 and line["config"]["reserved_cus"] == 0


def test_roofline_names_the_source_of_its_traffic_figure():
    line = run_bench("--steps", 3, "--warmup", 2, "--no-cpu-baseline", "--no-extra")      # 256^3: a committed PMC pass exists
    rf = line["roofline"]
    assert rf["traffic"] and "profiles/jacobi_pmc_traffic.json" in rf["traffic_source"] and "NOT measured in this run" in rf["traffic_source"]


def test_config5_command_slab_dumps_stitch_to_the_single_gpu_dump(tmp_path):
    from gpufluidsimulation_amd.solver import read_density_dump
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    common = ["--scene", "leapfrog", "--grid", 64, 64, 32, "--steps", 3, "--warmup", 0, "--jacobi-iters", 30,
              "--no-cpu-baseline", "--no-extra"]
    a = run_bench(*common, "--dump", one)
    b = run_bench("--gpus", 2, "--transport", "host", *common, "--dump", two)
    assert "leapfrogging" in a["metric"] and a["config"]["global_grid"] == [64, 64, 32] and b["n_gpus"] == 2
    # the ring axis passes between nodes (scenes.py): no 0/0 in the emitter, the run stays finite (rounds 1-2: NaN from frame 0)
    assert a["config"]["nonfinite_velocity_seen"] is False and b["config"]["nonfinite_velocity_seen"] is False
    assert "dumped every frame" in a["config"]["workload"]
    frames = sorted(os.listdir(one))
    assert frames == [f"density_render_{i:04d}.bqd" for i in (1, 2, 3)], frames
    for f in frames:
        hd, rec = read_density_dump(os.path.join(one, f))
        parts = sorted(p for p in os.listdir(two) if p.startswith(f[:-4] + ".k"))
        assert len(parts) == 2, (f, os.listdir(two))
        stitched = []
        covered = 0
        for p in parts:
            h2, r2 = read_density_dump(os.path.join(two, p))
            assert (h2["nx"], h2["ny"], h2["nz"]) == (64, 64, 32) and h2["k_offset"] == covered
            covered += int(h2["nz_local"])
            stitched.append(r2)
        assert covered == 32
        stitched = np.concatenate(stitched)
        assert len(rec) > 100 and stitched.tobytes() == rec.tobytes(), f
