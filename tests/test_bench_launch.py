"""bench.py --gpus N without an external launcher: the script spawns N ranks itself and reports as `n_gpus` the value of an
all_reduce of ones over the process group (`--gpus` used to be parsed and ignored)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=600):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_rank_environments():
    sys.path.insert(0, ROOT)
    import bench
    envs = bench.rank_environments(3, base_env={"FOO": "1"}, port=12345)
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert all(e["WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "12345" and e["FOO"] == "1" for e in envs)
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)


@pytest.mark.parametrize("n", [1, 2, 3])
def test_gpus_flag_spawns_that_many_ranks_cpu(n):
    """Dry run (gloo, no GPU work): `--gpus n` alone must produce n ranks that see each other."""
    line = _run(["--gpus", str(n)], {"ASDF_BENCH_DRYRUN": "1"})
    assert line["n_gpus"] == n and line["ranks_requested"] == n and line["world_size_env"] == n


@pytest.mark.gpu
def test_gpus_2_on_one_device():
    """The real bench with two ranks sharing the box's single GPU (gloo for the record gather): n_gpus == 2, twice the
    samples of one rank."""
    line = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--grid", "64", "--no-cpu-baseline"],
                {"ASDF_BENCH_BACKEND": "gloo", "ASDF_BENCH_SHARE_DEVICE": "1"})
    assert line["n_gpus"] == 2 and line["config"]["world_size_env"] == 2
    assert line["config"]["samples_per_gpu"] == 2 and line["scaling"] == "weak"
    assert abs(line["value"] - 2 * 2 * 2 / (line["ms_per_step"] * 2 * 1e-3)) < 1e-6 * line["value"]


@pytest.mark.gpu
def test_gpus_8_on_one_device():
    """the REAL bench as the driver would start it on an 8-GPU node - `--gpus 8`, eight ranks, the barrier +
    MAX-over-ranks timing, gather_records to rank 0 - with the eight ranks folded onto the box's single MI355X and the gather over
    gloo (the only difference to the node: RCCL and one device each).  n_gpus is the all-reduce of ones; every rank's records
    arrive; every rank was bound to its own block of host cores (dist_reconstruct.bind_host_cores)."""
    line = _run(["--gpus", "8", "--steps", "2", "--warmup", "1", "--grid", "64", "--no-cpu-baseline"],
                {"ASDF_BENCH_BACKEND": "gloo", "ASDF_BENCH_SHARE_DEVICE": "1"}, timeout=900)
    assert line["n_gpus"] == 8 and line["config"]["world_size_env"] == 8 and line["config"]["ranks_requested"] == 8
    assert line["config"]["samples_per_gpu"] == 2 and line["scaling"] == "weak"
    assert abs(line["value"] - 8 * 2 * 2 / (line["ms_per_step"] * 2 * 1e-3)) < 1e-6 * line["value"]
    assert line["config"]["records_gathered"] == 16 and line["config"]["ranks_in_records"] == list(range(8))
    assert line["config"]["host_threads_per_rank"] >= 1
    # round 5: the first multi-rank contact verifies itself - ranks seen over the collective, DISTINCT devices
    # behind them (here: eight ranks folded onto one device, and the line says so), every rank's own time and the spread
    c = line["config"]
    assert c["rccl_ranks_seen"] == 8 and c["backend"] == "gloo" and c["devices_distinct"] == 1 and c["hosts"] == 1
    t = c["per_rank_ms_per_step"]
    assert len(t["all"]) == 8 and 0 < t["min"] <= t["max"] and abs(t["max"] - line["ms_per_step"]) < 1e-6 * t["max"] + 1e-3
    assert t["spread"] >= 0.0 and [d["rank"] for d in c["rank_devices"]] == list(range(8))
    assert all(d["device_index"] == 0 and d["pci"] for d in c["rank_devices"])
    masks = [d["host_cpus"] for d in c["rank_devices"]]
    assert len(set(masks)) == 8                     # eight disjoint blocks of host cores


@pytest.mark.gpu
def test_bench_line_contract_hand_only_config0():
    """`--branches hand --grid 64` = BASELINE configs[0]: one mesh per sample, every field of the line the driver and the judge read.
    Round 6: `value` / `dtype` / the top-level `roofline` describe the product's DEFAULT - ordinary sweeps, every
    voxel of both lattices in the reference's arithmetic class - and the opt-in audited one-plane sweeps are a named scalar."""
    line = _run(["--branches", "hand", "--grid", "64", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs"], {})
    assert line["metric"] == "meshes_per_sec_hand_only_N64" and line["unit"] == "meshes/s" and line["n_gpus"] == 1
    cfg = line["config"]
    assert cfg["meshes_per_sample"] == 1 and cfg["coarse_pass"] == "exact" and cfg["fine_pass"] == "exact" and cfg["math"] == "f16x3"
    assert cfg["sweeps_are"] == "product default" and "every voxel" in cfg["coarse_pass_is"] and "every voxel" in cfg["fine_pass_is"]
    assert abs(line["value"] - 3 / (line["ms_per_step"] * 3e-3)) < 1e-6 * line["value"]
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None and line["data"] == "synthetic"
    assert "every voxel" in line["dtype"] and "2 x f16 planes" in line["dtype"] and "1 plane" not in line["dtype"]
    r = line["roofline"]
    # the kernel that produced `value`: the split-half kernel, 2 launches per sample
    assert r["kernel"] == "sdf_mlp_f16w_kernel" and r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["launches_timed"] == 6 and r["peak"] == 2516.6
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["achieved"] > 0
    assert abs(r["achieved"] - r["executed_flop_per_launch"] / (r["launch_ms"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
    assert "frac_algorithmic" not in r and r["reference_dense_fp32_tflops_equivalent"] > 0 and 0.5 < r["shader_clock_ghz"] < 2.6 and 0.0 < r["pipe_busy"] < 1.0
    assert 2 * r["launch_ms"] <= line["ms_per_step"]                  # the dominant kernel's two launches fit in the step
    # the line stays under the driver's 8 KB tail; what the parser keeps are SCALARS of `config` / `roofline` (it drops nested dicts)
    assert len(json.dumps(line)) < 6000 and set(line) <= {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                                          "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "details_file"}
    assert len(line["dtype"]) <= 130 and len(cfg["workload"]) <= 130 and cfg["baseline_config"] == "configs[0]"
    assert all(not isinstance(v, (dict, list)) for v in r.values()), [k for k, v in r.items() if isinstance(v, (dict, list))]
    # the fp32 MFMA chain (the strict reading) and the audited one-plane sweeps (narrower arithmetic for the signs: NOT the metric)
    assert cfg["meshes_per_s_fp32_mfma"] > 0 and cfg["meshes_per_s_fp32_mfma"] < line["value"] < cfg["meshes_per_s_audited_sign_sweeps"]
    assert r["fp32_kernel"] == "sdf_mlp_kernel" and r["fp32_peak"] == 157.3 and 0 < r["fp32_frac"] < 1 and r["fp32_launch_ms"] > r["launch_ms"]
    assert r["audited_kernel"] == "sdf_mlp_f16p1_kernel" and 0 < r["audited_frac"] < 1 and r["audited_launch_ms"] < r["launch_ms"]
    assert cfg["sweeps"]["of"].startswith("audited leg") and cfg["sweeps"]["refused"] == 0 and cfg["sweeps"]["calibrations"] >= 1
    assert cfg["sweeps"]["audited"] >= 4 and cfg["sweeps"]["min_tau_over_estimate"] >= 1.0 / 0.6
    q = cfg["parity_in_run"]
    assert q["samples"] == 3 and q["audited_sweeps_meshes_bit_identical_to_timed_run"] == 3 and q["faces_identical_to_fp32_chain"] == 3
    assert q["all_voxels_sign_differences"] == 0 and q["all_voxels_f16x3_vs_f32_max_abs"] < 4e-6
    with open(os.path.join(ROOT, line["details_file"])) as f:
        full = json.load(f)
    assert full["sweeps"]["coarse"] == "exact" and full["sweeps"]["refused_sweeps"] == 0 and full["sweeps"]["certificate"]["audited_sweeps"] == 0
    o = full["other_sweeps"]
    assert o["kind"] == "audited" and (o["coarse"], o["fine"]) == ("box", "band") and o["meshes_bit_identical"] == "3 / 3"
    c = o["sweeps"]["certificate"]
    assert c["calibrations"] >= 1 and c["refusals_for_error"] == 0 and c["min_margin_tau_over_estimate"] >= 1.0 / 0.6
    assert o["sweeps"]["refused_sweeps"] == 0 and o["sweeps"]["fine_sweeps"]["audit_evals"] > 0
    p = full["parity_in_run"]
    assert p["audited_sweeps_against_timed_run"]["vertices_identical"] == 3 and p["against_fp32_chain"]["faces_identical"] == 3
    assert p["volumes_f16x3_vs_f32"]["sign_differences"] == 0 and full["other_math"]["math"] == "f32"


@pytest.mark.gpu
def test_bench_line_under_fast_sweeps():
    """`--fast`: the timed region under the opt-in audited one-plane sweeps - the line says so in `dtype`, `config` and `roofline`, the
    every-voxel figure becomes the secondary one, and the meshes are the ordinary sweeps' bit for bit."""
    line = _run(["--branches", "hand", "--grid", "64", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-other-configs", "--no-other-math",
                 "--fast", "--sustained", "0"], {})
    cfg, r = line["config"], line["roofline"]
    assert cfg["coarse_pass"] == "box" and cfg["fine_pass"] == "band" and cfg["sweeps_are"] == "selected on the command line"
    assert "1 plane" in line["dtype"] and r["kernel"] == "sdf_mlp_f16p1_kernel" and r["launches_timed"] == 6
    assert 0.5 < r["shader_clock_ghz"] < 2.6 and 0.0 < r["pipe_busy"] < 1.0 and abs(r["frac_from_busy_and_clock"] - r["frac"]) < 0.02 * r["frac"]
    assert cfg["meshes_per_s_every_voxel_f16x3"] > 0 and cfg["meshes_per_s_every_voxel_f16x3"] < line["value"]
    assert r["every_voxel_kernel"] == "sdf_mlp_f16w_kernel" and 0 < r["every_voxel_frac"] < 1
    assert cfg["sweeps"]["of"] == "timed region" and cfg["sweeps"]["refused"] == 0
    assert cfg["parity_in_run"]["meshes_bit_identical_to_every_voxel_f16x3"] == 3


def test_short_line_keeps_the_reference_precision_figures_cpu():
    """short_line on a synthetic full record (no GPU): every figure the judge's ruling names ends up in `config` / `roofline`, the line
    is small, nothing else is at the top level."""
    sys.path.insert(0, ROOT)
    import bench
    big = {"blob": "x" * 20000}
    full = {"metric": "m", "value": 13.6, "unit": "meshes/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 147.0, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "d", "data": "synthetic", "config": {"workload": "w", "grid": 256},
            "roofline": {"bound": "mfma", "kernel": "k", "achieved": 1.0, "peak": 2.0, "unit": "TFLOP/s", "frac": 0.5, "traffic": None, "note": "n" * 900},
            "sweeps": {"refused_sweeps": 0, "certificate": dict(big, audited_sweeps=0, calibrations=0)},      # (the timed region: ordinary sweeps)
            "other_sweeps": dict(big, kind="audited", value=37.8, ms_per_step=52.9, kernel="sdf_mlp_f16p1_kernel", launch_ms=24.6, achieved=1450.0, frac=0.576,
                                 shader_clock_ghz=1.87, pipe_busy=0.8, meshes_bit_identical="20 / 20", sustained_ms_per_step=54.4, sustained_meshes_per_s=36.8,
                                 sustained_steps=130, sustained_recalibrations=2, sustained_refused_sweeps=0,
                                 sweeps={"refused_sweeps": 0, "samples_enqueued_in_one_go": 21,
                                         "certificate": dict(big, audited_sweeps=49, calibrations=1, tail_ratio_max=1.4, min_tau_over_sigma=40.0,
                                                             min_margin_tau_over_estimate=4.0, lattice_max_error=4e-4, lattice_max_over_sigma=7.0)}),
            "other_math": dict(big, value=4.15, ms_per_step=482.0, kernel="sdf_mlp_kernel", launch_ms=240.0, achieved=146.0, frac=0.93,
                               achieved_algorithmic=218.0),
            "roofline_marching_cubes": dict(big, bound="hbm", achieved=846.0, peak=8000.0, unit="GB/s", frac=0.106, chain_ms_both_volumes=0.165),
            "parity_in_run": dict(big, samples=list(range(20)), against_ordinary_sweeps_f16x3={"vertices_identical": 20},
                                  against_fp32_chain={"faces_identical": 20}, volumes_f16x3_vs_f32={"max_abs_difference": 5e-7, "sign_differences": 0},
                                  against_reference_runs=[{"V_F_equal_reference": True, "zoom_cube_bit_equal": True}] * 2),
            "other_configs": [dict(big, config="configs[0]: hand-only, N=64", tag="nerf3", grid=64, ms_per_step=1.0, V_F_equal_reference=True,
                                   sweeps={"refused_sweeps": 0}),
                              dict(big, config="grasp family ...", tag="grasp3", grid=256, ms_per_step=50.0, V_F_equal_reference=True,
                                   sweeps={"refused_sweeps": 0})],
            "cpu_baseline": dict(big, value=0.0063, unit="meshes/s", cores=16, kind="port", sample="s", seconds_per_sample=315.0, gpu_over_cpu=5000.0,
                                 checked_in_run={"zoom_cube_equal": True})}
    line = bench.short_line(full, "gpurun_out/x.json")
    assert len(json.dumps(line)) < 4000 and line["details_file"] == "gpurun_out/x.json"
    c, r = line["config"], line["roofline"]
    assert c["meshes_per_s_audited_sign_sweeps"] == 37.8 and c["ms_per_step_audited_sign_sweeps"] == 52.9 and c["meshes_per_s_fp32_mfma"] == 4.15
    assert c["audited_sustained_ms_per_step"] == 54.4 and c["audited_sustained_recalibrations"] == 2 and c["audited_sustained_refused_sweeps"] == 0
    assert c["other_configs_ms_per_step"] == {"configs0_hand_only_N64": 1.0, "grasp3_N256": 50.0}
    # what the driver's parser keeps is flat - no nested dict in `roofline`
    assert all(not isinstance(v, (dict, list)) for v in r.values())
    assert r["fp32_frac"] == 0.93 and r["fp32_launch_ms"] == 240.0 and r["fp32_peak"] == 157.3 and r["mc_frac"] == 0.106 and r["mc_chain_ms_both_volumes"] == 0.165
    assert r["audited_kernel"] == "sdf_mlp_f16p1_kernel" and r["audited_frac"] == 0.576 and r["audited_launch_ms"] == 24.6 and "note" not in r
    assert c["parity_in_run"]["reference_V_F_equal"] == 2 and c["parity_in_run"]["meshes_bit_identical_to_every_voxel_f16x3"] == 20
    assert c["sweeps"]["audited"] == 49 and c["sweeps"]["of"].startswith("audited leg") and line["cpu_baseline"]["cores"] == 16 and "blob" not in json.dumps(line)
