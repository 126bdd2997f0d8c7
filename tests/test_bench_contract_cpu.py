"""The bench.py output contract, checked on the committed record of the last GPU run (profiles/r02_bench.json) and on the
command line defaults -- no GPU needed.  Catches a renamed key, a wrong unit or an inconsistent roofline / value before the
driver does."""
import json
import math
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        lines = [l for l in f.read().strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "bench.py prints ONE JSON line"
    return json.loads(lines[0])


def test_bench_line_has_the_contract_keys():
    d = _record("r02_bench.json")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict),
                 ("cpu_baseline", dict)):
        assert k in d and isinstance(d[k], t), k
    assert "vs_baseline" in d and d["vs_baseline"] is None          # BASELINE.md holds no published number for this metric
    assert d["unit"] == base.get("unit", "frames/s") or d["unit"] == "frames/s"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f16" and d["data"] == "synthetic"
    assert d["n_gpus"] == 1 and "workload" in d["config"] and "configs[1]" in d["config"]["workload"]
    assert not any(k in d["config"] for k in ("model", "seq_len", "global_batch"))
    # value = frames of the clips / wall time: 16 frames per step
    frames = int(re.search(r"(\d+)f,", d["metric"]).group(1))
    assert math.isclose(d["value"], frames / (d["ms_per_step"] * 1e-3), rel_tol=1e-6)


def test_roofline_and_cpu_baseline_objects():
    d = _record("r02_bench.json")
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert (r["bound"] == "mfma") == (r["unit"] == "TFLOP/s")
    assert math.isclose(r["frac"], r["achieved"] / r["peak"], rel_tol=1e-9) and 0.0 < r["frac"] < 1.0
    assert r["peak"] == (2500.0 if r["bound"] == "mfma" else 8000.0)
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes"] > 0
    if r["traffic"] is not None:
        assert os.path.exists(os.path.join(ROOT, r["traffic_source"]["file"]))
    # the dominant kernel's time is a part of a step, not more
    assert r["avg_ms"] * r["launches"] / d.get("steps", 1) <= d["ms_per_step"] * 1.5
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["unit"] == d["unit"] and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert c["value"] < d["value"]
    # executed work never exceeds the machine: whole-loop MFMA fraction below 1
    assert 0.0 < d["mfma_frac_whole_loop"] < 1.0


def test_round4_line_measures_its_own_traffic_and_quotes_the_best_cpu_thread_count():
    d = _record("r04_bench.json")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "768" in base["metric"] and "768x768" in d["metric"] and "configs[1]" in d["config"]["workload"]
    assert d["n_gpus"] == d["n_ranks_seen"] == 1 and d["config"]["input_staging"].startswith("rank-local")
    frames = int(re.search(r"(\d+)f,", d["metric"]).group(1))
    assert math.isclose(d["value"], frames / (d["ms_per_step"] * 1e-3), rel_tol=1e-6)
    r = d["roofline"]
    assert r["bound"] == "mfma" and math.isclose(r["frac"], r["achieved"] / r["peak"], rel_tol=1e-9) and 0.25 < r["frac"] < 1.0
    # achieved = algorithmic FLOPs of the launch / its HIP-event average
    B, H, D, Lq, Lk = map(int, re.match(r"attention B=(\d+) H=(\d+) D=(\d+) Lq=(\d+) Lk=(\d+)", r["kernel"]).groups())
    assert math.isclose(r["achieved"], 4.0 * B * H * Lq * Lk * D / (r["avg_ms"] * 1e-3) / 1e12, rel_tol=1e-6)
    src = r["traffic_source"]
    assert "measured" in src and "separate passes" in src["measured"]                     # this run's own PMC passes, not a file
    assert math.isclose(r["traffic"], (2.0 * src["FETCH_SIZE_KiB"] + src["WRITE_SIZE_KiB"]) * 1024.0, rel_tol=1e-9)
    assert r["traffic"] >= r["algorithmic_bytes"] == 2.0 * B * H * D * (2 * Lq + 2 * Lk)
    c = d["cpu_baseline"]
    sweep = {int(k): v for k, v in c["thread_sweep_s_per_step_at_size"].items()}
    assert c["cores"] == min(sweep, key=sweep.get) and len(sweep) >= 3 and c["kind"] == "port"
    assert math.isclose(c["value"], 1.0 / (sweep[c["cores"]] * 20), rel_tol=2e-2) and c["value"] < d["value"]
    assert "extrapolated" in c["sample"]
    assert 0.0 < d["mfma_frac_whole_loop"] < 1.0 and d["e2e_frames_per_s"] < d["value"]


def test_other_records_are_labelled_with_their_config():
    assert "configs[2]" in _record("r02_bench_cfg2.json")["config"]["workload"]
    c4 = _record("r02_bench_cfg4.json")
    assert "configs[4]" in c4["config"]["workload"] and "1024x1024" in c4["metric"] and c4["peak_hbm_gb"] < 288
    two = _record("r02_bench_2rank_gloo_small.json")
    assert two["n_gpus"] == 2 and two["config"]["parallelism"] == "dp2"


def test_command_line_defaults():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300).stdout
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert re.search(r'"--gpus", type=int, default=1\b', src) and re.search(r'"--size", type=int, default=768\b', src)
    assert re.search(r'"--frames", type=int, default=16\b', src) and re.search(r'"--ddim-steps", type=int, default=20\b', src)


def test_json_line_is_last_even_when_stdout_and_stderr_are_captured_as_one_stream():
    """A native library's banner (RCCL prints one through the C-level stdio when its first communicator comes up) must not trail the JSON line:
    bench.py points fd 1 at stderr, flushes the C buffers before it writes the line to the real stdout, and nothing is left for exit."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--small", "--dry-run-cpu", "--size", "128", "--frames", "4",
           "--ddim-steps", "2"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600,
                       env=dict(os.environ, OMP_NUM_THREADS="2", MD_BENCH_TEST_BANNER="1"))
    assert r.returncode == 0, r.stdout[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.strip()]
    assert "stand-in banner" in r.stdout and lines[-1].startswith("{") and json.loads(lines[-1])["n_gpus"] == 1, r.stdout[-2000:]


def test_two_rank_launch_protocol_under_gloo():
    """bench.py exactly as the driver launches it for N = 2 (python -m torch.distributed.run ... bench.py --gpus 2 ...), on CPU
    with the gloo backend and the kernels replaced by a stand-in (--dry-run-cpu): rendezvous, rank-local staging AND the
    --scatter variant, barrier / max-over-ranks timing, gather on rank 0, ONE JSON line from rank 0 only, every rank exits 0
    through destroy_process_group()."""
    import socket
    for extra in ([], ["--scatter"]):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--small",
               "--dry-run-cpu", "--size", "128", "--frames", "4", "--ddim-steps", "2"] + extra
        # MD_BENCH_TEST_BANNER: every rank printf()s a line to the C-level stdout the way RCCL does when its first communicator comes up (seen on
        # MI355X: five lines that reach the file AFTER the JSON line, at exit); stdout must still hold the line and nothing else
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="2", MD_BENCH_TEST_BANNER="1"))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout
        assert r.stdout.strip().splitlines() == lines, r.stdout          # ... the LAST line of stdout is the first: a parser of either kind reads it
        assert r.stderr.count("stand-in banner") == 2, r.stderr[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["clips_gathered"] == 2 and d["steps"] == 2 and d["warmup"] == 1
        assert d["value"] is None and "dry-run" in d["data"] and d["config"]["parallelism"] == "dp2"
        assert d["config"]["input_staging"] == ("scatter" if extra else "rank-local")
        assert d["clip_means"][0] != d["clip_means"][1]                  # two different clips (seeds 100, 101) came back in rank order
        # diagnostics of a multi-rank run: one record per rank (own loop time, who computed where), the collective library, and the
        # scatter / gather of one batch timed on their own outside the timed region
        m = d["multi_gpu"]
        assert [r["rank"] for r in m["per_rank"]] == [0, 1] and len({r["pid"] for r in m["per_rank"]}) == 2
        assert all(r["own_elapsed_s"] > 0 and r["own_ms_per_step"] <= d["ms_per_step"] * 1.0001 for r in m["per_rank"])
        assert m["collectives"]["backend"] == "gloo" and m["collectives"]["world"] == 2 and m["mode"].startswith("clip data-parallel")
        assert m["comm_ms_outside_timed_region"]["gather_latents"] > 0 and ("scatter_clips" in m["comm_ms_outside_timed_region"]) == bool(extra)


def test_eight_rank_launch_protocol_under_gloo():
    """The driver's N = 8 command line (python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8 ...) on the CPU over gloo with
    the kernels replaced by a stand-in: 8 ranks rendezvous, every rank stages its own clip, ONE gather, one JSON line from rank 0 with 8
    different clips in rank order, every rank leaves through destroy_process_group()."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--small",
           "--dry-run-cpu", "--size", "128", "--frames", "4", "--ddim-steps", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["n_ranks_seen"] == 8 and d["clips_gathered"] == 8 and d["config"]["parallelism"] == "dp8"
    assert len(set(d["clip_means"])) == 8                               # eight different clips (seeds 100 .. 107) came back
    assert [r["local_rank"] for r in d["multi_gpu"]["per_rank"]] == list(range(8))


def test_cpu_quota_parsing(tmp_path):
    """bench.py's cpu_baseline reports min(affinity, cgroup CPU quota) as the usable cores: the GPU boxes of this pool run the container with
    cpu.max = "1600000 100000" (16 CPUs) on a 128-core host, which is the whole story behind "more than 16 threads are slower"."""
    sys.path.insert(0, ROOT)
    import bench
    (tmp_path / "cpu.max").write_text("1600000 100000\n")
    assert bench.cpu_quota(str(tmp_path)) == ("1600000 100000", 16.0)
    (tmp_path / "cpu.max").write_text("max 100000\n")
    assert bench.cpu_quota(str(tmp_path)) == ("max 100000", None)
    v1 = tmp_path / "v1" / "cpu"
    v1.mkdir(parents=True)
    (v1 / "cpu.cfs_quota_us").write_text("-1\n"); (v1 / "cpu.cfs_period_us").write_text("100000\n")
    assert bench.cpu_quota(str(tmp_path / "v1")) == ("-1", None)
    (v1 / "cpu.cfs_quota_us").write_text("800000\n")
    assert bench.cpu_quota(str(tmp_path / "v1")) == ("800000", 8.0)
    assert bench.cpu_quota(str(tmp_path / "nowhere")) == (None, None)


def test_round5_line_reports_the_usable_cores_of_its_cpu_baseline():
    d = _record("r05_bench.json")
    assert "768x768" in d["metric"] and "configs[1]" in d["config"]["workload"] and d["n_gpus"] == d["n_ranks_seen"] == 1
    frames = int(re.search(r"(\d+)f,", d["metric"]).group(1))
    assert math.isclose(d["value"], frames / (d["ms_per_step"] * 1e-3), rel_tol=1e-6) and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == "mfma" and math.isclose(r["frac"], r["achieved"] / r["peak"], rel_tol=1e-9) and 0.25 < r["frac"] < 1.0
    assert "measured" in r["traffic_source"] and r["traffic"] >= r["algorithmic_bytes"]
    c = d["cpu_baseline"]
    # cores = ALL usable cores = min(affinity, cgroup CPU quota); the line carries the evidence (affinity, quota, thread sweep) and says so
    assert c["kind"] == "port" and c["cores"] == c["usable_cores"] == min(c["affinity"], int(c["cgroup_quota_cpus"])) and c["value"] < d["value"]
    sweep = {int(k): v for k, v in c["thread_sweep_s_per_step_at_size"].items()}
    assert min(sweep, key=sweep.get) == c["single_process"]["threads"] and len(sweep) >= 3
    assert "quota" in c["sample"] and "extrapolated" in c["sample"] and str(c["cgroup_cpu_max"]) in c["sample"]
    assert math.isclose(c["value"], 1.0 / (c["single_process"]["s_per_frame_step"] * 20), rel_tol=2e-2)
    # the fused normalisations of round 5 are on the line: LayerNorm folded into q|k|v, GroupNorm inside proj_in
    labels = " ".join(s["label"] for s in d["top_launch_shapes"])
    assert " ln" in labels and d["kernel_families"]["layernorm"]["ms_per_clip"] < 80 and d["kernel_families"]["groupnorm"]["ms_per_clip"] < 105
    assert 0.0 < d["mfma_frac_whole_loop"] < 1.0 and d["e2e_frames_per_s"] < d["value"] and d["vae_ms_per_clip"] < 310


def test_round6_line_carries_the_diagnostics_the_first_multi_gpu_run_needs():
    d = _record("r06_bench.json")
    assert "768x768" in d["metric"] and "configs[1]" in d["config"]["workload"] and d["n_gpus"] == d["n_ranks_seen"] == 1 and d["scaling"] == "weak"
    frames = int(re.search(r"(\d+)f,", d["metric"]).group(1))
    assert math.isclose(d["value"], frames / (d["ms_per_step"] * 1e-3), rel_tol=1e-6) and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == "mfma" and math.isclose(r["frac"], r["achieved"] / r["peak"], rel_tol=1e-9) and "measured" in r["traffic_source"]
    # who computed where: one record per rank with the device's UUID and PCI bus id, its own loop time, the collective library
    m = d["multi_gpu"]
    assert len(m["per_rank"]) == 1 and m["n_distinct_gpus"] == 1 and m["gpu_aliasing"] is False
    me = m["per_rank"][0]
    assert me["uuid"] and re.match(r"[0-9a-f]{4}:[0-9a-f]{2}:[0-9a-f]{2}\.0$", me["pci_bus_id"]) and me["cus"] == 256 and me["hbm_gib"] > 250
    assert math.isclose(me["own_ms_per_step"], d["ms_per_step"], rel_tol=1e-3) and m["collectives"]["rccl_version"]
    # the VAE beside the loop, by config: absent face / hand guidance = 2F copies of one black frame, encoded once
    v = d["vae_by_config"]
    assert v["configs[1]"]["images"] == v["configs[2]"]["images"] == 3 * frames + 2 and v["configs[1]"]["encoded"] == frames + 3 and v["configs[2]"]["encoded"] == 3 * frames + 2
    assert v["configs[1]"]["ms"] < 0.7 * v["configs[2]"]["ms"] and d["vae_ms_per_clip"] == v["configs[1]"]["ms"]
    e = d["e2e_frames_per_s_by_config"]
    assert e["configs[2]"] < e["configs[1]"] < d["value"] and math.isclose(d["e2e_frames_per_s"], e["configs[1]"])
    # the CPU baseline shows its extrapolation: one step of 4 frames next to one step of 1 frame; value from the 1-frame step only within 10 %
    c = d["cpu_baseline"]
    lin = c["frames_linearity"]
    assert math.isclose(lin["ratio_to_linear"], lin["s_per_step_4_frames"] / (4 * lin["s_per_step_1_frame"]), rel_tol=2e-2)
    assert lin["within_10_percent"] == (abs(lin["ratio_to_linear"] - 1) <= 0.10)
    want = (1.0 / (lin["s_per_step_1_frame"] * 20)) if lin["within_10_percent"] else (4.0 / (lin["s_per_step_4_frames"] * 20))
    assert math.isclose(c["value"], want, rel_tol=2e-2) and c["cores"] == c["usable_cores"] and c["value"] < d["value"]
