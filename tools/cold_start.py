"""N ranks building the full-width UNet pair side by side on a COLD synthetic-weight cache -- how the first 8-GPU run of bench.py starts
(VERDICT r04 item 7).  One writer (LOCAL_RANK 0) publishes the fp16 cache atomically while the other N - 1 ranks synthesise the same 2.2 G
parameters beside it; then the same N ranks start again on the now WARM cache.  Reports, per phase: wall time, per-rank peak RSS, and that
every rank ended up with identical weights (checksums).  CPU only (device="cpu"): the host side is what is being measured.

    python tools/cold_start.py --ranks 8 [--small] [--out profiles/r05_cold_start_8rank.json]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, os, resource, sys, time
sys.path.insert(0, os.environ["MD_ROOT"])
import torch
torch.set_num_threads(max(1, (os.cpu_count() or 1) // int(os.environ["WORLD_SIZE"])))      # bench.py's cap for N ranks on one host
from mikudance_amd.selftest import build_models
geom = None if os.environ["MD_SMALL"] == "1" else dict(block_out_channels=(320, 640, 1280, 1280), cross_attention_dim=768)
t0 = time.time()
ref, den, _, _ = build_models(geom=geom, device="cpu", keep_state_dicts=False)
dt = time.time() - t0
rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2 ** 20        # GiB (ru_maxrss is in KiB on Linux)
cs = [float(sum(v.double().abs().sum() for v in m.state_dict().values())) for m in (ref, den)]
n = sum(v.numel() for m in (ref, den) for v in m.state_dict().values())
print("RANK " + json.dumps({"rank": int(os.environ["LOCAL_RANK"]), "build_s": round(dt, 1), "peak_rss_gib": round(rss, 2), "checksums": cs, "params": n}))
"""


def phase(ranks, small, cache_dir):
    env = dict(os.environ, MD_ROOT=ROOT, MD_SMALL="1" if small else "0", MD_SYNTH_CACHE_DIR=cache_dir, WORLD_SIZE=str(ranks))
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, "-c", WORKER], env=dict(env, LOCAL_RANK=str(r), RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(ranks)]
    recs = []
    for p in procs:
        out, err = p.communicate(timeout=3600)
        assert p.returncode == 0, err[-3000:]
        recs.append(json.loads([l for l in out.splitlines() if l.startswith("RANK ")][0][5:]))
    return {"wall_s": round(time.time() - t0, 1), "ranks": sorted(recs, key=lambda r: r["rank"])}


def run(ranks=8, small=False):
    with tempfile.TemporaryDirectory(prefix="mdance_cold_") as d:
        os.chmod(d, 0o700)
        cold = phase(ranks, small, d)
        files = sorted(f for f in os.listdir(d))
        warm = phase(ranks, small, d)
    rec = {"ranks": ranks, "width": "reduced" if small else "full (SD-1.5 geometry, 2.2 G parameters)", "host_cpus": os.cpu_count(),
           "cold_cache": cold, "cache_files_after_cold_phase": files, "warm_cache": warm}
    rec["peak_rss_gib_sum_cold"] = round(sum(r["peak_rss_gib"] for r in cold["ranks"]), 1)
    rec["peak_rss_gib_sum_warm"] = round(sum(r["peak_rss_gib"] for r in warm["ranks"]), 1)
    return rec


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    r = run(a.ranks, a.small)
    print(json.dumps(r))
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(r, fh, indent=1)
