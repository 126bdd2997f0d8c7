"""Turn a rocprofv3 (--kernel-trace --stats, rocpd sqlite output) database into the committed markdown summary.

    python profiles/summarize_rocprof.py gpurun_out/prof_x/name_results.db profiles/rNN_name.md [bench.json]

Per kernel: calls, total ms, average us, share.  For the kernel bench.py names in `roofline.kernel` the launches are
split into duration clusters (the same C++ kernel serves several shapes, e.g. self- and cross-attention) so that the
average of the dominant shape can be compared with the HIP-event average bench.py measured."""
import json
import sqlite3
import sys


def main(db, out, bench=None):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels group by name order by 3 desc"))
    total = sum(r[2] for r in rows)
    lines = ["# rocprofv3 --kernel-trace --stats summary", "", f"source: `{db}`  (total kernel time {total:.1f} ms)", "",
             "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for name, n, ms, us in rows[:25]:
        short = name if len(name) < 90 else name[:87] + "..."
        lines.append(f"| `{short}` | {n} | {ms:.2f} | {us:.2f} | {100 * ms / total:.2f} |")
    if bench:
        b = json.loads(open(bench).read().strip().splitlines()[-1])
        roof = b["roofline"]
        lines += ["", "## dominant kernel (bench.py `roofline`)", "", f"bench.py: `{roof['kernel']}` avg {roof['avg_ms']:.3f} ms over {roof['launches']} launches "
                  f"(HIP events), {roof['achieved']:.1f} {roof['unit']} = {100 * roof['frac']:.1f} % of peak", ""]
        fam = roof["kernel"].split(" ")[0]
        pat = {"attention": "attn%%_kernel<%s," % roof["kernel"].split("D=")[1].split(" ")[0] if "D=" in roof["kernel"] else "attn",
               "gemm": "gemm_kernel<false>", "conv3x3": "gemm_kernel<true>", "temporal_attention": "temporal_attn_kernel",
               "groupnorm": "gn_", "layernorm": "layernorm_kernel"}[fam]
        durs = sorted(r[0] / 1e6 for r in c.execute("select end-start from kernels where name like ?", (f"%{pat}%",)))
        target = roof["avg_ms"]
        near = [d for d in durs if 0.6 * target <= d <= 1.6 * target]
        lines += [f"rocprofv3: `{pat}` has {len(durs)} launches; {len(near)} of them fall within 0.6x..1.6x of the bench average "
                  f"(the dominant shape), average **{sum(near) / max(len(near), 1):.3f} ms** "
                  f"(profiled runs clock a few % lower than un-profiled ones, MI355X guide 'DVFS give-back').", ""]
        lines += ["## bench line of the profiled run", "", "```json", json.dumps({k: b[k] for k in ("value", "ms_per_step", "executed_tflop_per_clip", "mfma_frac_whole_loop", "roofline")}), "```"]
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:40]))


if __name__ == "__main__":
    main(*sys.argv[1:])
