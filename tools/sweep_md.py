#!/usr/bin/env python3
"""Render sweep JSON files (tools/sweep.py --out) as the markdown tables kept under profiles/."""
import json
import sys


def main():
    out, files = sys.argv[1], sys.argv[2:]
    recs = []
    for f in files:
        recs += json.load(open(f))["records"]
    groups = {}
    for r in recs:
        groups.setdefault((r["workload"], r["dtype"]), []).append(r)
    title = "# Kernel sweep" + (", " + sys.argv[0] if False else "") + " (one MI355X, synthetic twins, device time per launch from HIP events)"
    L = [title, "",
         "`GB/s` = algorithmic bytes `nnz*(V+4)+(m+1)*4+(n+m)*V` / time; `%` of the 8 TB/s HBM3E spec peak; `mem` = format "
         "footprint / CSR footprint.",
         "Rows with `auto` options are the engine's own choice for that format. cant/scircuit/pwtk (<= 140 MB) are "
         "Infinity-Cache resident after warm-up.", "",
         "Round 2 onward every handle runs on its own engine-placed x / y pair (csrc/placement.hip); round-1 tables were taken on "
         "torch-allocated vectors and carry the ~10 % placement lottery of that round (profiles/r02_placement.md).", ""]
    for (w, dt), rows in groups.items():
        best = min(rows, key=lambda r: r["ms"])
        L += ["", f"## {w} ({dt})", "", "| format | options | us/launch | GFLOP/s | GB/s | % peak | mem |", "|---|---|---|---|---|---|---|"]
        for r in rows:
            name = r["format"] + (" **best**" if r is best else "")
            L.append(f"| {name} | {r['opts'] if r['opts'] else 'auto'} | {r['ms'] * 1e3:.1f} | {r['gflops']:.0f} | {r['gbps']:.0f} | "
                     f"{100 * r['frac']:.1f} | {r['mem_ratio']:.2f} |")
    open(out, "w").write("\n".join(L) + "\n")


if __name__ == "__main__":
    main()
