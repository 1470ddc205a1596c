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
    L = ["# Kernel sweep, round 1 (one MI355X, synthetic twins, device time per launch from HIP events)", "",
         "`GB/s` = algorithmic bytes `nnz*(V+4)+(m+1)*4+(n+m)*V` / time; `%` of the 8 TB/s HBM3E spec peak; `mem` = format "
         "footprint / CSR footprint.",
         "Rows with `auto` options are the engine's own choice for that format. cant/scircuit/pwtk (<= 140 MB) are "
         "Infinity-Cache resident after warm-up.", "",
         "Box-to-box spread of the pool is about 10 % (the same binary on the nlpkkt240 twin: 1485 .. 1680 us): compare rows "
         "within one table, not across files.", ""]
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
