#!/usr/bin/env python3
"""Turn the PMC passes of tools/collect_traffic.sh into profiles/traffic_<round>.json (read by bench.py).

HBM bytes per launch of the dominant kernel:
  read  = 32*RDREQ_32B + 128*RDREQ_128B + 64*(RDREQ - RDREQ_32B - RDREQ_128B)      (TCC_EA0 request counters, exact sizes)
          cross-check: FETCH_SIZE [KiB] counts every request as 64 B on gfx950, i.e. exactly half of a 128-B stream
          (MI355X_MICROARCH.md, HBM section) -> 2*FETCH_SIZE*1024 is reported beside it;
  write = WRITE_SIZE [KiB] * 1024.
Infinity-Cache hits are counted by these fabric-side counters, so for matrices that fit the 256 MiB cache the number is
"bytes through the L2's memory side", not DRAM bytes.
"""
import csv, glob, json, os, sys, collections


def kernel_means(d):
    acc = collections.defaultdict(list)
    dur = {}
    name = ""
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r["Kernel_Name"]
                # the SpMV kernels only: not the fix-up / search / expansion helpers, not the conversion kernels
                if "spmv::" not in k or not any(t in k for t in ("csr_scalar_kernel", "csr_vector_kernel", "csr_vector_multi_kernel",
                        "csr_stream", "csr_window_kernel", "merge_kernel", "sell_kernel", "sell_delta", "sell_window_kernel", "sell_window_sym_kernel", "sell_wide_kernel",
                        "coo_kernel", "coo_blocked_kernel")):
                    continue
                name = k.split("(")[0].replace("void spmv::", "")
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[(f, r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out = {c: sum(v) / len(v) for c, v in acc.items()}
    out["_kernel"] = name
    out["_duration_us"] = sum(dur.values()) / max(len(dur), 1) / 1e3
    return out


def main():
    root, outp = sys.argv[1], sys.argv[2]
    records = []
    for tagdir in sorted(glob.glob(os.path.join(root, "*"))):
        tag = os.path.basename(tagdir)
        m = {}
        for p in sorted(glob.glob(os.path.join(tagdir, "pass*"))):
            km = kernel_means(p)
            durs = m.get("_durs", [])
            durs.append(km.pop("_duration_us"))
            m.update(km)
            m["_durs"] = durs
        if "FETCH_SIZE" not in m:
            continue
        parts = tag.split("_")
        rd = m.get("TCC_EA0_RDREQ_sum")
        if rd is not None:
            r32, r128 = m.get("TCC_EA0_RDREQ_32B_sum", 0.0), m.get("TCC_EA0_RDREQ_128B_sum", 0.0)
            read_bytes = 32 * r32 + 128 * r128 + 64 * (rd - r32 - r128)
        else:
            read_bytes = 2 * m["FETCH_SIZE"] * 1024
        write_bytes = m.get("WRITE_SIZE", 0.0) * 1024
        rec = dict(tag=tag, kernel=m.get("_kernel"), kernel_us_under_profiler=sum(m["_durs"]) / len(m["_durs"]),
                   fetch_size_kib=m["FETCH_SIZE"], write_size_kib=m.get("WRITE_SIZE"),
                   read_bytes_from_request_sizes=read_bytes, read_bytes_2x_fetch_size=2 * m["FETCH_SIZE"] * 1024,
                   write_bytes=write_bytes, hbm_bytes_per_launch=int(read_bytes + write_bytes),
                   counters={k: v for k, v in m.items() if not k.startswith("_")})
        records.append(rec)
        print(tag, rec["kernel"], "read %.3f GB (2xFETCH %.3f GB) write %.3f GB" % (read_bytes / 1e9, rec["read_bytes_2x_fetch_size"] / 1e9, write_bytes / 1e9))
    # what ran (tools/run_one.py --meta): bench.py attaches a record only to the same kernel, format name and kernel sources
    for rec in records:
        meta = os.path.join(root, rec["tag"], "meta.json")
        if os.path.exists(meta):
            with open(meta) as fh:
                m = json.load(fh)
            if m.get("jitter") or m.get("modes_off"):
                rec["variant"] = {"jitter": m.get("jitter", 0.0), "index_modes_off": m.get("modes_off", 0)}      # never matches a bench line
                m["scale"] = -1.0
            rec["stored_bytes_per_nnz"] = round(m.get("stored_bytes_per_nnz", 0.0), 4)
            rec.update(workload=m["workload"], format=m["format"], dtype=m["dtype"], opts=m["opts"], scale=m.get("scale", 1.0),
                       format_name=m["format_name"], kernel_src_sha=m["kernel_src_sha"], algorithmic_bytes=m["algorithmic_bytes"],
                       us_per_launch_unprofiled=m["us_per_launch"])
            rec["traffic_over_algorithmic"] = round(rec["hbm_bytes_per_launch"] / m["algorithmic_bytes"], 4)
            hit, miss = rec["counters"].get("TCC_HIT_sum"), rec["counters"].get("TCC_MISS_sum")
            if hit is not None and miss:
                rec["l2_hit_rate"] = round(hit / (hit + miss), 4)
    with open(outp, "w") as f:
        json.dump(dict(records=records), f, indent=1)


if __name__ == "__main__":
    main()
