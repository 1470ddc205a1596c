#!/usr/bin/env python3
"""bench.py — SpMV throughput of the MI355X engine on the BASELINE.json configurations.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU (RCCL). Rank 0 prints ONE JSON line.

A "step" is one y = A*x over the whole matrix with the matrix, x and y already resident in HBM:
  N = 1 : one launch of the selected HIP kernel through the C ABI (include/spmv_mi355x.h);
  N > 1 : row-partitioned SpMV with the x exchange forced every step (spmv-research_amd/python/bench_multi.py).
          Total work is fixed as N grows -> "scaling": "strong".

Headline workload: 'nlpkkt240' (config 5 of BASELINE.json: 28.0 M rows, ~770 M non-zeros, fp64) — the largest
single-GPU configuration, the one the multi-GPU target is quoted on, and far larger than the 256 MiB Infinity Cache,
so its algorithmic GB/s is real HBM traffic. At N = 1 the same run also times configs 1-4 (cant, scircuit, pwtk fp32,
soc-LiveJournal1) with the kernel BASELINE.json NAMES for them and with the engine's best kernel, and reports them
under "configs" — so the driver-run record carries all five fractions.
The matrices are synthetic twins (no SuiteSparse file exists in the reference tree and there is no network; a real
file under $SPMV_MTX_DIR/<name>.mtx is loaded through the product's own Matrix-Market reader instead).

metric/value: GFLOP/s = 2*nnz / t (true stored nnz; the reference's printed GFLOPS is ~2x inflated for general
matrices). ONE clock for value, hbm_pct_of_peak and every roofline fraction: the wall time per step of the synchronize
bracket around the K timed launches (ms_per_step); the HIP-event time of the same launches, recorded on the launch stream, is kept
beside it as roofline.kernel_ms. roofline.achieved: ALGORITHMIC bytes B_alg = nnz*(sizeof(V)+4) + (m+1)*4 + (n+m)*sizeof(V) per
launch / ms_per_step. roofline.frac == roofline.frac_algorithmic is that CSR-normalised figure (a format that stores fewer bytes
than CSR scores above what it moves); roofline.frac_hbm_measured is the PMC traffic of the same kernel build / time / peak (only
when profiles/traffic_*.json holds a record taken with these kernel sources and this format); roofline.frac_floor is the same
headline with every index-free mode of the format switched off (what a matrix without the twin's translation invariance gets).

Order of events at N = 1 (everything before the timed steps is reported under setup_s, none of it is timed): generate the twin ->
convert (GPU) -> first use of the handle's own x / y: bench.py asks for engine-placed vectors (--placement: opts.placement = 3 for
the headline handle, 1 for the small configs): the first handle of the process walks the device's free memory once for two to four
vector pools in regions of different class, every handle then takes its vectors from the pool its kernel runs fastest in, and level 3
also moves a matrix array the driver laid across two regions (csrc/placement.hip; worth 13-17 % on this device,
profiles/r02_placement.md, profiles/r03_placement_walk.txt; what the walk found is reported as "placement")
-> W warm-up launches -> settle (batches until two agree within 0.5 %, >= 500 launches; the reference driver warms GPU kernels
with 1000 calls) -> EXACTLY K timed launches between HIP events, inside a synchronize bracket (K = 1000 by default: 1.2 s of the
headline kernel; with K < 200 five such windows are timed and the median one is reported with the spread).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "spmv-research_amd", "python"))

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md; ~6.3 TB/s achievable)
INFINITY_CACHE_BYTES = 256 << 20
WORKLOADS = ("cant", "scircuit", "pwtk", "soc-LiveJournal1", "nlpkkt240")
# the engine's best kernel per workload (what `bench.py --workload W` times when --format is not given)
DEFAULT_FORMAT = {"nlpkkt240": "sell_c_sigma", "cant": "sell_c_sigma", "pwtk": "csr_stream",
                  "scircuit": "csr_vector", "soc-LiveJournal1": "coo"}
DEFAULT_DTYPE = {"pwtk": "f32"}
# options that go with a default format (only when --format is not given)
DEFAULT_OPTS = {"soc-LiveJournal1": {"col_blocks": -1}}        # column-blocked COO
# the kernel BASELINE.json's `configs` NAME for each workload (format, options)
NAMED_KERNEL = {
    "cant": ("csr_vector", {}),                           # config 1: CSR fp64 (the CPU path's format)
    "scircuit": ("csr_vector", {"lanes_per_row": 64, "rows_per_group": 2}),   # config 2: CSR-Vector, one wavefront per row (two rows
                                                                             # of a wavefront in flight: 27.9 vs 40.5 us)
    "pwtk": ("sell_c_sigma", {}),                         # config 3: SELL-C-sigma fp32
    "soc-LiveJournal1": ("csr_merge", {}),                # config 4: merge-based CSR = the literal CSR-order merge path (merge.cpp:256-319)
    "nlpkkt240": ("csr_stream", {}),                      # config 5: row-partitioned CSR (plain CSR storage)
}
# further kernels reported beside the named and the best one ("also"): on the graph matrix the column-blocked layout with merge-path
# balanced row ranges (what `csr_merge` runs when asked for it: col_blocks = -1) beside the best one (the same layout under `coo`)
ALSO_KERNELS = {"soc-LiveJournal1": [("csr_merge", {"col_blocks": -1})], "scircuit": [("csr_vector", {"lanes_per_row": 64})],
                "cant": [("csr_stream", {})]}            # cant: the CSR-storage kernel with the x window in LDS (16-bit columns), beside the named csr_vector
# configs 1-4: timed after the headline at N = 1
SMALL_CONFIGS = ("cant", "scircuit", "pwtk", "soc-LiveJournal1")
# the two banded FEM twins also get a symmetric-storage leg
SYMMETRIC_CONFIGS = ("cant", "pwtk")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000,
                    help="timed steps; the default covers ~1.3 s of the headline kernel, which alternates between two levels 5.5 %% apart in stretches of 0.1-0.4 s (profiles/r02_placement.md §6): a shorter window measures whichever level it lands in")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="nlpkkt240",
                    help="cant | scircuit | pwtk | soc-LiveJournal1 | nlpkkt240 (synthetic twins)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (tests only; invalid as a result)")
    ap.add_argument("--format", default=None, help="csr_scalar|csr_vector|csr_stream|csr_merge|sell_c_sigma|coo")
    ap.add_argument("--dtype", default=None, choices=["f64", "f32"])
    ap.add_argument("--opt", action="append", default=[], help="key=value for spmv_mi355x_opts (repeatable)")
    ap.add_argument("--lanes-per-row", type=int, default=0)
    ap.add_argument("--sell-c", type=int, default=0)
    ap.add_argument("--sell-sigma", type=int, default=0)
    ap.add_argument("--merge-items", type=int, default=0)
    ap.add_argument("--nontemporal", type=int, default=0)
    ap.add_argument("--col-blocks", type=int, default=0, help="coo / csr_merge: -1 = column-blocked with ~384 KiB blocks of x, >0 = that many blocks")
    ap.add_argument("--xcd-remap", type=int, default=0)
    ap.add_argument("--jitter", type=float, default=0.0,
                    help="nlpkkt240 only: fraction of rows whose off-diagonal columns are perturbed by up to +-jitter-span "
                         "(sensitivity of the compressed-index format to the twin's regularity; invalid as the headline)")
    ap.add_argument("--jitter-span", type=int, default=3)
    ap.add_argument("--index-modes-off", type=int, default=0,
                    help="SELL delta layout, sensitivity experiment: 1 = no affine slices, 2 = no per-slice lane offsets, 3 = neither "
                         "(every slice stores 8/16-bit deltas per lane); invalid as the headline")
    ap.add_argument("--configs", default="auto", choices=["auto", "on", "off"],
                    help="time configs 1-4 (named + best kernel) after the headline; auto = on for the default headline at N = 1")
    ap.add_argument("--configs-steps", type=int, default=300)
    ap.add_argument("--overlap", type=int, default=1, help="N>1: overlap the x exchange with the rows/columns that need no remote x")
    ap.add_argument("--exchange", default="auto", choices=["auto", "allgather", "p2p"],
                    help="N>1: RCCL allgather of the padded x slices, or grouped send/recv of only the sub-ranges each row "
                         "block reads (same result); auto = time both in the warm-up and keep the faster")
    ap.add_argument("--partition", default="auto", choices=["auto", "rows", "graph"],
                    help="N>1: 'rows' = the reference's nnz-balanced contiguous row blocks of A as it is; 'graph' = the same balance "
                         "cut out of a breadth-first order of the matrix graph; 'auto' = whichever makes the busiest rank read "
                         "fewer remote x entries")
    ap.add_argument("--halo", default="auto", choices=["auto", "alltoall", "p2p"],
                    help="N>1, graph partition: how the packed halo segments move (auto = all_to_all_single if it validates, else p2p)")
    ap.add_argument("--variants", default="auto", choices=["auto", "both", "one"],
                    help="N>1: 'both' = time the north-star scheme (row blocks + allgather(x)) AND the auto choice in one run and "
                         "report both under \"variants\" (value = the faster); auto = both unless --partition/--exchange is given")
    ap.add_argument("--host-chunk-nnz", type=int, default=48_000_000,
                    help="N>1: a rank generates and converts its rows in pieces of at most this many non-zeros (one handle per piece), so its "
                         "host copy of the matrix never exceeds a piece; 0 = the whole block at once")
    ap.add_argument("--piece-handles", action="store_true",
                    help="N>1: one handle per piece instead of one handle converted from the pieces in device memory (tests)")
    ap.add_argument("--placement", type=int, default=3, choices=[0, 1, 3],
                    help="opts.placement of the headline handle: 0 = plain allocations, 1 = vectors from the engine's pools, 3 = pools + the search over "
                         "the matrix arrays (a few seconds of setup; boxes differ in where the driver puts a process's first allocations: "
                         "profiles/r03_placement_walk.txt). The small configs always run with 1.")
    ap.add_argument("--idle-after-placement", type=float, default=0.0,
                    help="extra seconds without launches after the engine has placed the vectors (the driver clears the ballast the one "
                         "walk returned in the background; the settle phase runs through it; profiles/r02_placement.md §6)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the headline handle: no floor / jitter legs, no configs 1-4 (the profiled command of tools/gpu_profiles.sh: every launch of "
                         "the headline kernel the profiler then sees is this handle's, so its average can be held against roofline.kernel_ms)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=8.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL)")
    ap.add_argument("--force-multi", action="store_true",
                    help="run the N>1 code path with whatever WORLD_SIZE is, 1 included (tests: the only way to drive the RCCL calls on a one-GPU box)")
    return ap.parse_args(argv)


def algorithmic_bytes(m, n, nnz, vbytes):
    return nnz * (vbytes + 4) + (m + 1) * 4 + (n + m) * vbytes


def kkt_edge(scale):
    return max(4, int(round(240 * scale ** (1.0 / 3.0))))


_SRC_SHA = None


def kernel_source_sha():
    """Fingerprint of the device code (csrc/*.hip, *.hpp): a PMC traffic record is only attached to a run whose kernels are
    the ones the counters were collected on."""
    global _SRC_SHA
    if _SRC_SHA is None:
        h = hashlib.sha256()
        d = os.path.join(ROOT, "spmv-research_amd", "csrc")
        for f in sorted(os.listdir(d)):
            if f.endswith((".hip", ".hpp")):
                h.update(f.encode())
                with open(os.path.join(d, f), "rb") as fh:
                    h.update(fh.read())
        _SRC_SHA = h.hexdigest()[:16]
    return _SRC_SHA


def load_traffic(workload, format_name, dtype, kernel, src_sha=None, pdir=None):
    """HBM bytes per SpMV from the rocprofv3 PMC passes (tools/collect_traffic.py -> profiles/traffic_*.json), or None.

    A record is used only when it was collected on the SAME kernel (name), the SAME converted format (the handle's
    format_name encodes the options and block counts) and — when the record carries one — the same kernel sources."""
    best = None
    pdir = pdir or os.path.join(ROOT, "profiles")
    if not os.path.isdir(pdir):
        return None
    src_sha = src_sha or kernel_source_sha()
    for f in sorted(os.listdir(pdir)):
        if not (f.startswith("traffic_") and f.endswith(".json")):
            continue
        try:
            with open(os.path.join(pdir, f)) as fh:
                recs = json.load(fh).get("records", [])
        except Exception:
            continue
        for rec in recs:
            if rec.get("workload") != workload or rec.get("dtype") != dtype or rec.get("scale", 1.0) != 1.0:
                continue
            if rec.get("kernel", "").split("<")[0] != kernel or rec.get("format_name") != format_name:
                continue
            if rec.get("kernel_src_sha") != src_sha:
                continue
            v = rec.get("hbm_bytes_per_launch")
            if v is not None:
                # a format that needs several dispatches of its kernel per SpMV records the per-dispatch figure
                best = int(v * rec.get("dispatches_per_spmv", 1))
    return best


class QuietStdout:
    """Send the process's fd 1 to /dev/null for a block (C-level printf of the reference library included)."""

    def __enter__(self):
        import ctypes
        self.libc = ctypes.CDLL(None)
        sys.stdout.flush()
        self.libc.fflush(None)
        self.saved = os.dup(1)
        dn = os.open(os.devnull, os.O_WRONLY)
        os.dup2(dn, 1)
        os.close(dn)
        return self

    def __exit__(self, *exc):
        self.libc.fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def cpu_share():
    """cores this job may use: the cgroup CPU quota when there is one, else the affinity mask"""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except (OSError, ValueError):
        pass
    return n


def collect_opts(args, workload):
    opts = dict(DEFAULT_OPTS.get(workload, {})) if not args.format else {}
    if args.col_blocks:
        opts["col_blocks"] = args.col_blocks
    for k, v in (("lanes_per_row", args.lanes_per_row), ("sell_c", args.sell_c), ("sell_sigma", args.sell_sigma),
                 ("merge_items", args.merge_items), ("nontemporal", args.nontemporal), ("xcd_remap", args.xcd_remap)):
        if v:
            opts[k] = v
    for kv in args.opt:
        k, v = kv.split("=")
        opts[k] = int(v)
    return opts


def load_workload(H, workload, scale, jitter=0.0, jitter_span=3):
    """The CSR of a BASELINE.json workload: a real Matrix-Market file when $SPMV_MTX_DIR holds one (never fetched), else the
    synthetic twin. Returns (A, data) with data = "synthetic" | "file:<path>"."""
    mdir = os.environ.get("SPMV_MTX_DIR")
    if mdir and scale == 1.0 and not jitter:
        for ext in (".mtx", ".mtx.gz", ".mtx.zst", ".tar.gz"):
            p = os.path.join(mdir, workload + ext)
            if os.path.exists(p):
                info, rp, ci, va = H.mtx_to_csr(p)
                return dict(m=info["m"], n=info["n"], nnz=len(ci), row_ptr=rp, col_idx=ci, values=va), "file:" + p
    if workload == "nlpkkt240":
        A = H.gen_kkt(kkt_edge(scale))
        if jitter:
            H.jitter_columns(A, jitter, jitter_span)
        return A, "synthetic"
    return H.gen_named(workload, scale), "synthetic"


def sampled_row_check(yh, rp, ci, va, x_host, np_dtype, col_map=None, count=2000):
    """max over sampled rows of |y_i - sum_j a_ij x_j| / sum_j |a_ij x_j| against host dot products in fp64."""
    lm = len(rp) - 1
    samp = np.unique(np.random.default_rng(1).integers(0, max(lm, 1), count)) if lm > 0 else np.array([], np.int64)
    xg = x_host.astype(np.float64)
    max_rel = 0.0
    for i in samp:
        cols = ci[rp[i]:rp[i + 1]].astype(np.int64)
        if col_map is not None:
            cols = col_map(cols)
        vals = va[rp[i]:rp[i + 1]].astype(np_dtype).astype(np.float64)
        ref = float(np.dot(vals, xg[cols]))
        den = float(np.dot(np.abs(vals), np.abs(xg[cols]))) or 1.0
        max_rel = max(max_rel, abs(ref - float(yh[i])) / den)
    return max_rel, samp


def cold_launches(torch, M, xp, yp, sp, count=15, flush_bytes=1 << 30):
    """Isolated launches with the caches flushed in between: a 1 GiB buffer (4x the 256 MiB Infinity Cache) is overwritten on the
    launch stream before every single event-timed launch — the GPU analogue of the reference driver's CLEAR_CACHES
    (bench_spmv.cpp:331-348). These times include the ramp-up and drain of one kernel on an idle chip. Median ms."""
    flush = torch.empty(flush_bytes, dtype=torch.uint8, device="cuda")
    ts = []
    for i in range(count):
        flush.fill_(i & 0xff)
        ts.append(M.time_device(xp, yp, 1, sp))
    del flush
    return float(np.median(ts))


def time_handle(E, torch, A, fmt, dts, opts, steps, warmup, x_host=None, min_warm_seconds=0.0, idle_after_placement=0.0, windows=1,
                cold=False):
    """Build one handle, run warm-up + `windows` x `steps` back-to-back launches, each window timed by HIP events on the launch
    stream inside a synchronize bracket, check sampled rows. Returns a dict with the measured figures (the MEDIAN window's) and leaves
    nothing on the device."""
    np_dtype = np.float64 if dts == "f64" else np.float32
    t_dtype = torch.float64 if dts == "f64" else torch.float32
    vbytes = 8 if dts == "f64" else 4
    m, n, nnz = A["m"], A["n"], A["nnz"]
    if x_host is None:
        x_host = np.random.default_rng(14).uniform(-1.0, 1.0, n).astype(np_dtype)
    t0 = time.time()
    opts = dict(opts)
    opts.setdefault("placement", 1)          # vectors from the engine's pools (csrc/placement.hip): opt-in, and bench.py opts in
    opts.setdefault("placement_budget_gib", 160)   # this process owns the device: the one walk may go deep before it gives up
    M = E.Matrix(A["row_ptr"], A["col_idx"], A["values"], m, n, fmt, np_dtype, **opts)
    t_conv = time.time() - t0
    # the handle's own x / y pair: allocated and PLACED by the engine (csrc/placement.hip — where y lives relative to the value
    # array is worth 12 % on this device, profiles/r02_placement.md); the search runs here, outside the timed region
    t0 = time.time()
    M.upload_x(x_host)
    t_place = time.time() - t0
    if t_place > 0.15 and idle_after_placement > 0:
        # the first handle of the process has just walked the device's free memory for the vector pools and returned the ballast; the
        # driver clears freed memory in the background, and kernels launched into that alternate between their normal level and one
        # ~5 % slower (time series in profiles/r02_placement.md §6). The settle phase below runs through it; --idle-after-placement
        # adds an idle wait on top (default 0). Not part of any timed region; reported in setup_s.
        time.sleep(idle_after_placement)
        t_place += idle_after_placement
    xp, yp = M.x_device(), M.y_device()
    M.upload_y(np.ones(m, np_dtype))                               # driver canary (bench_spmv.cpp:606-609)
    sp = torch.cuda.current_stream().cuda_stream
    for _ in range(warmup):
        M.spmv_device(xp, yp, 0, sp)
    torch.cuda.synchronize()
    # settle: the reference driver warms GPU kernels with 1000 untimed calls (bench_spmv.cpp:287-294). Here: batches of launches for
    # min_warm_seconds, then at least 500 launches and on until two batches in a row agree within 0.5 % (at most 1500) — after the
    # host-to-device uploads above the first few hundred launches run up to 6 % slower (profiles/r02_placement.md §6)
    t_w = time.time()
    batch, settle = max(min(steps, 100), 20), []
    while time.time() - t_w < min_warm_seconds:
        M.time_device(xp, yp, batch, sp)
    while len(settle) * batch < 1500:
        settle.append(M.time_device(xp, yp, batch, sp))
        if len(settle) * batch >= 500 and abs(settle[-1] - settle[-2]) <= 0.005 * settle[-1]:
            break
    if os.environ.get("SPMV_BENCH_VERBOSE"):
        print("[bench] settle batches (ms): " + " ".join(f"{v:.4f}" for v in settle), file=sys.stderr)
    wins = []
    for _ in range(max(1, windows)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k_ms = M.time_device(xp, yp, steps, sp)                   # HIP events on the launch stream
        torch.cuda.synchronize()
        wins.append(((time.perf_counter() - t0) / steps * 1e3, k_ms))
    wins.sort()
    wall_ms, kernel_ms = wins[len(wins) // 2]
    cold_ms = cold_launches(torch, M, xp, yp, sp) if cold else None
    yh = M.download_y().astype(np.float64)
    max_rel, samp = sampled_row_check(yh, A["row_ptr"], A["col_idx"], A["values"], x_host, np_dtype)
    tol = 1e-12 if dts == "f64" else 1e-5
    if not (max_rel <= tol) or not np.all(yh == yh):
        raise SystemExit(f"bench sanity check failed for {M.format_name}: sampled rows differ from the host dot products (max {max_rel})")
    ki = M.kernel_info()
    B = algorithmic_bytes(m, n, nnz, vbytes)
    out = dict(format_name=M.format_name, kernel=ki["name"], kernel_ms=kernel_ms, wall_ms=wall_ms, convert_s=t_conv,
               algorithmic_bytes=B, gbps=B / (kernel_ms * 1e-3) / 1e9, gflops=2.0 * nnz / (kernel_ms * 1e-3) / 1e9,
               mem_footprint=M.mem_footprint, check=max_rel, yh=yh, samp=samp, x_host=x_host, place_s=t_place,
               wall_ms_windows=[round(w[0], 6) for w in wins], cold_ms=cold_ms)
    M.close()
    return out


def roofline_record(workload, dts, t, with_traffic=True):
    """The "roofline" object of one timed kernel. ONE clock for every fraction: the wall time per step of the synchronize bracket
    (= ms_per_step of the line); the HIP-event time of the same launches is kept beside it as kernel_ms."""
    traffic = load_traffic(workload, t["format_name"], dts, t["kernel"]) if with_traffic else None
    resident = t["algorithmic_bytes"] < INFINITY_CACHE_BYTES
    ms = t["wall_ms"]
    gbps = t["algorithmic_bytes"] / (ms * 1e-3) / 1e9
    frac = gbps / HBM_PEAK_GBPS
    return {"bound": "hbm", "achieved": round(gbps, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(frac, 4),
            # CSR-normalised ("algorithmic") bytes: a format that stores fewer bytes than CSR scores above what it moves
            "frac_algorithmic": round(frac, 4),
            # bytes the PMC passes saw move (profiles/traffic_*.json, stored per kernel-source fingerprint, not re-counted in this run)
            "frac_hbm_measured": None if traffic is None else round(traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "traffic": traffic, "traffic_source": None if traffic is None else "profiles/traffic_*.json (rocprofv3 PMC pass of the same kernel sources)",
            "kernel": t["kernel"], "kernel_ms": round(t["kernel_ms"], 6), "ms": round(ms, 6),
            "algorithmic_bytes_per_launch": int(t["algorithmic_bytes"]),
            # a working set under the 256 MiB Infinity Cache is served on-die when launched back to back: its GB/s is a cache
            # rate, not an HBM rate
            "cache_resident": bool(resident)}


def run_small_configs(E, torch, H, args):
    """Configs 1-4 of BASELINE.json: the NAMED kernel and the engine's best kernel on each twin, a few hundred launches each."""
    out = []
    for w in SMALL_CONFIGS:
        t0 = time.time()
        A, data = load_workload(H, w, args.scale)
        t_gen = time.time() - t0
        dts = DEFAULT_DTYPE.get(w, "f64")
        nf, no = NAMED_KERNEL[w]
        bf, bo = DEFAULT_FORMAT[w], DEFAULT_OPTS.get(w, {})
        steps = args.configs_steps if A["nnz"] < 20_000_000 else max(20, args.configs_steps // 6)
        warm = 0.1 if A["nnz"] < 20_000_000 else 0.0
        resident = algorithmic_bytes(A["m"], A["n"], A["nnz"], 8 if dts == "f64" else 4) < INFINITY_CACHE_BYTES
        # cache-resident matrices also get a COLD figure (caches flushed before every isolated launch): the warm one is a cache rate
        named = time_handle(E, torch, A, nf, dts, dict(no), steps, 20, min_warm_seconds=warm, cold=resident)
        best = time_handle(E, torch, A, bf, dts, dict(bo), steps, 20, x_host=named["x_host"], min_warm_seconds=warm, cold=resident)
        rn, rb = roofline_record(w, dts, named), roofline_record(w, dts, best)
        also = []
        for af, ao in ALSO_KERNELS.get(w, []):
            t = time_handle(E, torch, A, af, dts, dict(ao), steps, 20, x_host=named["x_host"], min_warm_seconds=warm)
            also.append({"kernel": t["kernel"], "format": t["format_name"], "ms": round(t["wall_ms"], 6), "kernel_ms": round(t["kernel_ms"], 6),
                         "frac": roofline_record(w, dts, t, with_traffic=False)["frac"], "traffic": load_traffic(w, t["format_name"], dts, t["kernel"])})
        rec = {"workload": f"{w} ({'synthetic twin' if data == 'synthetic' else data})", "dtype": dts,
               "rows": int(A["m"]), "nnz": int(A["nnz"]), "algorithmic_bytes": int(named["algorithmic_bytes"]),
               "cache_resident": rn["cache_resident"],
               "named_kernel": named["kernel"], "named_format": named["format_name"], "named_ms": round(named["wall_ms"], 6),
               "named_kernel_ms": round(named["kernel_ms"], 6),
               "named_gflops": round(2.0 * A["nnz"] / (named["wall_ms"] * 1e-3) / 1e9, 2), "named_frac": rn["frac"], "named_traffic": rn["traffic"],
               "best_kernel": best["kernel"], "best_format": best["format_name"], "best_ms": round(best["wall_ms"], 6),
               "best_kernel_ms": round(best["kernel_ms"], 6),
               "best_gflops": round(2.0 * A["nnz"] / (best["wall_ms"] * 1e-3) / 1e9, 2), "best_frac": rb["frac"], "traffic": rb["traffic"],
               "frac_hbm_measured": rb["frac_hbm_measured"], "also": also,
               "check_max_err_over_abs_row": max(named["check"], best["check"]),
               "setup_s": round(t_gen + named["convert_s"] + best["convert_s"], 2)}
        if resident:
            B = named["algorithmic_bytes"]
            rec.update({"named_cold_ms": round(named["cold_ms"], 6), "named_cold_frac": round(B / (named["cold_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                        "best_cold_ms": round(best["cold_ms"], 6), "cold_frac": round(B / (best["cold_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                        "cold": "isolated launches, a 1 GiB buffer overwritten before each (bench_spmv.cpp:331-348 CLEAR_CACHES); median of 15"})
        if w in SYMMETRIC_CONFIGS:
            rec["symmetric_storage"] = symmetric_storage_leg(E, torch, A, w, dts, steps, warm)
        if not args.no_cpu_baseline:
            # the reference's CPU CSR kernel on the same matrix (whole matrix where it has <= 64 M non-zeros: config 1 IS this on cant)
            saved = args.cpu_baseline_seconds
            args.cpu_baseline_seconds = min(saved, 2.0)
            rec["cpu_baseline"] = cpu_baseline(args, w, dts, A, named["x_host"], best["yh"])
            args.cpu_baseline_seconds = saved
        out.append(rec)
        del A, named, best
    return out


def symmetric_storage_leg(E, torch, A, w, dts, steps, warm):
    """Row f4 (KEEP_SYMMETRY builds, csr_sym.cpp): the LOWER triangle of the twin as the stored triangle of a symmetric matrix
    (T + T^t - diag T: the twin mirrored, not the twin itself), multiplied by the symmetric-storage kernel WITHOUT expanding it, beside
    the same symmetric matrix expanded and run through the general SELL path. Fractions are on the EXPANDED matrix's algorithmic bytes."""
    import scipy.sparse as sp
    m = A["m"]
    M = sp.csr_matrix((A["values"], A["col_idx"], A["row_ptr"]), shape=(m, m))
    T = sp.tril(M).tocsr()
    T.sort_indices()
    Ex = (T + sp.tril(M, -1).T).tocsr()
    Ex.sort_indices()
    tri = dict(m=m, n=m, nnz=int(T.nnz), row_ptr=T.indptr.astype(np.int32), col_idx=T.indices.astype(np.int32), values=T.data.astype(np.float64))
    full = dict(m=m, n=m, nnz=int(Ex.nnz), row_ptr=Ex.indptr.astype(np.int32), col_idx=Ex.indices.astype(np.int32), values=Ex.data.astype(np.float64))
    B = algorithmic_bytes(m, m, int(Ex.nnz), 8 if dts == "f64" else 4)
    # the sampled-row check of time_handle multiplies the arrays it is given: hand it the expanded matrix, the handle the triangle
    np_dtype = np.float64 if dts == "f64" else np.float32
    x_host = np.random.default_rng(14).uniform(-1.0, 1.0, m).astype(np_dtype)
    S = E.Matrix(tri["row_ptr"], tri["col_idx"], tri["values"], m, m, "sell_c_sigma", np_dtype, symmetric_input=1, sell_window=1, placement=1)
    S.upload_x(x_host)
    xp, yp = S.x_device(), S.y_device()
    sp_ = torch.cuda.current_stream().cuda_stream
    t0 = time.time()
    while time.time() - t0 < max(warm, 0.05):
        S.time_device(xp, yp, 50, sp_)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k_ms = S.time_device(xp, yp, steps, sp_)
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) / steps * 1e3
    yh = S.download_y().astype(np.float64)
    err, _ = sampled_row_check(yh, full["row_ptr"], full["col_idx"], full["values"], x_host, np_dtype)
    tol = 1e-12 if dts == "f64" else 1e-5
    if not (err <= tol):
        raise SystemExit(f"bench sanity check failed for {S.format_name}: {err}")
    out = {"what": "lower triangle of the twin stored, y = (T + T^t - diag T) x without expanding it (csr_sym.cpp:191-267)",
           "format": S.format_name, "kernel": S.kernel_info()["name"], "stored_nnz": int(T.nnz), "expanded_nnz": int(Ex.nnz),
           "mem_footprint": S.mem_footprint, "ms": round(wall_ms, 6), "kernel_ms": round(k_ms, 6),
           "frac_on_expanded_bytes": round(B / (wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "check_max_err_over_abs_row": err,
           "traffic": load_traffic(w + ":sym1", S.format_name, dts, "sell_window_sym_kernel")}
    S.close()
    g = time_handle(E, torch, full, "sell_c_sigma", dts, {}, steps, 20, x_host=x_host, min_warm_seconds=warm)
    out["expanded"] = {"format": g["format_name"], "ms": round(g["wall_ms"], 6), "mem_footprint": g["mem_footprint"],
                       "frac": round(B / (g["wall_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                       "traffic": load_traffic(w + ":sym2", g["format_name"], dts, g["kernel"])}
    if out["traffic"] and out["expanded"]["traffic"]:
        out["traffic_ratio_to_expanded"] = round(out["traffic"] / out["expanded"]["traffic"], 4)
    return out


def strided_sample(rp, ci, va, lm, max_nnz, chunks=64):
    """An nnz-bounded sample of the matrix for the CPU baseline: `chunks` row ranges spread evenly over the whole matrix (each
    holding 1/chunks of the budget), concatenated into one CSR over the full column space."""
    lnnz = int(rp[lm])
    if lnnz <= max_nnz:
        return rp[:lm + 1], ci[:lnnz], va[:lnnz], np.arange(lm), f"whole matrix ({lnnz} nnz)"
    per = max_nnz // chunks
    rows = []
    for c in range(chunks):
        r0 = int(np.searchsorted(rp, int(lnnz * c / chunks), side="left"))
        r0 = min(r0, lm)
        r1 = int(np.searchsorted(rp, int(rp[r0]) + per, side="right")) - 1
        r1 = max(r0, min(r1, lm))
        if rows and r0 < rows[-1][1]:
            r0 = rows[-1][1]
        if r1 > r0:
            rows.append((r0, r1))
    srp = [np.zeros(1, np.int64)]
    acc = 0
    for r0, r1 in rows:
        seg = rp[r0:r1 + 1].astype(np.int64)
        srp.append(seg[1:] - seg[0] + acc)
        acc += int(seg[-1] - seg[0])
    srp = np.concatenate(srp).astype(np.int32)
    sci = np.concatenate([ci[rp[r0]:rp[r1]] for r0, r1 in rows])
    sva = np.concatenate([va[rp[r0]:rp[r1]] for r0, r1 in rows])
    row_ids = np.concatenate([np.arange(r0, r1) for r0, r1 in rows])
    return srp, sci, sva, row_ids, f"{len(rows)} row ranges spread evenly over the matrix ({len(row_ids)} rows, {acc} nnz)"


def cpu_baseline(args, workload, dts, A, x_host, yh):
    """The reference's CPU CSR backend (oracle/_ref, compiled in place from /root/reference by oracle/Makefile) timed on this
    host's cores on a bounded sample of the same workload; falls back to the oracle's port of the same loop."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc                                  # checker only: timed beside the GPU, never shipped
    # the GPU box shows every host CPU but one GPU's share is 16 cores (oversubscribing 256 threads is 10x slower)
    cores = int(os.environ.get("SPMV_CPU_THREADS", min(cpu_share(), 16)))
    rp, ci, va, lm, n = A["row_ptr"], A["col_idx"], A["values"], A["m"], A["n"]
    srp, sci, sva, row_ids, sample = strided_sample(rp, ci, va, lm, 64_000_000)
    rs, snnz = len(srp) - 1, int(srp[-1])
    kind, tb, ys = "port", None, None
    try:
        import refdrv
        flavour = "native" if refdrv.available("csr", "d", "native") else "v3"
        prec = "f" if dts == "f32" else "d"
        if refdrv.available("csr", prec, flavour):
            with QuietStdout():                                   # the reference prints to C stdout; ours carries ONE JSON line
                rb = refdrv.RefBackend("csr", prec, flavour, threads=cores)
                rb.csr_to_format(srp, sci, sva, rs, n)
                ys = rb.spmv(x_host).astype(np.float64)
                tb = rb.time_spmv(x_host, min_loops=5, min_runtime=args.cpu_baseline_seconds)
            kind = "reference"
    except Exception as e:                                        # e.g. an AVX-512 build on a CPU without it
        print(f"[bench] reference CPU backend unavailable ({e}); timing the oracle port", file=sys.stderr)
        tb = None
    if tb is None:
        tb = orc.time_csr_spmv(srp, sci, sva, x_host.astype(np.float64), cores, min_loops=5, min_runtime=args.cpu_baseline_seconds)
        ys = np.asarray(tb["y"], np.float64)
    # the baseline must have done the work: its y agrees with the GPU's on the sampled rows
    probe = np.unique(np.random.default_rng(2).integers(0, max(rs, 1), 2000))
    if len(probe):
        g = yh[row_ids[probe]]
        dmax = float(np.max(np.abs(ys[probe] - g) / np.maximum(np.abs(g), 1e-300)))
        assert dmax < 1e-9 or dts == "f32", f"CPU baseline result differs from the GPU result ({dmax})"
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            cpu_model = next((l.split(":", 1)[1].strip() for l in fh if l.startswith("model name")), "")
    except OSError:
        pass
    vb = (4 if dts == "f32" else 8) if kind == "reference" else 8
    touched = np.zeros(n, bool)
    touched[sci] = True
    x_read = int(touched.sum())
    # bytes of the sample: its matrix stream, its row pointers, the x entries it actually reads, the y entries it writes
    sbytes = snnz * (vb + 4) + (rs + 1) * 4 + (x_read + rs) * vb
    return {"value": round(2.0 * snnz / tb["median"] / 1e9, 3), "unit": "GFLOP/s", "cores": cores, "cpu_model": cpu_model,
            "kind": kind, "sample": f"{sample} of {workload}", "median_s": tb["median"], "loops": tb["loops"],
            "gbps": round(sbytes / tb["median"] / 1e9, 2), "sample_bytes": int(sbytes), "x_entries_read": x_read}


def main():
    args = parse()
    # Host threads: the GPU box shows all 256 cores but a job owns a cgroup CPU quota (16 cores per GPU); an OpenMP team
    # of 256 burns the quota in a few ms and the kernel then freezes the whole process — the thread feeding the GPU
    # included — until the next 100 ms period (csrc/host_threads.hpp has the measurement). torch.distributed.run on the
    # other hand pins OMP_NUM_THREADS=1. Either way: give every rank its share, before any OpenMP runtime starts.
    world = int(os.environ.get("WORLD_SIZE", "1"))
    share = max(1, min(cpu_share(), len(os.sched_getaffinity(0))) // int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    if "OMP_NUM_THREADS" not in os.environ or (world > 1 and os.environ["OMP_NUM_THREADS"] == "1"):
        os.environ["OMP_NUM_THREADS"] = str(min(share, 16) if world > 1 else share)
    os.environ.setdefault("KMP_BLOCKTIME", "0")
    import torch
    import spmv_host as H
    import spmv_mi355x as E

    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    if world > 1 or args.force_multi:
        import bench_multi
        result = bench_multi.run(args, sys.modules[__name__])
        if rank == 0:
            print(json.dumps(result), flush=True)
        return

    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    workload = args.workload
    fmt = args.format or DEFAULT_FORMAT.get(workload, "csr_vector")
    dts = args.dtype or DEFAULT_DTYPE.get(workload, "f64")
    opts = collect_opts(args, workload)

    if args.index_modes_off:
        os.environ["SPMV_MI355X_SELL_MODES_OFF"] = str(args.index_modes_off)
    t0 = time.time()
    A, data = load_workload(H, workload, args.scale, args.jitter, args.jitter_span)
    t_gen = time.time() - t0
    m, n, nnz = A["m"], A["n"], A["nnz"]
    small = nnz < 20_000_000
    # a short sample (the driver's --steps 20 is 25 ms of a kernel whose level moves by a few % over tenths of a second) is timed as
    # 5 windows of K steps, each in its own synchronize bracket; the line carries the MEDIAN window and the spread
    windows = 5 if args.steps < 200 else 1
    opts.setdefault("placement", args.placement)
    t = time_handle(E, torch, A, fmt, dts, opts, args.steps, args.warmup, min_warm_seconds=0.25 if small else 0.0,
                    idle_after_placement=args.idle_after_placement, windows=windows)
    ms_per_step = t["wall_ms"]                    # synchronize bracket around exactly K launches (the median window of `windows`)
    gflops = 2.0 * nnz / (ms_per_step * 1e-3) / 1e9
    B_alg = t["algorithmic_bytes"]
    wl = f"{workload} ({'synthetic twin' if data == 'synthetic' else data})" + ("" if args.scale == 1.0 else f" scale={args.scale}")
    if args.jitter:
        wl += f" jitter={args.jitter}x+-{args.jitter_span}"
    if args.index_modes_off:
        wl += f" index-modes-off={args.index_modes_off}"
    result = {
        "metric": f"GFLOP/s (2*nnz/t, {'fp64' if dts == 'f64' else 'fp32'} SpMV y=A*x); achieved HBM GB/s and % of peak in 'roofline'",
        "value": round(gflops, 3), "unit": "GFLOP/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 6), "windows": windows, "ms_per_step_min": min(t["wall_ms_windows"]), "ms_per_step_max": max(t["wall_ms_windows"]),
        "higher_is_better": True,
        "scaling": "strong",            # total work (one SpMV of the whole matrix) is fixed as N grows
        "vs_baseline": None, "dtype": dts, "data": "synthetic" if data == "synthetic" else data,
        "config": {"workload": wl, "format": t["format_name"], "rows": int(m), "cols": int(n), "nnz": int(nnz),
                   "parallelism": "single GPU", "stored_bytes_per_nnz": round(t["mem_footprint"] / max(nnz, 1), 3)},
        "hbm_gbps_algorithmic": round(B_alg / (ms_per_step * 1e-3) / 1e9, 2),
        "hbm_pct_of_peak": round(100.0 * B_alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 2),
        "roofline": roofline_record(workload, dts, t, with_traffic=args.scale == 1.0 and not args.jitter and not args.index_modes_off),
        "check_max_err_over_abs_row": t["check"],
        "setup_s": {"generate": round(t_gen, 2), "convert_upload": round(t["convert_s"], 2), "upload_x_place_vectors": round(t["place_s"], 2)},
        "placement": E.placement_info(torch.cuda.current_device()),
    }
    if workload == "nlpkkt240" and fmt == "sell_c_sigma" and not args.index_modes_off and not args.jitter and args.scale == 1.0 and not args.headline_only:
        # the other end of the headline: the same matrix with every index-free mode of the delta layout switched off (8/16-bit deltas per
        # lane and step everywhere) — what a matrix without the twin's translation invariance gets from the format
        os.environ["SPMV_MI355X_SELL_MODES_OFF"] = "7"
        tf = time_handle(E, torch, A, fmt, dts, opts, min(args.steps, 300), args.warmup, x_host=t["x_host"], windows=windows)
        del os.environ["SPMV_MI355X_SELL_MODES_OFF"]
        rf = roofline_record(workload, dts, tf, with_traffic=False)
        result["roofline"]["frac_floor"] = rf["frac"]
        result["roofline"]["floor"] = {"what": "index-free modes off (SPMV_MI355X_SELL_MODES_OFF=7)", "ms": rf["ms"], "kernel_ms": rf["kernel_ms"],
                                       "stored_bytes_per_nnz": round(tf["mem_footprint"] / max(nnz, 1), 3)}
        del tf
        # ... and in between: 5 % of the rows out of line (their off-diagonal columns moved by up to +-3): the slices they sit in keep
        # their lane offsets and carry one signed byte per step for the rows that do not fit (mode 5 of the layout)
        Aj, _ = load_workload(H, workload, args.scale, 0.05, 3)
        tj = time_handle(E, torch, Aj, fmt, dts, opts, min(args.steps, 300), args.warmup, windows=windows)
        rj = roofline_record(workload, dts, tj, with_traffic=False)
        result["roofline"]["jitter"] = {"what": "5 % of the rows perturbed by +-3 columns (bench.py --jitter 0.05)", "frac": rj["frac"], "ms": rj["ms"],
                                        "kernel_ms": rj["kernel_ms"], "stored_bytes_per_nnz": round(tj["mem_footprint"] / max(Aj["nnz"], 1), 3)}
        del Aj, tj
    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, workload, dts, A, t["x_host"], t["yh"])
    del A, t
    want_configs = args.configs == "on" or (args.configs == "auto" and workload == "nlpkkt240" and args.format is None
                                            and args.scale == 1.0 and not args.jitter and not args.index_modes_off)
    if want_configs and not args.headline_only:
        result["configs"] = run_small_configs(E, torch, H, args)
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
