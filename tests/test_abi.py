"""The C-ABI shared objects load and export every symbol the headers in include/ declare (no compute calls: there
is no GPU in the CPU test tier), and the product fails loudly — never falls back — without a device."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, has_gpu


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spmv_(?:mi355x|host)_[a-z0-9_]+)\s*\(", text)))


@pytest.mark.parametrize("header,module", [("spmv_mi355x.h", "spmv_mi355x"), ("spmv_host.h", "spmv_host")])
def test_every_declared_symbol_is_exported(header, module):
    mod = __import__(module)
    lib = mod.lib()
    assert isinstance(lib, ctypes.CDLL)
    names = declared_functions(header)
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/{header} but not exported by {mod.LIB_PATH}"
    assert sorted(mod.SYMBOLS) == names, "python binding symbol list out of date with the header"


def test_opts_struct_layout_matches_header(tmp_path):
    """sizeof and every field offset of the ctypes mirrors against what a C compiler makes of include/spmv_mi355x.h"""
    import subprocess
    import spmv_mi355x as E
    lines = []
    for cname, cls in (("spmv_mi355x_opts", E.Opts), ("spmv_mi355x_solver_info", E.SolverInfo), ("spmv_mi355x_dist_ops", E.DistOps)):
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "spmv_mi355x.h"\nint main(void) {\n' + "\n".join(lines) + "\nreturn 0; }\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for cname, cls in (("spmv_mi355x_opts", E.Opts), ("spmv_mi355x_solver_info", E.SolverInfo), ("spmv_mi355x_dist_ops", E.DistOps)):
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


@pytest.mark.skipif(has_gpu(), reason="CPU tier only: checks the no-device failure mode")
def test_no_cpu_fallback_without_a_device():
    import spmv_mi355x as E
    assert E.device_count() == 0
    rp = np.array([0, 1], np.int32)
    with pytest.raises(E.SpmvError, match="no HIP device|no CPU fallback"):
        E.Matrix(rp, np.array([0], np.int32), np.array([1.0]), 1, 1, "csr_vector")


def test_argument_errors_of_the_entry_points_that_need_no_device():
    """Entry points added in round 3 reject bad arguments with a message and without touching a device: placement_info /
    placement_release on a device number out of range, stored_array / place_arrays / output_alloc / input_alloc without a handle."""
    import spmv_mi355x as E
    lib = E.lib()
    st, cand, npool, gib, us = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_long(), (ctypes.c_double * 4)()
    for dev in (-1, 64, 1000):
        assert lib.spmv_mi355x_placement_info(ctypes.c_int(dev), ctypes.byref(st), ctypes.byref(cand), ctypes.byref(gib), ctypes.byref(npool), us) == 1
        assert b"placement_info" in lib.spmv_mi355x_last_error()
    info = E.placement_info(0)                           # no walk was made in this process: a query, not an action
    assert info == dict(state="no walk", candidates=0, walked_gib=0, pools=0, us_per_pool=[])
    assert lib.spmv_mi355x_placement_release(ctypes.c_int(-1)) == 0          # nothing to release: fine
    out, nb = ctypes.c_void_p(), ctypes.c_size_t()
    assert lib.spmv_mi355x_stored_array(None, b"val", ctypes.byref(out), ctypes.byref(nb)) == 1
    assert b"stored_array" in lib.spmv_mi355x_last_error() and out.value is None
    assert lib.spmv_mi355x_place_arrays(None, None, None) == 1 and b"place_arrays" in lib.spmv_mi355x_last_error()
    assert lib.spmv_mi355x_output_alloc(None, ctypes.c_size_t(64), ctypes.byref(out)) == 1 and b"output_alloc" in lib.spmv_mi355x_last_error()
    assert lib.spmv_mi355x_input_alloc(None, ctypes.c_size_t(64), ctypes.byref(out)) == 1 and b"input_alloc" in lib.spmv_mi355x_last_error()
    assert lib.spmv_mi355x_output_free(None) == 0                            # free(NULL)


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under spmv-research_amd/ may include, link or import it."""
    pkg = os.path.join(ROOT, "spmv-research_amd")
    for d, _, files in os.walk(pkg):
        if os.sep + "build" in d or os.sep + "lib" in d or os.sep + "bin" in d or "__pycache__" in d:
            continue
        for f in files:
            if f.endswith((".so", ".o")):
                continue
            text = open(os.path.join(d, f), errors="ignore").read()
            assert "oracle" not in text.lower() or f == "spmv_kernel_mi355x.cpp" and False, f"{os.path.join(d, f)} mentions the oracle"
    import subprocess
    for so in ("libspmv_mi355x.so", "libspmv_host.so"):
        out = subprocess.run(["ldd", os.path.join(pkg, "lib", so)], capture_output=True, text=True).stdout
        assert "liboracle" not in out and "libref_" not in out


def test_driver_prints_the_reference_csv_header():
    import subprocess
    exe = os.path.join(ROOT, "spmv-research_amd", "bin", "spmv_mi355x_bench")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0
    # column list of bench_spmv.cpp:417-445 + check_accuracy_labels :94-105
    want = ("matrix_name,num_threads,csr_m,csr_n,csr_nnz,symmetry,time,time_iter_min,time_iter_median,time_iter_max,"
            "gflops,csr_mem_footprint,W_avg,J_estimated,format_name,m,n,nnz,mem_footprint,mem_ratio,num_loops,"
            "spmv_mae,spmv_max_ae,spmv_mse,spmv_mape,spmv_smape,spmv_lnQ_error,spmv_mlare,spmv_gmare")
    assert r.stderr.strip() == want


def test_driver_prints_the_reference_artificial_matrix_csv_header():
    """USE_ARTIFICIAL_MATRICES=1 (bench.cpp:497, config.sh conf_vars): no argument -> the synthetic-dataset label line of
    bench_spmv.cpp:493-522."""
    import subprocess
    exe = os.path.join(ROOT, "spmv-research_amd", "bin", "spmv_mi355x_bench")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60, env=dict(os.environ, USE_ARTIFICIAL_MATRICES="1"))
    assert r.returncode == 0
    want = ("matrix_name,distribution,placement,seed,nr_rows,nr_cols,nr_nzeros,density,mem_footprint,mem_range,avg_nnz_per_row,"
            "std_nnz_per_row,avg_bw,std_bw,avg_bw_scaled,std_bw_scaled,avg_sc,std_sc,avg_sc_scaled,std_sc_scaled,skew,"
            "avg_num_neighbours,cross_row_similarity,format_name,time,gflops,W_avg,J_estimated")
    assert r.stderr.strip() == want
    # too few generator parameters: usage, not a crash (the reference reads argv unchecked)
    r = subprocess.run([exe, "100", "100"], capture_output=True, text=True, timeout=60, env=dict(os.environ, USE_ARTIFICIAL_MATRICES="1"))
    assert r.returncode == 1 and "avg_nnz_per_row" in r.stderr


def test_driver_prints_the_reference_solver_csv_header():
    import subprocess
    exe = os.path.join(ROOT, "spmv-research_amd", "bin", "spmv_mi355x_bench")
    want = ("matrix_name,num_threads,csr_m,csr_n,csr_nnz,time,error,num_iterations,csr_mem_footprint,W_avg,J_estimated,"
            "format_name,m,n,nnz,mem_footprint,mem_ratio")                  # bench_cg.cpp:423-440 = bench_bicg.cpp:568-585
    for flag in ("--cg", "--bicgstab"):
        r = subprocess.run([exe, flag], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0
        assert r.stderr.strip() == want
