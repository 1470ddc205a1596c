"""The drop-in boundary, guarded: the Matrix_Format adapter TU (spmv-research_amd/host/spmv_kernel_mi355x.cpp) compiles against
the REFERENCE's own spmv_kernels/spmv_kernel.h (benchmark_code/BENCH/src/spmv_kernels/spmv_kernel.h:8-29) in both precisions and
exports the two factory functions of the plug-in API, and our rendering of that header (host/spmv_kernel.h, used where
/root/reference does not exist) gives `struct Matrix_Format` the same size and member offsets.

CPU tier; skipped where the reference tree is absent (the GPU box). Nothing of the reference is copied: its header is only
named on the compiler's include path, outputs go to a scratch directory."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_KERNELS = "/root/reference/benchmark_code/BENCH/src/spmv_kernels"
REF_LIB = "/root/reference/lib"
ADAPTER = os.path.join(ROOT, "spmv-research_amd", "host", "spmv_kernel_mi355x.cpp")
OURS = os.path.join(ROOT, "spmv-research_amd", "host")

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF_KERNELS, "spmv_kernel.h")) or shutil.which("g++") is None,
                                reason="needs the reference tree and g++ (build container only)")

# the macros the reference build supplies (make.sh:166,191-192,212-216)
FLAGS = {"d": ["-DINT_T=int32_t", "-DValueType=double", "-DValueTypeReference=double", "-DDOUBLE=1"],
         "f": ["-DINT_T=int32_t", "-DValueType=float", "-DValueTypeReference=double", "-DDOUBLE=0"]}

LAYOUT_SRC = r"""
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include "spmv_kernel.h"
struct Probe : Matrix_Format {
	Probe() : Matrix_Format(7, 5, 11) {}
	void spmv(ValueType *, ValueType *) {}
	void statistics_start() {}
	int statistics_print_data(char *, long) { return 0; }
};
int main()
{
	Probe p;
	printf("sizeof=%zu format_name=%zu m=%zu n=%zu nnz=%zu mem_footprint=%zu csr_mem_footprint=%zu\n", sizeof(Matrix_Format),
			offsetof(Matrix_Format, format_name), offsetof(Matrix_Format, m), offsetof(Matrix_Format, n), offsetof(Matrix_Format, nnz),
			offsetof(Matrix_Format, mem_footprint), offsetof(Matrix_Format, csr_mem_footprint));
	printf("m=%ld n=%ld nnz=%ld csr_mem_footprint=%.1f\n", p.m, p.n, p.nnz, p.csr_mem_footprint);
	return 0;
}
"""


@pytest.mark.parametrize("prec", ["d", "f"])
def test_adapter_compiles_against_the_reference_header(tmp_path, prec):
    # as a maintainer would: the TU dropped into a directory that does NOT hold our rendering of the header (a quoted include looks
    # beside the including file first), the reference's spmv_kernels/ on the include path
    tu = tmp_path / "spmv_kernel_mi355x.cpp"
    shutil.copyfile(ADAPTER, tu)
    obj = tmp_path / f"adapter_{prec}.o"
    cmd = ["g++", "-std=gnu++17", "-O1", "-Wall", "-c"] + FLAGS[prec] + ["-I", REF_KERNELS, "-I", REF_LIB, "-I", os.path.join(ROOT, "include"),
                                                                         str(tu), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    # the dependency file proves WHICH spmv_kernel.h the adapter saw
    dep = subprocess.run(cmd[:-2] + ["-MM"], capture_output=True, text=True, cwd=tmp_path)
    assert os.path.join(REF_KERNELS, "spmv_kernel.h") in dep.stdout and os.path.join(OURS, "spmv_kernel.h") not in dep.stdout, dep.stdout
    syms = subprocess.run(["nm", "-C", str(obj)], capture_output=True, text=True).stdout
    vt = "double" if prec == "d" else "float"
    assert " T csr_to_format(int*, int*, double*, long, long, long, long, long)" in syms       # spmv_kernel.h:28 (values are fp64 in both builds)
    assert " T statistics_print_labels(char*, long)" in syms                                  # spmv_kernel.h:29
    assert f"MI355XFormat::spmv({vt}*, {vt}*)" in syms                                         # the virtual the driver's loop calls
    assert " U spmv_mi355x_create" in syms and " U spmv_mi355x_spmv" in syms                   # device side only through the C ABI


@pytest.mark.parametrize("prec", ["d", "f"])
def test_our_header_gives_matrix_format_the_reference_layout(tmp_path, prec):
    src = tmp_path / "layout.cpp"
    src.write_text(LAYOUT_SRC)
    outs = {}
    for name, inc in (("reference", ["-I", REF_KERNELS, "-I", REF_LIB]), ("ours", ["-I", OURS])):
        exe = tmp_path / f"layout_{name}_{prec}"
        r = subprocess.run(["g++", "-std=gnu++17", "-O0", "-Wno-invalid-offsetof"] + FLAGS[prec] + inc + [str(src), "-o", str(exe)],
                           capture_output=True, text=True, cwd=tmp_path)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[name] = subprocess.run([str(exe)], capture_output=True, text=True).stdout
    assert outs["ours"] == outs["reference"] and "sizeof=56" in outs["ours"], outs
    vb = 8 if prec == "d" else 4
    assert f"csr_mem_footprint={11 * (vb + 4) + 8 * 4:.1f}" in outs["ours"]                    # spmv_kernel.h:23
