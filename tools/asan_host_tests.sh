#!/bin/bash
# CPU-only: the host library (reader, in-process decompression, COO->CSR, partitioner, generators) under
# AddressSanitizer + UBSan, driven by its own test module. GPU sanitizers are not available on the pool.
set -e
cd "$(dirname "$0")/.."
S=spmv-research_amd
mkdir -p /tmp/asan_lib
g++ -O1 -g -std=c++17 -fPIC -fopenmp -march=x86-64-v3 -fsanitize=address,undefined -fno-omit-frame-pointer -shared \
	$S/host/file_load.cpp $S/host/matrix_market.cpp $S/host/csr_convert.cpp $S/host/partition.cpp $S/host/graph_partition.cpp $S/host/synthetic.cpp \
	$S/host/host_api.cpp -o /tmp/asan_lib/libspmv_host.so -lz -ldl
cp $S/lib/libspmv_host.so /tmp/asan_lib/libspmv_host.so.orig
trap 'cp /tmp/asan_lib/libspmv_host.so.orig '$S'/lib/libspmv_host.so' EXIT
cp /tmp/asan_lib/libspmv_host.so $S/lib/libspmv_host.so
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
	python -m pytest tests/test_host_golden.py tests/test_graph_partition.py tests/test_reader_fuzz.py \
		tests/test_bench_helpers.py::test_filtered_kkt_rows_equal_the_filtered_block -x -q
