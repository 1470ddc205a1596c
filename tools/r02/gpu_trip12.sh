#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
run() { echo "$*: $(timeout -k 10 300 python tools/run_one.py "$@" --iters 300 2>&1 | tail -1)"; }
run --workload cant --format sell_c_sigma
run --workload cant --format sell_c_sigma --opt sell_window=1 --opt sell_split=8 --opt sell_group=2
run --workload cant --format sell_c_sigma --opt sell_window=1 --opt sell_split=8 --opt sell_group=1
run --workload cant --format sell_c_sigma --opt sell_window=1 --opt sell_split=4 --opt sell_group=4 --opt nontemporal=1
run --workload cant --format sell_c_sigma --opt sell_window=1 --opt sell_split=4 --opt sell_group=4 --opt xcd_remap=2
run --workload cant --format sell_c_sigma --opt sell_window=1 --opt sell_split=4 --opt sell_group=4 --opt xcd_remap=3
run --workload cant --format sell_c_sigma
run --workload scircuit --format sell_c_sigma --opt sell_window=1
run --workload scircuit --format csr_vector
run --workload scircuit --format csr_stream
