#!/bin/bash
set -o pipefail
run() { echo "$*: $(timeout -k 10 300 python tools/run_one.py "$@" --iters 300 2>&1 | tail -1)"; }
for nt in 1 2; do
run --workload cant --format sell_c_sigma --opt nontemporal=$nt
run --workload cant --dtype f32 --format sell_c_sigma --opt nontemporal=$nt
run --workload cant --format csr_stream --opt nontemporal=$nt
run --workload pwtk --dtype f32 --format sell_c_sigma --opt nontemporal=$nt
run --workload pwtk --dtype f32 --format csr_stream --opt nontemporal=$nt
run --workload pwtk --format sell_c_sigma --opt nontemporal=$nt
run --workload pwtk --format csr_stream --opt nontemporal=$nt
run --workload scircuit --format csr_vector --opt nontemporal=$nt
run --workload scircuit --format csr_stream --opt nontemporal=$nt
run --workload cant --format csr_vector --opt nontemporal=$nt
done
