// Matrix_Format adapter: the ONE translation unit a maintainer adds to benchmark_code/BENCH/src/spmv_kernels/ to make
// the MI355X engine one more link-time backend of the reference harness (INTEGRATION.md has the Makefile_in rule).
// Compiled by g++ (no HIP headers): everything device-side sits behind the C ABI of include/spmv_mi355x.h, the same
// split the reference's own ROCm backends use (GPU_clean/csr_rocm_vector.cpp:215,239).
//
// Mirrors spmv_kernels/spmv_kernel_template.cpp:20-93 (a Matrix_Format subclass + csr_to_format +
// statistics_print_labels). The kernel family is chosen at compile time with -DSPMV_MI355X_FORMAT=<id> (one executable
// per format, like every other backend) or at run time with the environment variable SPMV_MI355X_FORMAT
// (csr_scalar | csr_vector | csr_merge | sell_c_sigma | coo | csr_stream); tunables via SPMV_MI355X_LANES_PER_ROW, _SELL_C,
// _SELL_SIGMA, _SELL_DELTA, _SELL_SPLIT, _SELL_WINDOW, _SELL_GROUP, _MERGE_ITEMS, _STREAM_MODE, _ROWS_PER_GROUP, _COL_BLOCKS, _XCD_REMAP
// (the fields of spmv_mi355x_opts). SPMV_MI355X_NGPUS=N (N > 1) cuts the matrix into N nnz-balanced row blocks over the node's
// GPUs behind the same Matrix_Format (spmv_mi355x_create_partitioned: RCCL allgather of x overlapped with the local columns);
// SPMV_MI355X_EXCHANGE = 1 forces RCCL, 2 peer copies. Errors end the process the way the reference's error() does
// (lib/debug.h:83-135).

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spmv_kernel.h"
#include "spmv_mi355x.h"

#define mi355x_error(...)                                                                  \
	do {                                                                               \
		fprintf(stderr, "%s:%s:%d: ERROR: ", __FILE__, __func__, __LINE__);        \
		fprintf(stderr, __VA_ARGS__);                                              \
		fprintf(stderr, "\n");                                                     \
		exit(EXIT_FAILURE);                                                        \
	} while (0)

static int
env_int(const char * name, int dflt)
{
	const char * s = getenv(name);
	return (s && *s) ? atoi(s) : dflt;
}

static int
chosen_format()
{
	const char * s = getenv("SPMV_MI355X_FORMAT");
	if (s && *s)
	{
		static const char * names[] = {"csr_scalar", "csr_vector", "csr_merge", "sell_c_sigma", "coo", "csr_stream"};
		for (int i = 0; i < SPMV_MI355X_NUM_FORMATS; i++)
			if (!strcmp(s, names[i]))
				return i;
		mi355x_error("SPMV_MI355X_FORMAT='%s' is not one of csr_scalar, csr_vector, csr_merge, sell_c_sigma, coo, csr_stream", s);
	}
	#ifdef SPMV_MI355X_FORMAT
		return SPMV_MI355X_FORMAT;
	#else
		return SPMV_MI355X_CSR_VECTOR;
	#endif
}

struct MI355XFormat : Matrix_Format
{
	spmv_mi355x_matrix * handle;
	spmv_mi355x_partitioned * parts;       // SPMV_MI355X_NGPUS > 1: the same matrix as row blocks on several GPUs

	MI355XFormat(INT_T * row_ptr, INT_T * col_ind, ValueTypeReference * values, long m, long n, long nnz, int symmetric_input)
		: Matrix_Format(m, n, nnz), handle(NULL), parts(NULL)
	{
		spmv_mi355x_opts o;
		memset(&o, 0, sizeof(o));
		o.struct_size = sizeof(o);
		o.device = env_int("SPMV_MI355X_DEVICE", -1);
		o.lanes_per_row = env_int("SPMV_MI355X_LANES_PER_ROW", 0);
		o.sell_c = env_int("SPMV_MI355X_SELL_C", 0);
		o.sell_sigma = env_int("SPMV_MI355X_SELL_SIGMA", 0);
		o.merge_items = env_int("SPMV_MI355X_MERGE_ITEMS", 0);
		o.stream_mode = env_int("SPMV_MI355X_STREAM_MODE", 0);
		o.rows_per_group = env_int("SPMV_MI355X_ROWS_PER_GROUP", 0);
		o.col_blocks = env_int("SPMV_MI355X_COL_BLOCKS", 0);
		o.sell_delta = env_int("SPMV_MI355X_SELL_DELTA", 0);
		o.sell_split = env_int("SPMV_MI355X_SELL_SPLIT", 0);
		o.xcd_remap = env_int("SPMV_MI355X_XCD_REMAP", 0);
		o.sell_window = env_int("SPMV_MI355X_SELL_WINDOW", 0);
		o.sell_group = env_int("SPMV_MI355X_SELL_GROUP", 0);
		o.symmetric_input = symmetric_input;
		const int precision = (sizeof(ValueType) == 8) ? SPMV_MI355X_F64 : SPMV_MI355X_F32;
		const int ngpus = env_int("SPMV_MI355X_NGPUS", 1);
		if (ngpus > 1)
		{
			if (symmetric_input)
				mi355x_error("SPMV_MI355X_NGPUS needs the expanded matrix (KEEP_SYMMETRY builds hand over one triangle)");
			o.device = -1;
			if (spmv_mi355x_create_partitioned(&parts, ngpus, NULL, env_int("SPMV_MI355X_EXCHANGE", 0), chosen_format(), precision, m, n, nnz,
					row_ptr, col_ind, values, &o))
				mi355x_error("%s", spmv_mi355x_last_error());
			mem_footprint = spmv_mi355x_partitioned_mem_footprint(parts);
			format_name = strdup(spmv_mi355x_partitioned_format_name(parts));
			return;
		}
		// deep copy happens inside create(): the driver frees its CSR right after this call (bench.cpp:605-629)
		if (spmv_mi355x_create(&handle, chosen_format(), precision, m, n, nnz, row_ptr, col_ind, values, &o))
			mi355x_error("%s", spmv_mi355x_last_error());
		mem_footprint = spmv_mi355x_mem_footprint(handle);
		format_name = strdup(spmv_mi355x_format_name(handle));
	}

	// x is uploaded when its host pointer is new, y is downloaded on the first call only — the convention of every GPU
	// backend in the reference (csr_rocm_vector.cpp:224-257). CG/BiCG-style callers set SPMV_MI355X_ALWAYS_COPY=1.
	void spmv(ValueType * x, ValueType * y)
	{
		if (parts ? spmv_mi355x_spmv_partitioned(parts, x, y) : spmv_mi355x_spmv(handle, x, y))
			mi355x_error("%s", spmv_mi355x_last_error());
	}

	void statistics_start() {}
	int statistics_print_data(__attribute__((unused)) char * buf, __attribute__((unused)) long buf_n) { return 0; }
};

struct Matrix_Format *
csr_to_format(INT_T * row_ptr, INT_T * col_ind, ValueTypeReference * values, long m, long n, long nnz, long symmetric, long symmetry_expanded)
{
	// Both conventions of the harness are accepted: expanded input (the default build, csr.cpp:221-226) and, in KEEP_SYMMETRY
	// builds, one stored triangle (symmetric = 1, symmetry_expanded = 0: what csr_sym.cpp:118-123 takes). Matrix_Format's
	// m/n/nnz/csr_mem_footprint stay those of the arrays the harness passed, as in the reference.
	MI355XFormat * mf = new MI355XFormat(row_ptr, col_ind, values, m, n, nnz, (symmetric && !symmetry_expanded) ? 1 : 0);
	if (env_int("SPMV_MI355X_ALWAYS_COPY", 0))
	{
		if (mf->parts)
			spmv_mi355x_partitioned_set_always_copy(mf->parts, 1);
		else
			spmv_mi355x_set_always_copy(mf->handle, 1);
	}
	return mf;
}

// For bench_cg.cpp / bench_bicg.cpp: the device-resident solvers of the C ABI need the handle behind the Matrix_Format
// (INTEGRATION.md §6). NULL when MF is not this backend's format cannot happen: one backend per executable.
extern "C" spmv_mi355x_matrix *
spmv_mi355x_handle_of(struct Matrix_Format * MF)
{
	return static_cast<MI355XFormat *>(MF)->handle;
}

int
statistics_print_labels(__attribute__((unused)) char * buf, __attribute__((unused)) long buf_n)
{
	return 0;
}
