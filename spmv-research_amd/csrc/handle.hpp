// The engine's handle (= one `struct Matrix_Format` instance of the reference, spmv_kernel.h:8-25) and the helpers shared by
// the per-format builders (build_*.hip) and the C ABI (spmv_mi355x.hip). Host code only.
#pragma once

#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <math.h>
#include <algorithm>
#include <numeric>
#include <vector>
#include <omp.h>

#include "../../include/spmv_mi355x.h"
#include "launch.hpp"

struct spmv_mi355x_matrix {
	int format = 0, precision = 0;
	long m = 0, n = 0, nnz = 0;            // local rows, columns, local non-zeros
	int device = 0;
	bool f32 = false;
	size_t vbytes = 8;
	spmv::LaunchCfg cfg{};
	int remap = 1;

	// CSR family
	int * d_row_ptr = nullptr;
	int * d_col = nullptr;
	void * d_val = nullptr;
	int lanes_per_row = 0;
	int rows_per_group = 1;                // CSR_VECTOR: rows a lane group keeps in flight (1, 2, 4)
	int * d_win_row = nullptr;             // CSR_STREAM mode 4: row block boundaries, window start, window length (0 = no LDS window)
	int * d_win_lo = nullptr;
	int * d_win_w = nullptr;
	unsigned short * d_col16 = nullptr;    // mode 4 with every window <= 65 536 columns: indices relative to the block's window
	int win_blocks = 0, win_lds_bytes = 0;
	int stream_mode = 0;                   // CSR_STREAM: 1 = products in LDS (row-major gather), 2 = (val,col) in LDS, lane-per-row walk
	// merge
	int merge_ipt = 0, merge_tile = 0, merge_num_tiles = 0;
	int * d_coords = nullptr;
	int * d_carry_row = nullptr;
	void * d_carry_val = nullptr;
	// column-blocked COO (opts.col_blocks; kernels_coo.hip): 8*P row ranges, 32 workgroups per range, entries sorted by column in full batches
	int * d_coob_wg_rows = nullptr;        // [ranges*32] rows of y a workgroup keeps in LDS
	int * d_coob_range_row = nullptr;      // [ranges+1]
	int * d_coob_chunk_ptr = nullptr;      // [ranges*32+1] first chunk of every workgroup
	int * d_coob_chunk_row = nullptr;      // [chunks] first row of a chunk of 16 rows
	int * d_coob_batch_ptr = nullptr;      // [ranges*32+1] first batch of every workgroup
	int * d_coob_batch_base = nullptr;     // [batches] base column of a batch
	unsigned * d_coob_ent = nullptr;       // [batches * K * 1024] (column - base) << 15 | LDS slot of the row
	int * d_coob_range_long = nullptr;     // [ranges+1] prefix of the split (hub) rows per range
	int * d_coob_long_row = nullptr;       // [num_long] their global row numbers
	void * d_coob_carry = nullptr;         // [num_long][32] partial sums of the split rows
	int coob_ranges = 0, coob_chunk_rows = 0, coob_lds = 0, coob_num_long = 0;
	long coob_batches = 0, coob_chunks = 0;
	// SELL
	int sell_c = 0;
	long sell_sigma = 0, sell_slices = 0, sell_nnz_ext = 0;
	int64_t * d_slice_ptr = nullptr;
	int * d_row_of_sorted = nullptr;
	bool sell_delta = false;               // delta-compressed column indices (C = 64 only)
	bool convert_on_device = true;         // build the delta layout on the GPU (convert_sell.hip) or on the host
	int sell_split = 1;                    // waves sharing one slice (delta format): 1, 2 or 4
	int64_t * d_sell_desc = nullptr;
	unsigned char * d_sell_idx = nullptr;
	long sell_idx_bytes = 0;
	bool sell_window = false;              // x window of a slice group in LDS, 16-bit window-relative indices (kernels_sell_window.hip)
	bool sell_sym = false;                 // the window layout holds ONE TRIANGLE of a symmetric matrix (sell_window_sym_kernel)
	int * d_sellw_grp = nullptr;           // [groups][4]: window start, width, first slice, slices
	int sellw_groups = 0, sellw_ns = 0, sellw_lds = 0;
	long sell_mode_slices[4] = {0, 0, 0, 0};  // slices stored with 8-bit / 16-bit / 32-bit indices / none (affine)
	// COO
	int coo_k = 0, coo_num_waves = 0;
	int * d_rowind = nullptr;

	// host-buffer path
	void * d_x = nullptr;
	void * d_y = nullptr;
	int placement_level = 0;               // opts.placement: 0 / 2 = off, 1 = the device's vector pools, 3 = + search over the matrix arrays
	long placement_budget_gib = 0;         // transient ballast the one walk of a device may hold (0 = 160)
	const void * cached_x_host = nullptr;
	bool y_downloaded = false;
	bool always_copy = false;
	hipStream_t stream = nullptr;

	double mem_footprint = 0, csr_mem_footprint = 0;
	char format_name[96] = "";
	char kernel_name[64] = "";
	int kernel_block = 256;                // threads per workgroup of the dominant kernel
	long last_grid = 0;
};

namespace spmv {

// ---- device memory helpers (handle.hip)
int dev_alloc_bytes(void ** p, size_t bytes);
int build_sell_delta_resident(spmv_mi355x_matrix * A, const spmv_mi355x_opts & o, const int * d_rp, const int * d_ci, const double * d_va);   // build_sell.hip
void init_handle(spmv_mi355x_matrix * A, int format, int precision, int device, const spmv_mi355x_opts & o, long m, long n, long nnz);   // spmv_mi355x.hip
int ensure_x(spmv_mi355x_matrix * A);                                    // spmv_mi355x.hip: stream + the handle's own (zeroed) x
int tune_placement(spmv_mi355x_matrix * A);                              // placement.hip: allocates the handle's y (and re-homes its x) in the device's vector pools
int place_vector(spmv_mi355x_matrix * A, void ** out, size_t bytes, bool is_output);   // placement.hip: a zero-filled vector A's SpMV writes / reads
int vector_free(void * p);                                               // placement.hip: what place_vector returned (pool slice or plain allocation)
template <typename T>
inline int
dev_alloc(T ** p, size_t count)
{
	return dev_alloc_bytes((void **) p, (count ? count : 1) * sizeof(T));
}
void free_all(spmv_mi355x_matrix * A);
// narrow fp64 reference values to the handle's precision (csr.cpp:72 `a[i] = values[i]`) and upload, with STREAM_SLACK spare entries
int upload_values(spmv_mi355x_matrix * A, const double * v, size_t count, void ** d_out);
int upload_ints(const int * src, size_t count, int ** d_out);
int upload_bytes(const void * src, size_t bytes, size_t slack_bytes, void ** d_out);
int pick_lanes_per_row(double mean);
int resolve_remap(int requested, long ntiles);
// every stored value (narrowed to the handle's precision) equals *v0_out: Matrix-Market `pattern` matrices carry the dummy 1.0
bool values_uniform(const spmv_mi355x_matrix * A, const double * va, long nnz, double * v0_out);

// ---- input stage of create() (build_input.hip): what csr_to_format() receives -> the local CSR a format is built from
struct LocalCsr {
	std::vector<int> e_rp, e_ci;          // symmetric expansion (opts.symmetric_input)
	std::vector<double> e_va;
	std::vector<int> l_rp, l_ci;          // row block / column filter copy
	std::vector<double> l_va;
	const int * rp = nullptr;
	const int * ci = nullptr;
	const double * va = nullptr;
	long m = 0, nnz = 0;
};
int prepare_local_csr(const spmv_mi355x_opts & o, long m, long n, long nnz, const int32_t * row_ptr, const int32_t * col_idx,
		const double * values, LocalCsr & out);

// ---- per-format builders (= the reference's csr_to_format constructors): fill the handle from the local CSR
int build_csr_family(spmv_mi355x_matrix * A, const spmv_mi355x_opts & o, const int * rp, const int * ci, const double * va);   // build_csr.hip
int build_sell_family(spmv_mi355x_matrix * A, const spmv_mi355x_opts & o, const int * rp, const int * ci, const double * va);  // build_sell.hip
int build_sell_symmetric(spmv_mi355x_matrix * A, const spmv_mi355x_opts & o, const int * rp, const int * ci, const double * va);   // build_sell.hip: 0 built, 1 error, 2 not applicable
int build_coo_family(spmv_mi355x_matrix * A, const spmv_mi355x_opts & o, const int * rp, const int * ci, const double * va);   // build_coo.hip
// the column-blocked layout shared by COO (ranges = equal shares of the non-zeros) and merge path (equal shares of rows + non-zeros)
int build_blocked_layout(spmv_mi355x_matrix * A, const int * rp, const int * ci, const double * va, int col_blocks, bool merge_balance);

}  // namespace spmv
