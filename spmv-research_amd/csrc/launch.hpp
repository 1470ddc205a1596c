// Host-callable launchers of the HIP kernels (one translation unit per kernel family).
// All launchers enqueue on `stream` and return 0 / 1 (error text via spmv::set_error). `f32` selects float.
#pragma once

#include <vector>

#include "common.hpp"

namespace spmv {

struct LaunchCfg {
	XcdMap map;         // tile order over the 8 XCDs (built at conversion time for the kernel's tile geometry)
	int nt;             // nontemporal matrix streams
	int beta;           // 0: y = A x, 1: y += A x
	int unit;           // merge path: every stored value equals unit_value (pattern matrices: 1.0) and the value array is not kept
	double unit_value;
	int kahan;          // csr_scalar: Kahan-compensated row sums (csr.cpp:353-373)
};

// tile geometry of each kernel family (units per workgroup), needed to build the XCD map
inline long csr_scalar_rows_per_tile() { return 256; }
inline long csr_vector_rows_per_tile(int lanes_per_row, int rows_per_group = 1) { return 256L / lanes_per_row * rows_per_group; }
inline long csr_stream_rows_per_tile(int rows_per_wave) { return 4L * rows_per_wave; }
inline long sell_slices_per_tile() { return 4; }
inline long coo_waves_per_tile() { return 4; }

// ---- CSR (kernels_csr.hip)
int launch_csr_scalar(bool f32, const int * row_ptr, const int * col, const void * val, const void * x, void * y,
		int m, const LaunchCfg & cfg, hipStream_t stream, long * grid_out);
int launch_csr_vector(bool f32, int lanes_per_row, int rows_per_group, const int * row_ptr, const int * col, const void * val,
		const void * x, void * y, int m, const LaunchCfg & cfg, hipStream_t stream, long * grid_out);

// ---- CSR-Stream: one wavefront per block of R consecutive rows (kernels_csr_stream.hip)
int csr_stream_cap();                                                  // non-zeros a row block may hold on the LDS path
int launch_csr_stream(bool f32, int rows_per_wave, const int * row_ptr, const int * col, const void * val, const void * x, void * y,
		int m, const LaunchCfg & cfg, hipStream_t stream, long * grid_out);

int csr_stream_t_cap(int rows_per_wave);                               // same, transposed-consumption variant
int launch_csr_stream_t(bool f32, int rows_per_wave, const int * row_ptr, const int * col, const void * val, const void * x, void * y,
		int m, const LaunchCfg & cfg, hipStream_t stream, long * grid_out);

int csr_stream_d_cap(int rows_per_wave);
long csr_stream_d_rows_per_tile(int rows_per_wave);                    // rows per workgroup (workgroup size varies with R)                               // same, LDS-DMA variant
int launch_csr_stream_d(bool f32, int rows_per_wave, const int * row_ptr, const int * col, const void * val, const void * x, void * y,
		int m, const LaunchCfg & cfg, hipStream_t stream, long * grid_out);
constexpr int STREAM_SLACK = 512;                                      // entries the LDS-DMA copy may read past a row block

// CSR with the block's x window in LDS (kernels_csr_window.hip)
int csr_window_lds_budget();                                           // bytes of LDS a block's window may take
int launch_csr_window(bool f32, int lanes_per_row, const int * row_ptr, const void * col, int col16, const void * val, const void * x, void * y,
		const int * blk_row, const int * blk_lo, const int * blk_w, int lds_bytes, const LaunchCfg & cfg, hipStream_t stream,
		long * grid_out);

// ---- merge-path CSR (kernels_merge.hip)
int merge_tile_items(bool f32, int items_per_thread);                 // merge items (rows + nnz) per workgroup
int launch_merge_search(const int * row_ptr, int m, int nnz, int tile_items, int num_tiles, int * coords /* [2*(num_tiles+1)] */,
		hipStream_t stream);
int launch_merge(bool f32, int items_per_thread, const int * row_ptr, const int * col, const void * val, const void * x, void * y,
		int m, int nnz, int num_tiles, const int * coords, int * carry_row, void * carry_val,
		const LaunchCfg & cfg, hipStream_t stream, long * grid_out);

// ---- SELL-C-sigma (kernels_sell.hip)
int launch_sell(bool f32, int C, const int64_t * slice_ptr, const int * col, const void * val, const int * row_of_sorted,
		const void * x, void * y, int m, int num_slices, const LaunchCfg & cfg, hipStream_t stream, long * grid_out);

// SELL-64-sigma with delta-compressed column indices (desc: 2 int64 per slice + terminator, idx: byte stream)
int launch_sell_delta(bool f32, int waves_per_slice, const int64_t * desc, const unsigned char * idx, const void * val, const int * row_of_sorted,
		const void * x, void * y, int m, int num_slices, const LaunchCfg & cfg, hipStream_t stream, long * grid_out);

// SELL-64 with the slice group's x window in LDS and 16-bit window-relative indices (kernels_sell_window.hip)
int sell_window_lds_budget();
int launch_sell_window(bool f32, int waves_per_slice, int slices_per_group, const int * grp, const int64_t * sdesc, const unsigned short * idx,
		const void * val, const int * row_of_sorted, const void * x, void * y, int m, int lds_window_bytes, const LaunchCfg & cfg,
		hipStream_t stream, long * grid_out);

// the same layout holding ONE TRIANGLE of a symmetric matrix: y window in LDS beside the x window, mirrored entries as LDS atomics
int launch_sell_window_sym(bool f32, int waves_per_slice, int slices_per_group, const int * grp, const int64_t * sdesc, const unsigned short * idx,
		const void * val, const int * row_of_sorted, const void * x, void * y, int m, int lds_window_bytes, const LaunchCfg & cfg,
		hipStream_t stream, long * grid_out);

// Sensitivity experiments on the delta layout's index-free modes: SPMV_MI355X_SELL_MODES_OFF, bit 0 = no affine slices (mode 0),
// bit 1 = no per-slice lane offsets (mode 3), bit 2 = no lane offsets with exceptions (mode 5); such slices then store 8- / 16-bit
// deltas per lane. Read at every create().
inline int
sell_modes_off()
{
	const char * e = getenv("SPMV_MI355X_SELL_MODES_OFF");
	return e ? atoi(e) & 7 : 0;
}

// Where value (step k, lane r) of a 64-row slice of the delta layout lies behind the slice's first element: steps in pairs, a lane's steps
// 2p and 2p+1 side by side (kernels_sell.hip: sell_group_values); the last step of an odd width stands alone, one element per lane.
__host__ __device__ inline long
sell_pair_pos(long k, long width, long r)
{
	return (k | 1) < width ? (k / 2) * 128 + r * 2 + (k & 1) : (k / 2) * 128 + r;
}

// ... and of the LDS-window layout, whose slices are padded to whole groups of 4 steps: 16 bytes per lane and load — fp64 as above,
// fp32 a lane's 4 steps of a group side by side (kernels_sell_window.hip: sellw_values)
__host__ __device__ inline long
sellw_val_pos(long k, long r, bool f32)
{
	return f32 ? (k / 4) * 256 + r * 4 + (k & 3) : (k / 2) * 128 + r * 2 + (k & 1);
}

// CSR -> SELL-64-sigma-delta on the GPU (convert_sell.hip); outputs are device arrays owned by the caller
int sell_delta_convert_device(bool f32, long m, long n_cols, long nnz, long sigma, const int * rp_host, const int * ci_host,
		const double * va_host, int ** d_row_of_sorted_out, int64_t ** d_desc_out, unsigned char ** d_idx_out, void ** d_val_out,
		std::vector<int64_t> & val_ptr_host, long mode_counts[4], int64_t * nnz_ext_out, int64_t * idx_bytes_out);

// the same on a CSR already resident in device memory (rp, ci, va are device pointers; va fp64)
int sell_delta_convert_resident(bool f32, long m, long n_cols, long nnz, long sigma, const int * rp, const int * ci,
		const double * va, int ** d_row_of_sorted_out, int64_t ** d_desc_out, unsigned char ** d_idx_out, void ** d_val_out,
		std::vector<int64_t> & val_ptr_host, long mode_counts[4], int64_t * nnz_ext_out, int64_t * idx_bytes_out);
// the entry arrays of the column-blocked layout on the GPU (convert_coo.hip)
int blocked_entries_convert_device(bool f32, bool uniform, long m, long nnz, const int * rp_host, const int * ci_host, const double * va_host, long NR, int WGS, long CH,
		const std::vector<int> & range_row, const std::vector<int> & range_long, const std::vector<int> & long_row, const std::vector<int> & chunk_ptr,
		const std::vector<int> & chunk_row, const std::vector<int> & wg_rows, long SPAN, long BATCH, int slot_bits, int SPARE, long ghost_batches,
		std::vector<int> & batch_ptr, unsigned ** d_ent_out, void ** d_val_out, int ** d_batch_base_out);
// CSR -> the plain column-major SELL-C-sigma layout on the GPU (convert_sell.hip)
int sell_plain_convert_device(bool f32, long m, long nnz, int C, int TPR, long sigma, const int * rp_host, const int * ci_host, const double * va_host,
		int64_t ** d_slice_ptr_out, int ** d_col_out, void ** d_val_out, int ** d_row_of_sorted_out, std::vector<int64_t> & slice_ptr_host);
// CSR -> the LDS-window layout on the GPU (convert_sell.hip); 0 = built, 1 = error, 2 = a group's window is too wide
int sell_window_convert_device(bool f32, long m, long nnz, int NS, bool sym, long lds_budget_bytes, const int * rp_host, const int * ci_host,
		const double * va_host, int ** d_grp_out, int ** d_row_of_sorted_out, int64_t ** d_desc_out, unsigned short ** d_idx_out, void ** d_val_out,
		std::vector<int64_t> & desc_host, int * max_w_out);

// ---- COO (kernels_coo.hip)
int coo_wave_items(int items_per_lane);                                // entries per wavefront
int launch_coo(bool f32, int items_per_lane, const int * rowind, const int * col, const void * val, const void * x, void * y,
		int m, long nnz, int num_waves, int * carry_row, void * carry_val,
		const LaunchCfg & cfg, hipStream_t stream, long * grid_out);

// column-blocked COO for graph matrices: y of a workgroup's rows in LDS, entries ordered by column block, one dword per entry
int coo_blocked_rows_cap(bool f32);                                    // rows of y one workgroup can keep in LDS
int coo_blocked_wgs_per_range();                                       // workgroups a row range is dealt to (32 = CUs per XCD)
int coo_blocked_chunk_rows();                                          // rows per chunk of that round-robin deal (16)
int coo_blocked_batch_entries(bool unit);                              // entries of one batch of a workgroup (8 or 4 per lane)
int coo_blocked_spare_slots();                                         // LDS slots behind a workgroup's rows that padding entries add into
int coo_blocked_slot_bits();                                           // low bits of an entry that hold the LDS slot (the rest: column - base)
int coo_blocked_max_long_rows();                                       // rows per range that may be split over its workgroups
int launch_coo_blocked(bool f32, const int * wg_rows, const int * range_row, const int * chunk_ptr, const int * chunk_row, const int * batch_ptr, const int * batch_base,
		const int * range_long,
		const int * long_row, int num_long, const unsigned * ent, const void * val, const void * x, void * y, void * carry, int num_ranges,
		int chunk_rows, int lds_bytes, const LaunchCfg & cfg, hipStream_t stream, long * grid_out);

// ---- small utility kernels (kernels_csr.hip)
int launch_expand_rows(const int * row_ptr, int m, int * rowind, hipStream_t stream);   // CSR -> COO row indices (mkl_coo.cpp:79-90)

}  // namespace spmv
