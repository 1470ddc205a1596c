// csr_to_format() for the CSR family: scalar, vector, stream (+ the x-window form) and merge path. The device arrays are the
// reference's (csr.cpp:50-167: row_ptr, ja, a); what differs per kernel is the tile geometry the XCD map is built for and the
// small side tables (row-block windows, merge-path tile coordinates).
#include "handle.hpp"

namespace spmv {

static int
build_csr_stream(spmv_mi355x_matrix * A, const spmv_mi355x_opts & o, const int * rp, const int * ci, const double * va)
{
	const long lm = A->m, lnnz = A->nnz;
	const char * pf = A->f32 ? "f" : "d";
	int rc = 0;
	int R = o.lanes_per_row;
	const double mean = lm > 0 ? (double) lnnz / lm : 0;
	int mode = o.stream_mode;
	// x window in LDS (kernels_csr_window.hip): nnz-balanced row blocks, a multiple of the 256 CUs. Forced by
	// stream_mode 4; in auto mode adopted when (nearly) every block's window fits the LDS budget and rows are long
	// enough to amortise the per-row butterfly (measured: pwtk twin fp32 22.1 -> 18.6 us; short-row / scattered
	// matrices keep the other modes). With 16-bit window-relative indices it also wins in fp64 (cant twin 11.2 -> 9.9 us,
	// pwtk twin 24.7 -> 23.7 us); fp64 with 32-bit indices ties with csr_vector and is not adopted automatically.
	auto try_window = [&](bool force) -> int {
		int G = R ? R : std::max(8, pick_lanes_per_row(mean));
		if (G != 8 && G != 16 && G != 32 && G != 64)
		{
			if (!force)
				return 0;
			set_error("csr_stream mode 4: lanes_per_row must be 8, 16, 32 or 64 (got %d)", G);
			return -1;
		}
		const int NG = 1024 / G;
		long nb = 256L * (o.merge_items > 0 ? o.merge_items : std::max(1L, std::min(8L, lnnz / (256L * 24576L))));
		nb = std::max(1L, std::min(nb, (lm + 2 * NG - 1) / (2 * NG)));
		std::vector<int> b_row, b_lo, b_w;
		std::vector<long> b_nnz;
		const long budget = csr_window_lds_budget() / (long) A->vbytes;
		long with_window = 0;
		int max_w = 0;
		for (int attempt = 0; attempt < 2; attempt++)
		{
			b_row.assign((size_t) nb + 1, 0);
			b_lo.assign((size_t) nb, 0);
			b_w.assign((size_t) nb, 0);
			b_nnz.assign((size_t) nb + 1, 0);
			for (long b = 0; b <= nb; b++)
			{
				const long target = (long) ((double) lnnz * b / nb);
				long r = std::lower_bound(rp, rp + lm + 1, (int) std::min<long>(target, 0x7fffffffL)) - rp;
				b_row[b] = (int) (b == 0 ? 0 : b == nb ? lm : std::min<long>(std::max<long>(r, b_row[b - 1]), lm));
				b_nnz[b] = rp[b_row[b]];
			}
			with_window = 0;
			max_w = 0;
			#pragma omp parallel for num_threads(spmv::host_threads()) schedule(dynamic, 4) reduction(+ : with_window) reduction(max : max_w)
			for (long b = 0; b < nb; b++)
			{
				int lo = 0x7fffffff, hi = -1;
				for (long j = rp[b_row[b]]; j < rp[b_row[b + 1]]; j++)
				{
					lo = std::min(lo, ci[j]);
					hi = std::max(hi, ci[j]);
				}
				if (hi >= 0 && (long) hi - lo + 1 <= budget)
				{
					b_lo[b] = lo;
					b_w[b] = hi - lo + 1;
					with_window++;
					max_w = std::max(max_w, b_w[b]);
				}
			}
			// One 1024-thread block per CU is half the CU's wave slots. When the window is set by the matrix's
			// bandwidth rather than by the block (it does not shrink with the block) and two of them fit the CU's
			// 160 KiB of LDS, twice the blocks put two on every CU: pwtk twin fp32 (54 KiB windows) 17.8 -> 16.2 us.
			// Small blocks lose more than they gain (cant twin: 7.8 k non-zeros per block, 9.9 -> 13.0 us).
			if (attempt == 0 && o.merge_items == 0 && nb == 256 && with_window == nb && (long) max_w * A->vbytes > 16 * 1024 &&
			    2L * (((long) max_w * A->vbytes + 15) / 16 * 16) <= 144L * 1024 && lnnz / 512 >= 20000 && lm >= 2L * 512 * NG)
			{
				nb = 512;
				continue;
			}
			break;
		}
		// 16-bit window-relative indices need every non-empty block to have a window of at most 65 536 columns
		bool eligible16 = max_w <= 65536 && lnnz > 0;
		for (long b = 0; b < nb && eligible16; b++)
			eligible16 = b_w[b] > 0 || rp[b_row[b + 1]] == rp[b_row[b]];
		if (!force && (with_window * 100 < nb * 95 || mean < 16 || lnnz < (2L << 20) || !(A->f32 || eligible16)))
			return 0;
		A->stream_mode = 4;
		A->lanes_per_row = G;
		A->win_blocks = (int) nb;
		A->win_lds_bytes = (int) (((long) max_w * A->vbytes + 15) / 16 * 16);
		if (upload_ints(b_row.data(), (size_t) nb + 1, &A->d_win_row) || upload_ints(b_lo.data(), (size_t) nb, &A->d_win_lo) ||
		    upload_ints(b_w.data(), (size_t) nb, &A->d_win_w))
			return -1;
		A->cfg.map = xcd_map_balanced(b_nnz.data(), nb, 1, resolve_remap(A->remap, nb));
		A->mem_footprint += (3.0 * nb + 1) * 4;
		// every block has its window and none is wider than 65 536 columns: store the column indices relative to the
		// window in 16 bits (2 B per non-zero of stream instead of 4) and let go of the int32 array
		const bool short_idx = eligible16;
		if (short_idx)
		{
			std::vector<unsigned short> c16((size_t) lnnz);
			#pragma omp parallel for num_threads(spmv::host_threads()) schedule(dynamic, 4)
			for (long b = 0; b < nb; b++)
				for (long j = rp[b_row[b]]; j < rp[b_row[b + 1]]; j++)
					c16[j] = (unsigned short) (ci[j] - b_lo[b]);
			if (dev_alloc_bytes((void **) &A->d_col16, ((size_t) lnnz + STREAM_SLACK) * 2))
				return -1;
			if (hipMemcpy(A->d_col16, c16.data(), (size_t) lnnz * 2, hipMemcpyHostToDevice) != hipSuccess ||
			    hipMemset(A->d_col16 + lnnz, 0, STREAM_SLACK * 2) != hipSuccess)
			{
				set_error("upload of the 16-bit column indices failed");
				return -1;
			}
			(void) hipFree(A->d_col);
			A->d_col = nullptr;
			A->mem_footprint -= 2.0 * lnnz;
		}
		snprintf(A->format_name, sizeof(A->format_name), "MI355X_CSR_WINDOW%s_g%d_b%ld_w%ld_%s", short_idx ? "16" : "", G, nb,
				with_window * 100 / nb, pf);
		snprintf(A->kernel_name, sizeof(A->kernel_name), "csr_window_kernel");
		return 1;
	};
	if (mode == 4 || (mode == 0 && R == 0))
	{
		const int took = try_window(mode == 4);
		if (took < 0)
		{
			rc = 1;
			return rc;
		}
		if (took > 0)
			return rc;
	}
	if (mode < 1 || mode > 3)
	{
		// The lane-per-row walk (modes 2/3) pays when neighbouring rows touch neighbouring columns (stencil /
		// FEM matrices: one x gather instruction then hits a few lines). Estimate that on a sample of rows; with
		// scattered columns (graphs) the row-major product staging of mode 1 is the better CSR-Stream.
		long similar = 0, tried = 0;
		const long stride = std::max<long>(1, lm / 4096);
		for (long i = 0; i + 1 < lm; i += stride)
		{
			if (rp[i + 1] == rp[i] || rp[i + 2] == rp[i + 1])
				continue;
			tried++;
			long d = (long) ci[rp[i + 1]] - ci[rp[i]];
			similar += (d >= -2 && d <= 2);
		}
		mode = (tried == 0 || 4 * similar >= tried) ? 3 : 1;      // twins: stencil/FEM 0.8-1.0, circuit 0.38, social graph 0.13
	}
	A->stream_mode = mode;
	if (R == 0)
	{
		if (mode == 1)
		{
			// largest power of two with R * mean nnz/row <= 60% of the LDS strip
			R = 16;                     // more rows per wave only lengthen the per-lane LDS walk (measured)
			while (R > 4 && R * mean > 0.6 * csr_stream_cap())
				R /= 2;
		}
		else
		{
			// largest R <= 16 whose row blocks overflow the LDS strip (slow path) in at most 0.5 % of the cases:
			// measured optimum is a block of ~150-450 non-zeros per wave (profiles/sweep_r01.md)
			for (R = (mode == 3 ? 16 : 32); R > (mode == 3 ? 4 : 8); R /= 2)
			{
				const long cap = mode == 3 ? csr_stream_d_cap(R) : csr_stream_t_cap(R);
				long over = 0, blocks = (lm + R - 1) / R;
				#pragma omp parallel for num_threads(spmv::host_threads()) reduction(+ : over)
				for (long b = 0; b < blocks; b++)
					over += (rp[std::min(lm, (b + 1) * R)] - rp[b * R]) > cap;
				if (over * 200 <= blocks)
					break;
			}
		}
	}
	const bool okR = (A->stream_mode != 2) ? (R == 4 || R == 8 || R == 16 || R == 32 || R == 64)
	                                       : (R == 8 || R == 16 || R == 32 || R == 64);
	if (!okR)
	{
		set_error("csr_stream: rows per wavefront (lanes_per_row) must be %s (got %d)",
				A->stream_mode != 2 ? "4,8,16,32 or 64" : "8,16,32 or 64", R);
		rc = 1;
		return rc;
	}
	A->lanes_per_row = R;
	{
		const long rpt = A->stream_mode == 3 ? csr_stream_d_rows_per_tile(R) : csr_stream_rows_per_tile(R);
		A->cfg.map = xcd_map_balanced(rp, lm, rpt, resolve_remap(A->remap, lm / rpt));
	}
	snprintf(A->format_name, sizeof(A->format_name), "MI355X_CSR_STREAM%s_r%d_%s",
			A->stream_mode == 3 ? "D" : A->stream_mode == 2 ? "T" : "", R, pf);
	snprintf(A->kernel_name, sizeof(A->kernel_name), A->stream_mode == 3 ? "csr_stream_d_kernel" :
			A->stream_mode == 2 ? "csr_stream_t_kernel" : "csr_stream_kernel");
	return rc;
}

static int
build_csr_vector(spmv_mi355x_matrix * A, const spmv_mi355x_opts & o, const int * rp)
{
	const long lm = A->m, lnnz = A->nnz;
	const char * pf = A->f32 ? "f" : "d";
	int rc = 0;
	int G = o.lanes_per_row;
	if (G == 0)
		G = pick_lanes_per_row(lm > 0 ? (double) lnnz / lm : 0);
	if (G != 2 && G != 4 && G != 8 && G != 16 && G != 32 && G != 64)
	{
		set_error("lanes_per_row must be 2,4,8,16,32 or 64 (got %d)", G);
		rc = 1;
		return rc;
	}
	A->lanes_per_row = G;
	int RPG = o.rows_per_group;
	if (RPG == 0)
	{
		// two rows of a lane group in flight: measured +10 % on the nlpkkt240 twin (2.90 -> 2.63 ms), +8 % on
		// scircuit as 16 lanes x 2 rows instead of 8 lanes x 1, neutral on cant / pwtk fp64; four rows cost occupancy
		if (G == 8 && o.lanes_per_row == 0)
			G = 16;
		RPG = (G == 16 || G == 32) ? 2 : 1;
		A->lanes_per_row = G;
	}
	if ((RPG != 1 && RPG != 2 && RPG != 4) || (RPG > 1 && G < 8))
	{
		set_error("rows_per_group must be 1, 2 or 4 (2 and 4 need lanes_per_row >= 8), got %d with %d lanes", RPG, G);
		rc = 1;
		return rc;
	}
	A->rows_per_group = RPG;
	const long rpt = csr_vector_rows_per_tile(G, RPG);
	A->cfg.map = xcd_map_balanced(rp, lm, rpt, resolve_remap(A->remap, lm / rpt));
	if (RPG > 1)
		snprintf(A->format_name, sizeof(A->format_name), "MI355X_CSR_VECTOR_g%d_r%d_%s", G, RPG, pf);
	else
		snprintf(A->format_name, sizeof(A->format_name), "MI355X_CSR_VECTOR_g%d_%s", G, pf);
	snprintf(A->kernel_name, sizeof(A->kernel_name), RPG > 1 ? "csr_vector_multi_kernel" : "csr_vector_kernel");
	return rc;
}

static int
build_csr_merge(spmv_mi355x_matrix * A, const spmv_mi355x_opts & o, const double * va)
{
	const long lm = A->m, lnnz = A->nnz;
	const char * pf = A->f32 ? "f" : "d";
	int rc = 0;
	A->merge_ipt = o.merge_items;
	A->merge_tile = merge_tile_items(A->f32, A->merge_ipt);
	A->merge_ipt = A->merge_tile / 256;
	long total = lm + lnnz;
	A->merge_num_tiles = (int) ((total + A->merge_tile - 1) / A->merge_tile);
	rc = dev_alloc(&A->d_coords, 2 * ((size_t) A->merge_num_tiles + 1)) ||
	     dev_alloc(&A->d_carry_row, (size_t) A->merge_num_tiles) ||
	     dev_alloc_bytes(&A->d_carry_val, (size_t) A->merge_num_tiles * A->vbytes);
	if (rc)
		return rc;
	rc = launch_merge_search(A->d_row_ptr, (int) lm, (int) lnnz, A->merge_tile, A->merge_num_tiles, A->d_coords, nullptr);
	if (rc)
		return rc;
	if (hipDeviceSynchronize() != hipSuccess)
	{
		set_error("merge tile search failed");
		rc = 1;
		return rc;
	}
	A->cfg.map = xcd_map_uniform((unsigned) A->merge_num_tiles, resolve_remap(A->remap, 0));      // tiles hold equal work by construction
	A->mem_footprint += 2.0 * (A->merge_num_tiles + 1) * 4;
	// Pattern matrices (Matrix-Market `pattern`: every value is the dummy 1.0, matrix_market.c:308-317 — the
	// soc-LiveJournal1 configuration) carry no information in the value array: keep the constant, drop the stream.
	double v0 = 0;
	const bool uniform = values_uniform(A, va, lnnz, &v0);
	if (uniform)
	{
		(void) hipFree(A->d_val);
		A->d_val = nullptr;
		A->cfg.unit = 1;
		A->cfg.unit_value = v0;
		A->mem_footprint -= (double) lnnz * A->vbytes;
	}
	if (uniform)
		snprintf(A->format_name, sizeof(A->format_name), "MI355X_CSR_MERGE_i%d_unit_%s", A->merge_ipt, pf);
	else
		snprintf(A->format_name, sizeof(A->format_name), "MI355X_CSR_MERGE_i%d_%s", A->merge_ipt, pf);
	snprintf(A->kernel_name, sizeof(A->kernel_name), "merge_kernel");
	return rc;
}

int
build_csr_family(spmv_mi355x_matrix * A, const spmv_mi355x_opts & o, const int * rp, const int * ci, const double * va)
{
	const long lm = A->m, lnnz = A->nnz;
	// merge path: col_blocks = 0 (default) / -2 = the CSR-order merge path (merge.cpp:256-319: deterministic sums); -1 / > 0 = the
	// merge-balanced column-blocked layout (kernels_coo.hip: LDS atomics, sums to tolerance) — only when the caller asks for it
	if (A->format == SPMV_MI355X_CSR_MERGE && (o.col_blocks == -1 || o.col_blocks > 0))
		return build_blocked_layout(A, rp, ci, va, o.col_blocks, true);
	if (upload_ints(rp, (size_t) lm + 1, &A->d_row_ptr) || upload_ints(ci, (size_t) lnnz, &A->d_col) ||
	    upload_values(A, va, (size_t) lnnz, &A->d_val))
		return 1;
	A->mem_footprint = A->csr_mem_footprint;
	switch (A->format)
	{
		case SPMV_MI355X_CSR_SCALAR:
			A->cfg.map = xcd_map_balanced(rp, lm, csr_scalar_rows_per_tile(), resolve_remap(A->remap, lm / csr_scalar_rows_per_tile()));
			A->cfg.kahan = o.kahan ? 1 : 0;
			snprintf(A->format_name, sizeof(A->format_name), "MI355X_CSR_SCALAR%s_%s", o.kahan ? "_KAHAN" : "", A->f32 ? "f" : "d");
			snprintf(A->kernel_name, sizeof(A->kernel_name), "csr_scalar_kernel");
			return 0;
		case SPMV_MI355X_CSR_STREAM:
			return build_csr_stream(A, o, rp, ci, va);
		case SPMV_MI355X_CSR_VECTOR:
			return build_csr_vector(A, o, rp);
		default:
			return build_csr_merge(A, o, va);
	}
}

}  // namespace spmv
