/* TEST INFRASTRUCTURE. The only thing of ours linked into oracle/_ref/<flavour>/libref_sellcs.so: it CALLS the reference's own
 * SELL-C-sigma format code (benchmark_code/BENCH/src/spmv_kernels/sell-C-s/RISC-V/{sellcs_format,radix_sort,sellcs_utils}.c,
 * compiled from where they lie by oracle/Makefile) the way sell_c_s.cpp:58-75 does and copies the resulting arrays out. */
#include <stdint.h>
#include <string.h>

#include "sellcs-spmv.h"

long
ref_sellcs_convert(int32_t nrows, int32_t ncols, const int32_t * rp, const int32_t * ci, const double * va, long C, long sigma,
		int32_t * row_order_out, int32_t * widths_out, int64_t * slice_ptr_out, int32_t * col_out, double * val_out, long cap)
{
	sellcs_matrix_t mtx;
	long i, total;
	memset(&mtx, 0, sizeof(mtx));
	sellcs_init_params((uint64_t) C, (uint64_t) sigma, &mtx);
	sellcs_create_matrix_from_CSR_rd(nrows, ncols, rp, ci, va, 0, 0, &mtx);
	total = (long) mtx.slice_pointers[mtx.nslices];
	for (i = 0; i < nrows; i++)
		row_order_out[i] = mtx.row_order[i];
	for (i = 0; i < mtx.nslices; i++)
		widths_out[i] = mtx.slice_widths[i];
	for (i = 0; i <= mtx.nslices; i++)
		slice_ptr_out[i] = mtx.slice_pointers[i];
	if (total <= cap)
		for (i = 0; i < total; i++)
		{
			col_out[i] = mtx.column_indices[i];
			val_out[i] = mtx.values[i];
		}
	free(mtx.values);
	free(mtx.column_indices);
	free(mtx.slice_widths);
	free(mtx.slice_pointers);
	free(mtx.row_order);
	return total;
}
