// Backend plug-in interface of the SpMV-Research benchmark harness, as this engine sees it.
//
// This header is OUR rendering of the interface the reference declares in
// benchmark_code/BENCH/src/spmv_kernels/spmv_kernel.h:8-29 (same member order, same virtual order, same two factory
// functions, so an object built against either header has the same layout). Inside the reference tree the adapter
// TU (spmv_kernel_mi355x.cpp) is compiled against the reference's own header instead; this copy exists so the
// adapter and the stand-alone driver build and run where /root/reference does not exist (the GPU box).
//
// Build-time macros, as the reference build supplies them (make.sh:166,191,212-216):
//   INT_T = int32_t, ValueType in {double,float}, ValueTypeReference = double, DOUBLE in {1,0}.
#ifndef SPMV_KERNELS_H
#define SPMV_KERNELS_H

#include <stdint.h>

#ifndef INT_T
	#define INT_T int32_t
#endif
#ifndef ValueType
	#define ValueType double
#endif
#ifndef ValueTypeReference
	#define ValueTypeReference double
#endif

struct Matrix_Format
{
	char * format_name;          // borrowed string, printed in the CSV
	long m;                      // rows
	long n;                      // columns
	long nnz;                    // stored non-zeros
	double mem_footprint;        // bytes of the backend's own format (set by the backend)
	double csr_mem_footprint;    // bytes of plain CSR in ValueType precision (set here)

	virtual void spmv(ValueType * x, ValueType * y) = 0;
	virtual void statistics_start() = 0;
	virtual int statistics_print_data(char * buf, long buf_n) = 0;

	Matrix_Format(long rows, long cols, long nonzeros) : m(rows), n(cols), nnz(nonzeros)
	{
		csr_mem_footprint = nonzeros * (sizeof(ValueType) + sizeof(INT_T)) + (rows + 1) * sizeof(INT_T);
	}
};

struct Matrix_Format * csr_to_format(INT_T * row_ptr, INT_T * col_ind, ValueTypeReference * values,
		long m, long n, long nnz, long symmetric, long symmetry_expanded);
int statistics_print_labels(char * buf, long buf_n);

#endif /* SPMV_KERNELS_H */
