// Loop partitioners with the reference's semantics (lib/parallel_util.h:47-91,156-184; the "closest value" binary
// search of lib/macros/macrolib.h:537-590). Used for the row-block partition across GPUs (the per-thread
// nnz-balanced row ranges of csr.cpp:140 applied with W = number of GPUs) and by the host-side converters.
#include <stdlib.h>

#include "host.hpp"

namespace spmv_host {

void
partition_iterations(long num_workers, long worker_pos, long start, long end, long * s, long * e)
{
	const long len = end - start;
	if (len < 1)
	{
		*s = (worker_pos == 0) ? start : end;
		*e = end;
		return;
	}
	long per = len / num_workers;
	long rem = len % num_workers;
	if (rem != 0 && worker_pos < rem)
	{
		per++;
		rem = 0;
	}
	const long ls = start + per * worker_pos + rem;
	*s = ls;
	*e = (worker_pos == num_workers - 1) ? end : ls + per;
}

// index in [lo,hi] whose value is closest to target; ties and exact hits resolved as the reference's macro does
static long
closest(const int32_t * A, long lo, long hi, long target)
{
	if (target < A[lo])
		return lo;
	if (target > A[hi])
		return hi;
	long s = lo, e = hi;
	for (;;)
	{
		long mid = (s + e) / 2;
		if (mid == s || mid == e)
			break;
		if (target > A[mid])
			s = mid;
		else
			e = mid;
	}
	if (target == A[s])
		return s;
	if (target == A[e])
		return e;
	return (labs(target - (long) A[s]) < labs(target - (long) A[e])) ? s : e;
}

void
partition_prefix_sums(long num_workers, long worker_pos, const int32_t * sums, long N, long total_sum, long * s, long * e)
{
	const long target = sums[0] + (total_sum * worker_pos) / num_workers;
	const long target_next = sums[0] + (total_sum * (worker_pos + 1)) / num_workers;
	*s = (worker_pos == 0) ? 0 : closest(sums, 0, N - 1, target);
	*e = (worker_pos == num_workers - 1) ? N : closest(sums, 0, N - 1, target_next);
}

}  // namespace spmv_host
