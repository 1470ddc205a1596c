// How many host threads a parallel region of this library may use.
//
// A GPU node hands a job a SHARE of its cores through a cgroup CPU quota (measured on the MI355X boxes: 256 cores
// visible, cpu.max = "1600000 100000" = 16 cores per 100 ms period). omp_get_max_threads() still says 256; 256 threads
// on a 16-core quota burn the whole period's budget in ~6 ms and the kernel then freezes EVERY thread of the process —
// including the one feeding the GPU queue — until the next period. Measured: a CG loop whose device time is 0.26 ms
// per iteration ran at 0.8 ms per iteration with the queue starved in 80-100 ms gaps (cpu.stat: nr_throttled 35).
// LLVM's libomp makes it worse by spinning its workers for KMP_BLOCKTIME = 200 ms after every region.
// So: regions take num_threads(host_threads()) = min(OpenMP's limit, affinity mask, cgroup quota), and under clang/libomp
// the calling thread's blocktime is set to 0 (workers sleep as soon as a region ends).
#pragma once

#include <omp.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>

namespace spmv {

inline int
cgroup_cpu_limit()
{
	long quota = -1, period = -1;
	if (FILE * f = fopen("/sys/fs/cgroup/cpu.max", "r"))                       // cgroup v2
	{
		char q[64];
		if (fscanf(f, "%63s %ld", q, &period) == 2 && q[0] != 'm')
			quota = atol(q);
		fclose(f);
	}
	else
	{
		if (FILE * g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"))          // cgroup v1
		{
			if (fscanf(g, "%ld", &quota) != 1)
				quota = -1;
			fclose(g);
		}
		if (FILE * g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r"))
		{
			if (fscanf(g, "%ld", &period) != 1)
				period = -1;
			fclose(g);
		}
	}
	if (quota <= 0 || period <= 0)
		return 1 << 20;
	long n = (quota + period - 1) / period;
	return n < 1 ? 1 : (int) n;
}

inline int
host_threads()
{
	static const int limit = [] {
		int n = omp_get_max_threads();
		if (const char * e = getenv("SPMV_MI355X_HOST_THREADS"))
			if (atoi(e) > 0)
				return atoi(e);
		cpu_set_t set;
		if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0 && CPU_COUNT(&set) < n)
			n = CPU_COUNT(&set);
		const int q = cgroup_cpu_limit();
		return n < q ? n : q;
	}();
#if defined(__clang__)
	static thread_local bool blocktime_set = false;
	if (!blocktime_set)
	{
		kmp_set_blocktime(0);
		blocktime_set = true;
	}
#endif
	const int now = omp_get_max_threads();            // honours a later omp_set_num_threads() of the host program
	return now < limit ? now : limit;
}

}  // namespace spmv
