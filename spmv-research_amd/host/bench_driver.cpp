// spmv_mi355x_bench — stand-alone stand-in for the reference's SpMV driver (benchmark_code/BENCH/src/bench.cpp main()
// + bench_spmv.cpp bench()/compute()/check_accuracy()), for machines where the reference tree is absent (the GPU box).
// Same contract, so run.sh-style tooling keeps working:
//   * no argument  -> the CSV header on stderr, exit (bench.cpp:507-511)
//   * <file.mtx>   -> "time read" / "time coo to csr" / "time convert to format" on stdout, then the protocol of
//                     bench_spmv.cpp:247-487: x = ones, y = 1.0 canary with 64 elements of slack, warm-up (1000 calls
//                     when GPU_KERNEL=1), per-call CLOCK_MONOTONIC_RAW timing until >= 64 loops and >= 2.0 s,
//                     min/median/max, the _Float128 Kahan gold check, one CSV row on stderr with the reference's columns.
//   * --twin NAME [scale] -> same, on a synthetic twin of a BASELINE.json matrix (our generator).
//   * --cg | --bicgstab as FIRST argument -> the solver drivers instead (bench_cg.cpp / bench_bicg.cpp bench()+compute()):
//                     b from <matrix>_b.mtx when that file exists, else all ones (bench_cg.cpp:497-523); one solve and one
//                     CSV row (the reference's solver columns) per entry of CG_MAX_NUM_ITERS (space separated,
//                     bench_cg.cpp:527-553; the reference exits when it is unset, here it defaults to "1000"); the solve
//                     is the device-resident spmv_mi355x_pcg / spmv_mi355x_pbicgstab of the C ABI.
//   * KEEP_SYMMETRY=1 (environment) -> the reference's -DKEEP_SYMMETRY build: symmetric files are passed on un-expanded
//                     (csr_to_format(..., symmetric, 0)) and checked by the symmetric branch of check_accuracy.
// Environment variables of the reference are honoured when set (GPU_KERNEL, CLEAR_CACHES is N/A on the GPU, PROGG);
// unlike the reference they may be absent (it dereferences getenv() unchecked: SURVEY §5).
// Deliberate differences, both reported: GFLOPS uses the TRUE stored nnz (2*nnz/t; the reference multiplies general
// matrices by ~2, Q4) — the reference-convention number is printed beside it on stdout.

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include <time.h>
#include <algorithm>
#include <string>
#include <vector>
#include <omp.h>
#include <quadmath.h>

#include "spmv_kernel.h"
#include "spmv_host.h"
#include "spmv_mi355x.h"
#include "../csrc/host_threads.hpp"

extern "C" spmv_mi355x_matrix * spmv_mi355x_handle_of(struct Matrix_Format * MF);      // spmv_kernel_mi355x.cpp

static double
now()
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC_RAW, &t);       // lib/time_it.h:35-57
	return t.tv_sec + 1e-9 * t.tv_nsec;
}

static int
env_int(const char * name, int dflt)
{
	const char * s = getenv(name);
	return (s && *s) ? atoi(s) : dflt;
}

static const char * SOLVER_LABELS =      // bench_cg.cpp:423-440
	"matrix_name,num_threads,csr_m,csr_n,csr_nnz,time,error,num_iterations,csr_mem_footprint,W_avg,J_estimated,"
	"format_name,m,n,nnz,mem_footprint,mem_ratio";

static const char * AM_LABELS =          // bench_spmv.cpp:493-522 (artificial-matrix mode)
	"matrix_name,distribution,placement,seed,nr_rows,nr_cols,nr_nzeros,density,mem_footprint,mem_range,avg_nnz_per_row,std_nnz_per_row,"
	"avg_bw,std_bw,avg_bw_scaled,std_bw_scaled,avg_sc,std_sc,avg_sc_scaled,std_sc_scaled,skew,avg_num_neighbours,cross_row_similarity,"
	"format_name,time,gflops,W_avg,J_estimated";
static const char * LABELS =
	"matrix_name,num_threads,csr_m,csr_n,csr_nnz,symmetry,time,time_iter_min,time_iter_median,time_iter_max,gflops,"
	"csr_mem_footprint,W_avg,J_estimated,format_name,m,n,nnz,mem_footprint,mem_ratio,num_loops,"
	"spmv_mae,spmv_max_ae,spmv_mse,spmv_mape,spmv_smape,spmv_lnQ_error,spmv_mlare,spmv_gmare";

// bench_spmv.cpp:108-235: gold = per-row Kahan sum in _Float128; "Test failed" when the max relative difference over rows
// with y_gold > eps exceeds 1e-10 (fp64) / 1e-7 (fp32); then the eight array metrics (lib/array_metrics.c:1477-2149).
static int
check_accuracy(char * buf, long buf_n, const INT_T * ia, const INT_T * ja, const double * a, long m,
		const double * x_ref, const ValueType * y, bool symmetric_unexpanded)
{
	const __float128 epsilon = (sizeof(ValueType) == 8) ? (__float128) 1e-10 : (__float128) 1e-7;
	std::vector<__float128> gold((size_t) std::max<long>(m, 1), 0);
	if (symmetric_unexpanded)
	{
		// KEEP_SYMMETRY branch (bench_spmv.cpp:135-148): plain quad accumulation, every off-diagonal entry counted twice
		for (long i = 0; i < m; i++)
			for (long j = ia[i]; j < ia[i + 1]; j++)
			{
				const long col = ja[j];
				gold[i] += (__float128) a[j] * (__float128) x_ref[col];
				if (i != col)
					gold[col] += (__float128) a[j] * (__float128) x_ref[i];
			}
	}
	else
	#pragma omp parallel for num_threads(spmv::host_threads())
	for (long i = 0; i < m; i++)
	{
		__float128 sum = 0, comp = 0;
		for (long j = ia[i]; j < ia[i + 1]; j++)
		{
			__float128 val = (__float128) a[j] * (__float128) x_ref[ja[j]] - comp;
			__float128 tmp = sum + val;
			comp = (tmp - sum) - val;
			sum = tmp;
		}
		gold[i] = sum;
	}
	__float128 max_diff = 0;
	for (long i = 0; i < m; i++)
		if (gold[i] > epsilon)
		{
			__float128 d = fabsq(gold[i] - (__float128) y[i]) / fabsq(gold[i]);
			if (d > max_diff)
				max_diff = d;
		}
	if (max_diff > epsilon)
		printf("Test failed! (%g)\n", (double) max_diff);
	double mae = 0, max_ae = 0, mse = 0, mare = 0, smare = 0, lnq = 0;
	for (long i = 0; i < m; i++)
	{
		double A = (double) gold[i], F = (double) (__float128) y[i];
		double ae = fabs(A - F);
		mae += ae;
		max_ae = std::max(max_ae, ae);
		mse += (A - F) * (A - F);
		mare += ae / std::max(fabs(A), DBL_EPSILON);
		smare += ae / std::max(fabs(A) + fabs(F), DBL_EPSILON);
		lnq += log10(std::max(fabs(F), DBL_EPSILON)) - log10(std::max(fabs(A), DBL_EPSILON));
	}
	const double N = (double) m;
	mae /= N; mse /= N;
	const double mape = 100.0 * mare / N, smape = 100.0 * smare / N, lnQ = lnq / N;
	const double mlare = (double) log10l(fabsl(powl(10, (long double) lnQ) - 1));
	const double gmare = pow(10, mlare);
	printf("errors spmv: mae=%g, max_ae=%g, mse=%g, mape=%g, smape=%g, lnQ_error=%g, mlare=%g, gmare=%g\n",
			mae, max_ae, mse, mape, smape, lnQ, mlare, gmare);
	return snprintf(buf, buf_n, ",%g,%g,%g,%g,%g,%g,%g,%g", mae, max_ae, mse, mape, smape, lnQ, mlare, gmare);
}

int
main(int argc, char ** argv)
{
	int num_threads = spmv::host_threads();
	printf("max threads %d\n", num_threads);
	int solver = 0;                               // 0 = SpMV protocol, 1 = CG, 2 = BiCGSTAB
	if (argc > 1 && (!strcmp(argv[1], "--cg") || !strcmp(argv[1], "--bicgstab")))
	{
		solver = !strcmp(argv[1], "--cg") ? 1 : 2;
		argv++;
		argc--;
	}
	// USE_ARTIFICIAL_MATRICES=1 (config.sh conf_vars; bench.cpp:497): argv holds the generator's feature vector instead of a file
	// name (bench.cpp:569-579) and the CSV row / label line are the synthetic-dataset ones (bench_spmv.cpp:489-563)
	const int use_artificial = env_int("USE_ARTIFICIAL_MATRICES", 0);
	if (argc == 1)
	{
		fprintf(stderr, "%s\n", solver ? SOLVER_LABELS : use_artificial ? AM_LABELS : LABELS);
		return 0;
	}

	long m = 0, n = 0, nnz = 0, symmetric = 0, nnz_diag = 0, nnz_non_diag = 0;
	int keep_symmetry = env_int("KEEP_SYMMETRY", 0);
	std::vector<INT_T> ia, ja;
	std::vector<double> a_ref;
	char matrix_name[1000];
	double t;

	double am[15] = {0};
	char am_range[32] = "", am_distribution[64] = "", am_placement[64] = "";
	long am_seed = 0;
	if (use_artificial && strcmp(argv[1], "--twin"))
	{
		// nr_rows nr_cols avg_nnz_per_row std_nnz_per_row distribution placement avg_bw skew avg_num_neighbours cross_row_similarity seed
		// [matrix_name]. The reference's generator is an un-vendored submodule: the matrix comes from OUR generator driven by the
		// same feature vector (host/synthetic.cpp); `distribution` and `placement` are echoed, the generator has one of each
		// (normal row lengths — log-normal when std > avg/2 — and random placement inside the band).
		if (argc < 12)
		{
			fprintf(stderr, "usage (USE_ARTIFICIAL_MATRICES=1): %s nr_rows nr_cols avg_nnz_per_row std_nnz_per_row distribution placement "
					"avg_bw skew avg_num_neighbours cross_row_similarity seed [matrix_name]\n", argv[0]);
			return 1;
		}
		const long nr_rows = atol(argv[1]), nr_cols = atol(argv[2]);
		snprintf(am_distribution, sizeof(am_distribution), "%s", argv[5]);
		snprintf(am_placement, sizeof(am_placement), "%s", argv[6]);
		am_seed = atol(argv[11]);
		spmv_host_csr csr;
		t = now();
		if (spmv_host_gen_twin(nr_rows, nr_cols, atof(argv[3]), atof(argv[4]), atof(argv[7]), atof(argv[8]), atof(argv[9]), atof(argv[10]),
				(unsigned long) am_seed, 0, &csr))
		{
			fprintf(stderr, "%s\n", spmv_host_last_error());
			return 1;
		}
		printf("time generate artificial matrix: %lf\n", now() - t);
		keep_symmetry = 0;
		m = csr.m; n = csr.n; nnz = csr.nnz;
		ia.assign(csr.row_ptr, csr.row_ptr + m + 1);
		ja.assign(csr.col_idx, csr.col_idx + nnz);
		a_ref.assign(csr.values, csr.values + nnz);
		spmv_host_csr_free(&csr);
		for (long i = 0; i < m; i++)
			for (long j = ia[i]; j < ia[i + 1]; j++)
				(ja[j] == i ? nnz_diag : nnz_non_diag)++;
		spmv_host_csr_am_stats(ia.data(), ja.data(), m, n, am, am_range, sizeof(am_range));
		if (argc > 12)
			snprintf(matrix_name, sizeof(matrix_name), "%s_artificial", argv[12]);
		else
			snprintf(matrix_name, sizeof(matrix_name), "%ld_%ld_%ld_%g_%g_%g_%g", m, n, nnz, am[4], am[5], am[8], am[9]);      // bench.cpp:583-585
	}
	else if (!strcmp(argv[1], "--twin"))
	{
		if (argc < 3)
		{
			fprintf(stderr, "usage: %s --twin <cant|scircuit|pwtk|soc-LiveJournal1|nlpkkt240> [scale]\n", argv[0]);
			return 1;
		}
		double scale = argc > 3 ? atof(argv[3]) : 1.0;
		spmv_host_csr csr;
		t = now();
		if (spmv_host_gen_named(argv[2], scale, &csr))
		{
			fprintf(stderr, "%s\n", spmv_host_last_error());
			return 1;
		}
		printf("time generate twin: %lf\n", now() - t);
		keep_symmetry = 0;                                               // the twins are general matrices
		m = csr.m; n = csr.n; nnz = csr.nnz;
		ia.assign(csr.row_ptr, csr.row_ptr + m + 1);
		ja.assign(csr.col_idx, csr.col_idx + nnz);
		a_ref.assign(csr.values, csr.values + nnz);
		spmv_host_csr_free(&csr);
		for (long i = 0; i < m; i++)
			for (long j = ia[i]; j < ia[i + 1]; j++)
				(ja[j] == i ? nnz_diag : nnz_non_diag)++;
		snprintf(matrix_name, sizeof(matrix_name), "%s_twin", argv[2]);
	}
	else
	{
		// import_file(): bench.cpp:126-239
		spmv_host_coo coo;
		t = now();
		if (spmv_host_mtx_read(argv[1], &coo))
		{
			fprintf(stderr, "%s\n", spmv_host_last_error());
			return 1;
		}
		printf("time read: %lf\n", now() - t);
		m = coo.m; n = coo.n; nnz = coo.nnz; symmetric = coo.symmetric;
		// KEEP_SYMMETRY=1 in the environment = the reference's -DKEEP_SYMMETRY build (bench.cpp:131-136,186-192): only the
		// file's own entries (the first nnz_sym of the expanded list) go on, flagged symmetric and NOT expanded
		if (keep_symmetry && symmetric)
			nnz = coo.nnz_sym;
		else
			keep_symmetry = 0;
		nnz_diag = coo.nnz_diag; nnz_non_diag = coo.nnz_non_diag;
		t = now();
		ia.assign((size_t) m + 1, 0);
		ja.assign((size_t) std::max<long>(nnz, 1), 0);
		a_ref.assign((size_t) std::max<long>(nnz, 1), 0.0);
		if (spmv_host_coo_to_csr(coo.R, coo.C, coo.V, m, n, nnz, ia.data(), ja.data(), a_ref.data()))
		{
			fprintf(stderr, "%s\n", spmv_host_last_error());
			return 1;
		}
		spmv_host_coo_free(&coo);
		printf("time coo to csr: %lf\n", now() - t);
		snprintf(matrix_name, sizeof(matrix_name), "%s", argv[1]);
	}
	const long nnz_expanded_symmetry = 2 * nnz_non_diag + nnz_diag;      // bench.cpp:212 (quirk Q4)

	t = now();
	struct Matrix_Format * MF = csr_to_format(ia.data(), ja.data(), a_ref.data(), m, n, nnz, symmetric, keep_symmetry ? 0 : 1);
	printf("time convert to format: %lf\n", now() - t);

	// "Reallocate CSR arrays to ensure the format does not rely on them" (bench.cpp:605-629)
	{
		std::vector<INT_T> ia2(ia), ja2(ja);
		std::vector<double> a2(a_ref);
		std::fill(ia.begin(), ia.end(), -1);
		std::fill(ja.begin(), ja.end(), -1);
		ia.swap(ia2); ja.swap(ja2); a_ref.swap(a2);
	}

	if (solver)
	{
		if (m != n)
		{
			fprintf(stderr, "the matrix must be square\n");               // bench_cg.cpp:487-488
			return 1;
		}
		// b: <matrix>_b.mtx (an n x 1 matrix) when it exists, else ones (bench_cg.cpp:497-523)
		std::vector<ValueType> b((size_t) std::max<long>(n, 1), (ValueType) 1.0), x_out((size_t) std::max<long>(n, 1));
		if (strcmp(argv[1], "--twin"))
		{
			std::string file_b(argv[1]);
			if (file_b.size() > 4)
				file_b = file_b.substr(0, file_b.size() - 4) + "_b.mtx";
			printf("%s\n", file_b.c_str());
			if (FILE * f = fopen(file_b.c_str(), "r"))
			{
				fclose(f);
				spmv_host_coo vb;
				t = now();
				if (spmv_host_mtx_read(file_b.c_str(), &vb) || vb.m != n || vb.n != 1)
				{
					fprintf(stderr, "%s: not an %ld x 1 Matrix Market file (%s)\n", file_b.c_str(), n, spmv_host_last_error());
					return 1;
				}
				std::fill(b.begin(), b.end(), (ValueType) 0);
				for (long j = 0; j < vb.nnz; j++)
					b[vb.R[j]] = (ValueType) vb.V[j];
				spmv_host_coo_free(&vb);
				printf("read vector file time = %lf\n", now() - t);
			}
		}
		const char * list = getenv("CG_MAX_NUM_ITERS");
		if (!list || !*list)
			list = "1000";
		spmv_mi355x_matrix * A = spmv_mi355x_handle_of(MF);
		for (const char * p = list; *p;)
		{
			const long max_num_loops = atol(p);
			spmv_mi355x_solver_info info;
			memset(&info, 0, sizeof(info));
			info.struct_size = sizeof(info);
			int rc = solver == 1
			         ? spmv_mi355x_pcg(A, ia.data(), ja.data(), a_ref.data(), b.data(), x_out.data(), max_num_loops, NULL, &info)
			         : spmv_mi355x_pbicgstab(A, ia.data(), ja.data(), a_ref.data(), b.data(), x_out.data(), max_num_loops, NULL, &info);
			if (rc)
			{
				fprintf(stderr, "%s\n", spmv_mi355x_last_error());
				return 1;
			}
			printf("eps = %g eps_counter = %g\n", info.eps, info.eps_counter);
			printf("error = %-12.4g\n", info.error);
			char buf[10000];
			long i = 0;
			i += snprintf(buf + i, sizeof(buf) - i, "%s,%d,%lu,%lu,%lu", matrix_name, num_threads, m, n, nnz);
			i += snprintf(buf + i, sizeof(buf) - i, ",%lf,%g,%ld", info.seconds, info.error, info.iterations);
			i += snprintf(buf + i, sizeof(buf) - i, ",%lf,%lf,%lf", MF->csr_mem_footprint / (1024 * 1024), 0.0, 0.0);
			i += snprintf(buf + i, sizeof(buf) - i, ",%s,%lu,%lu,%lu", MF->format_name, MF->m, MF->n, MF->nnz);
			i += snprintf(buf + i, sizeof(buf) - i, ",%lf,%lf", MF->mem_footprint / (1024 * 1024), MF->mem_footprint / MF->csr_mem_footprint);
			fprintf(stderr, "%s\n", buf);
			while (*p && *p != ' ')                                      // next entry (bench_cg.cpp:541-552)
				p++;
			while (*p == ' ')
				p++;
		}
		return 0;
	}

	// bench(): bench_spmv.cpp:598-609
	std::vector<double> x_ref((size_t) std::max<long>(n, 1), 1.0);
	ValueType * x = (ValueType *) aligned_alloc(64, ((size_t) n * sizeof(ValueType) + 63) / 64 * 64 + 64);
	ValueType * y = (ValueType *) aligned_alloc(64, ((size_t) (m + 64) * sizeof(ValueType) + 63) / 64 * 64);
	for (long i = 0; i < n; i++)
		x[i] = 1.0;
	for (long i = 0; i < m + 64; i++)
		y[i] = 1.0;

	// compute(): bench_spmv.cpp:287-301 warm-up, :335-370 timed loop
	const long min_loops = 64;
	const double min_runtime = 2.0;
	const int gpu_kernel = env_int("GPU_KERNEL", 1);
	t = now();
	MF->spmv(x, y);                                   // the first call builds the engine's vectors and runs its placement pass
	{
		// The pass returns ~165 GiB to the driver, which clears it in the background for the next seconds; kernels launched into that
		// run up to 5.5 % slower in stretches (profiles/r02_placement.md §6). Wait it out before anything is timed (not in the
		// reference: its backends allocate nothing after the constructor).
		const double idle = getenv("SPMV_MI355X_IDLE_AFTER_PLACEMENT") ? atof(getenv("SPMV_MI355X_IDLE_AFTER_PLACEMENT")) : 6.0;
		const char * pl = getenv("SPMV_MI355X_PLACEMENT");
		const size_t vbytes = sizeof(ValueType);
		if (idle > 0 && (size_t) m * vbytes >= ((size_t) 8 << 20) && !(pl && atoi(pl) == 0) && now() - t > 0.15)
		{
			struct timespec ts = {(time_t) idle, (long) ((idle - (double) (time_t) idle) * 1e9)};
			nanosleep(&ts, nullptr);
		}
	}
	for (int i = 1; i < (gpu_kernel ? 1000 : 1); i++)
		MF->spmv(x, y);
	printf("time warm up %lf\n", now() - t);
	std::vector<double> iter_times;
	double time_total = 0;
	long num_loops = 0;
	while (time_total < min_runtime || num_loops < min_loops)
	{
		double t0 = now();
		MF->spmv(x, y);
		double dt = now() - t0;
		iter_times.push_back(dt);
		time_total += dt;
		num_loops++;
	}
	printf("number of loops = %ld\n", num_loops);
	std::sort(iter_times.begin(), iter_times.end());
	const double time_min = iter_times[0], time_median = iter_times[num_loops / 2], time_max = iter_times[num_loops - 1];
	printf("time iter: min=%g, median=%g, max=%g\n", time_min, time_median, time_max);
	const double gflops = nnz / time_median * 2 * 1e-9;
	const double gflops_refconv = nnz_expanded_symmetry / time_median * 2 * 1e-9;
	printf("GFLOPS = %lf (%s)   [reference convention: %lf]\n", gflops, getenv("PROGG") ? getenv("PROGG") : "", gflops_refconv);

	char buf[10000];
	long i = 0;
	if (use_artificial && am_distribution[0])
	{
		// bench_spmv.cpp:532-559 (the accuracy check still runs and prints to stdout; this row carries no error columns)
		i += snprintf(buf + i, sizeof(buf) - i, "synthetic,%s,%s,%ld,%lu,%lu,%lu", am_distribution, am_placement, am_seed, m, n, nnz);
		i += snprintf(buf + i, sizeof(buf) - i, ",%lf,%lf,%s", am[0], am[1], am_range);
		for (int k = 2; k < 15; k++)
			i += snprintf(buf + i, sizeof(buf) - i, ",%lf", am[k]);
		i += snprintf(buf + i, sizeof(buf) - i, ",%s,%lf,%lf,%lf,%lf", MF->format_name, time_total, gflops, 0.0, 0.0);
		char scratch[2000];
		check_accuracy(scratch, sizeof(scratch), ia.data(), ja.data(), a_ref.data(), m, x_ref.data(), y, false);
		fprintf(stderr, "%s\n", buf);
		free(x);
		free(y);
		return 0;
	}
	i += snprintf(buf + i, sizeof(buf) - i, "%s,%d,%lu,%lu,%lu,%lu", matrix_name, num_threads, m, n, nnz, symmetric);
	i += snprintf(buf + i, sizeof(buf) - i, ",%lf,%lf,%lf,%lf,%lf", time_total, time_min, time_median, time_max, gflops);
	i += snprintf(buf + i, sizeof(buf) - i, ",%lf,%lf,%lf", MF->csr_mem_footprint / (1024 * 1024), 0.0, 0.0);
	i += snprintf(buf + i, sizeof(buf) - i, ",%s,%lu,%lu,%lu", MF->format_name, MF->m, MF->n, MF->nnz);
	i += snprintf(buf + i, sizeof(buf) - i, ",%lf,%lf,%ld", MF->mem_footprint / (1024 * 1024), MF->mem_footprint / MF->csr_mem_footprint, num_loops);
	i += check_accuracy(buf + i, sizeof(buf) - i, ia.data(), ja.data(), a_ref.data(), m, x_ref.data(), y, keep_symmetry != 0);
	fprintf(stderr, "%s\n", buf);
	free(x);
	free(y);
	return 0;
}
