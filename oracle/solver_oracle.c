/* TEST INFRASTRUCTURE — CPU restatement of the reference's iterative-solver callers of Matrix_Format::spmv().
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
 *
 * What is restated (paths under /root/reference/benchmark_code/BENCH/src):
 *   orc_pcg_f64        preconditioned_cg()        bench_cg.cpp:93-322    Jacobi-preconditioned CG with the explicit-residual
 *                                                                      check / restart every 100 iterations
 *   orc_pbicgstab_f64  preconditioned_bicgstab()  bench_bicg.cpp:149-459 Jacobi-preconditioned BiCGSTAB (never breaks early)
 *
 * PARITY UNPINNED: bench_cg.cpp / bench_bicg.cpp include artificial_matrix_generation.h, which is absent from the
 * reference tree, so neither translation unit can be compiled here and the reference holds no solver fixtures. The
 * restatement follows the source statement by statement with the reduction order of ONE OpenMP thread (vector_dot's
 * `partial += x1[i]*x2[i]` left to right, bench_cg.cpp:66-80); it is checked in tests/ through solver properties
 * (A*x = b on SPD systems against a dense solve, monotone error_best) rather than against reference output.
 * SpMV inside is the reference CPU CSR kernel (csr.cpp:334-350): one FMA per non-zero, left to right.
 */

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

static void
spmv_seq(const int * row_ptr, const int * col, const double * val, long m, const double * x, double * y)
{
	for (long i = 0; i < m; i++)
	{
		double sum = 0;
		for (long j = row_ptr[i]; j < row_ptr[i + 1]; j++)
			sum = fma(val[j], x[col[j]], sum);
		y[i] = sum;
	}
}

/* vector_pw_add (bench_cg.cpp:49-56): y = x1 + a * x2 */
static void
pw_add(double * y, const double * x1, double a, const double * x2, long N)
{
	for (long i = 0; i < N; i++)
		y[i] = x1[i] + a * x2[i];
}

/* vector_dot (bench_cg.cpp:66-80) with one thread */
static double
dot(const double * a, const double * b, long N)
{
	double s = 0;
	for (long i = 0; i < N; i++)
		s += a[i] * b[i];
	return s;
}

static double
norm2(const double * a, long N)
{
	return sqrt(dot(a, a, N));
}

/* Jacobi preconditioner K = diag(A): first entry of the row whose column equals the row (bench_cg.cpp:114-134).
 * Returns 0, or 1 for "bad K, zero in diagonal". */
static int
jacobi_diag(const int * row_ptr, const int * col, const double * val, long m, double * K)
{
	for (long i = 0; i < m; i++)
	{
		K[i] = 0;
		for (long j = row_ptr[i]; j < row_ptr[i + 1]; j++)
			if (col[j] == i)
			{
				K[i] = val[j];
				break;
			}
		if (K[i] == 0)
			return 1;
	}
	return 0;
}

/* history (may be NULL): 3 doubles per executed loop body — error, error_explicit, error_best — i.e. exactly the numbers
 * of the reference's per-iteration printf (bench_cg.cpp:249). info_out (may be NULL): {eps, eps_counter, err_best,
 * restarts}. Returns num_loops_out (bench_cg.cpp:315) or -1 (zero on the diagonal), -2 (not square). */
long
orc_pcg_f64(const int * row_ptr, const int * col, const double * val, long m, long n, const double * b, double * x_out,
		long max_iterations, double * history, double * info_out)
{
	if (m != n)
		return -2;                                                       /* bench_cg.cpp:487-488 */
	double * rk = malloc(m * sizeof(double)), * rk_explicit = malloc(m * sizeof(double));
	double * pk = malloc(m * sizeof(double)), * zk = malloc(m * sizeof(double));
	double * Ap = malloc(m * sizeof(double)), * K = malloc(m * sizeof(double));
	double * x = calloc(n, sizeof(double)), * x_best = calloc(n, sizeof(double));
	long k = -1, restarts = 0;
	double eps = 1.0e-15, eps_counter = 1.0e-7, err, err_explicit, err_best;

	if (jacobi_diag(row_ptr, col, val, m, K))
		goto out;

	spmv_seq(row_ptr, col, val, m, x, Ap);                               /* r0 = b - A*x0, x0 = 0 (:136-150) */
	pw_add(rk, b, -1, Ap, m);
	for (long i = 0; i < m; i++)
		zk[i] = rk[i] / K[i];                                            /* solve K*z0 = r0 (:153-154) */
	memcpy(pk, zk, m * sizeof(double));                                  /* p0 = z0 (:157) */

	err = norm2(rk, m);                                                  /* :163-174 */
	{
		double b_norm = norm2(b, m);
		eps *= b_norm;
		eps_counter *= b_norm;
	}
	k = 0;
	const long restart_k = 100;
	err_explicit = err;
	err_best = err;
	while (k < max_iterations)
	{
		if (k > 0 && !(k % restart_k))                                   /* explicit residual (:186-207) */
		{
			spmv_seq(row_ptr, col, val, m, x, Ap);
			pw_add(rk_explicit, b, -1, Ap, m);
			err_explicit = norm2(rk_explicit, m);
			if (err_explicit < err_best)
			{
				memcpy(x_best, x, n * sizeof(double));
				err_best = err_explicit;
			}
		}
		err = norm2(rk, m);                                              /* :209-214 */
		if (k > 0 && !(k % restart_k))                                   /* restart (:217-236) */
		{
			if (err_best > eps_counter && err_explicit / err > 1e3)
			{
				memcpy(rk, rk_explicit, m * sizeof(double));
				for (long i = 0; i < m; i++)
					zk[i] = rk[i] / K[i];
				memcpy(pk, zk, m * sizeof(double));
				restarts++;
			}
		}
		if (err < eps)                                                   /* :238-239 */
			break;
		if (history)
		{
			history[3 * k + 0] = err;
			history[3 * k + 1] = err_explicit;
			history[3 * k + 2] = err_best;
		}
		spmv_seq(row_ptr, col, val, m, pk, Ap);                          /* A * pk (:252) */
		{
			double old_zr = dot(zk, rk, m);                              /* :259 */
			double ak = dot(zk, rk, m) / dot(pk, Ap, m);                 /* :262 */
			pw_add(x, x, ak, pk, m);                                     /* :265 */
			pw_add(rk, rk, -ak, Ap, m);                                  /* :268 */
			for (long i = 0; i < m; i++)
				zk[i] = rk[i] / K[i];                                    /* :271-274 */
			double bk = dot(zk, rk, m) / old_zr;                         /* :278 */
			pw_add(pk, zk, bk, pk, m);                                   /* :283 */
		}
		k++;
	}
	spmv_seq(row_ptr, col, val, m, x, Ap);                               /* final explicit residual (:288-306) */
	pw_add(rk_explicit, b, -1, Ap, m);
	err_explicit = norm2(rk_explicit, m);
	if (err_explicit < err_best)
	{
		memcpy(x_best, x, n * sizeof(double));
		err_best = err_explicit;
	}
	memcpy(x_out, x_best, n * sizeof(double));                           /* :308-309 */
	if (info_out)
	{
		info_out[0] = eps;
		info_out[1] = eps_counter;
		info_out[2] = err_best;
		info_out[3] = (double) restarts;
	}
out:
	free(rk); free(rk_explicit); free(pk); free(zk); free(Ap); free(K); free(x); free(x_best);
	return k;
}

/* Preconditioned BiCGSTAB (bench_bicg.cpp:149-459). The loop never breaks before max_iterations (the `err < eps` test
 * is commented out, :319-320) and there is no restart; every 100 iterations the explicit residual may promote x_best. */
long
orc_pbicgstab_f64(const int * row_ptr, const int * col, const double * val, long m, long n, const double * b, double * x_out,
		long max_iterations, double * history, double * info_out)
{
	if (m != n)
		return -2;                                                       /* bench_bicg.cpp:626-627 */
	double * r0_ = malloc(m * sizeof(double)), * rk = malloc(m * sizeof(double)), * rk_explicit = malloc(m * sizeof(double));
	double * pk = malloc(m * sizeof(double)), * z = malloc(m * sizeof(double)), * h = malloc(m * sizeof(double));
	double * s = malloc(m * sizeof(double)), * v = malloc(m * sizeof(double)), * buf = malloc(m * sizeof(double));
	double * K = malloc(m * sizeof(double));
	double * xk = calloc(n, sizeof(double)), * x_best = calloc(n, sizeof(double));
	double * y = buf, * t = buf;                                         /* :179-180 */
	long k = -1;
	double eps = 1.0e-15, eps_counter = 1.0e-7, err, err_explicit, err_best;
	double s_a = 0, s_pk_p;

	if (jacobi_diag(row_ptr, col, val, m, K))
		goto out;

	spmv_seq(row_ptr, col, val, m, xk, buf);                             /* rk = b - A*xk (:225-229) */
	pw_add(rk, b, -1, buf, m);
	memcpy(r0_, rk, m * sizeof(double));                                 /* r0_ = rk (:232-234) */
	s_pk_p = dot(r0_, rk, m);                                            /* :237-241 */
	memcpy(pk, rk, m * sizeof(double));                                  /* :244-246 */

	err = norm2(rk, m);                                                  /* :254-265 */
	{
		double b_norm = norm2(b, m);
		eps *= b_norm;
		eps_counter *= b_norm;
	}
	k = 0;
	const long restart_k = 100;
	err_explicit = err;
	err_best = err;
	while (k < max_iterations)
	{
		if (k > 0 && !(k % restart_k))                                   /* :277-302 */
		{
			spmv_seq(row_ptr, col, val, m, xk, buf);
			pw_add(rk_explicit, b, -1, buf, m);
			err_explicit = norm2(rk_explicit, m);
			if (err_explicit < err_best)
			{
				memcpy(x_best, xk, n * sizeof(double));
				err_best = err_explicit;
			}
		}
		err = norm2(rk, m);                                              /* :304-311 */
		if (history)
		{
			history[3 * k + 0] = err;
			history[3 * k + 1] = err_explicit;
			history[3 * k + 2] = err_best;
		}
		for (long i = 0; i < m; i++)
			y[i] = pk[i] / K[i];                                         /* y = inv(K) pk (:328-332) */
		spmv_seq(row_ptr, col, val, m, y, v);                            /* v = A y (:335) */
		s_a = s_pk_p / dot(r0_, v, m);                                   /* :343 */
		pw_add(h, xk, s_a, y, m);                                        /* :350 */
		pw_add(s, rk, -s_a, v, m);                                       /* :353 */
		for (long i = 0; i < m; i++)
			z[i] = s[i] / K[i];                                          /* :358-360 */
		spmv_seq(row_ptr, col, val, m, z, t);                            /* t = A z (:366); overwrites y */
		double s_w;
		{
			double p1 = 0, p2 = 0;                                       /* :374-391 */
			for (long i = 0; i < m; i++)
			{
				double v1 = t[i] / K[i];
				double v2 = s[i] / K[i];
				p1 += v1 * v2;
				p2 += v1 * v1;
			}
			s_w = p1 / p2;
		}
		pw_add(rk, s, -s_w, t, m);                                       /* :394 */
		pw_add(xk, h, s_w, z, m);                                        /* :397 */
		double s_pk = dot(r0_, rk, m);                                   /* :402 */
		double s_b = (s_pk / s_pk_p) * (s_a / s_w);                      /* :405 */
		for (long i = 0; i < m; i++)
			pk[i] = rk[i] + s_b * (pk[i] - s_w * v[i]);                  /* :408-412 */
		s_pk_p = s_pk;                                                   /* :415-419 */
		k++;
	}
	spmv_seq(row_ptr, col, val, m, xk, buf);                             /* :426-445 */
	pw_add(rk_explicit, b, -1, buf, m);
	err_explicit = norm2(rk_explicit, m);
	if (err_explicit < err_best)
	{
		memcpy(x_best, xk, n * sizeof(double));
		err_best = err_explicit;
	}
	memcpy(x_out, x_best, n * sizeof(double));
	if (info_out)
	{
		info_out[0] = eps;
		info_out[1] = eps_counter;
		info_out[2] = err_best;
		info_out[3] = 0;
	}
out:
	free(r0_); free(rk); free(rk_explicit); free(pk); free(z); free(h); free(s); free(v); free(buf); free(K); free(xk); free(x_best);
	return k;
}
