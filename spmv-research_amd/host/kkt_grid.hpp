// The analytic KKT matrix [H A^T; A 0] over an N^3 grid that stands in for nlpkkt240 (synthetic.cpp): grid geometry, the columns of
// any row on the fly. Shared by the generators and by the matrix-free graph partition (graph_partition.cpp), which walks
// the 28 M-vertex graph without ever holding the 9 GB matrix.
#pragma once

#include <string.h>
#include <algorithm>

#include "host.hpp"

namespace spmv_host {

struct Grid {
	long N, N2, n1, n2, S;
	inline int stencil7(long g, long * out) const      // in-bounds 7-point neighbours of g, ascending
	{
		long z = g / N2, y = (g / N) % N, x = g % N;
		int k = 0;
		if (z > 0) out[k++] = g - N2;
		if (y > 0) out[k++] = g - N;
		if (x > 0) out[k++] = g - 1;
		out[k++] = g;
		if (x < N - 1) out[k++] = g + 1;
		if (y < N - 1) out[k++] = g + N;
		if (z < N - 1) out[k++] = g + N2;
		return k;
	}
	inline int stencil27(long g, long * out) const     // ascending
	{
		long z = g / N2, y = (g / N) % N, x = g % N;
		int k = 0;
		for (long dz = -1; dz <= 1; dz++)
			for (long dy = -1; dy <= 1; dy++)
				for (long dx = -1; dx <= 1; dx++)
				{
					long zz = z + dz, yy = y + dy, xx = x + dx;
					if (zz < 0 || zz >= N || yy < 0 || yy >= N || xx < 0 || xx >= N)
						continue;
					out[k++] = zz * N2 + yy * N + xx;
				}
		return k;
	}
	inline long gk(long k) const { return k % n1; }                    // grid point of constraint row k
	inline long w(long g) const { return (g + S) % n1; }               // half-domain shift
	inline long winv(long g) const { return (g - S % n1 + n1) % n1; }
	// columns of constraint row k (grid indices), ascending
	inline int a_row(long k, long * out) const
	{
		long t[14];
		int c = stencil7(gk(k), t);
		c += stencil7(w(gk(k)), t + c);
		std::sort(t, t + c);
		c = (int) (std::unique(t, t + c) - t);
		memcpy(out, t, c * sizeof(long));
		return c;
	}
	// constraint rows k whose A-row touches grid point g, ascending
	inline int a_col(long g, long * out) const
	{
		long h[14], t[28];
		int c = stencil7(g, h);
		long s7[7];
		int c2 = stencil7(g, s7);
		for (int q = 0; q < c2; q++)
			h[c++] = winv(s7[q]);
		// h: grid points p with g in stencil7(p) (symmetric) or g in stencil7(w(p))
		int k = 0;
		for (int q = 0; q < c; q++)
		{
			t[k++] = h[q];
			if (h[q] + n1 < n2)
				t[k++] = h[q] + n1;
		}
		std::sort(t, t + k);
		k = (int) (std::unique(t, t + k) - t);
		memcpy(out, t, k * sizeof(long));
		return k;
	}
};

inline int
make_grid(long N, Grid & G)
{
	if (N < 4)
	{
		set_error("KKT grid edge must be >= 4");
		return 1;
	}
	G.N = N; G.N2 = N * N; G.n1 = N * N * N; G.n2 = G.n1 + 6 * N * N; G.S = G.n1 / 2 + N / 3;
	return 0;
}

inline int
kkt_row_len(const Grid & G, long i)
{
	long tmp[32];
	if (i < G.n1)
		return G.stencil27(i, tmp) + G.a_col(i, tmp);
	return G.a_row(i - G.n1, tmp);
}


// columns of row i, ascending (at most 55 of them); returns their number
inline int
kkt_row_cols(const Grid & G, long i, int32_t * out)
{
	long tmp[32];
	int k = 0;
	if (i < G.n1)
	{
		int c = G.stencil27(i, tmp);
		for (int q = 0; q < c; q++, k++)
			out[k] = (int32_t) tmp[q];
		c = G.a_col(i, tmp);
		for (int q = 0; q < c; q++, k++)
			out[k] = (int32_t) (G.n1 + tmp[q]);
	}
	else
	{
		int c = G.a_row(i - G.n1, tmp);
		for (int q = 0; q < c; q++, k++)
			out[k] = (int32_t) tmp[q];
	}
	return k;
}

}  // namespace spmv_host
