// Placement of the vectors beside the matrix in HBM.
//
// Measured on MI355X (profiles/r02_placement.md): the 288 GiB of a device behave as 32 GiB blocks that fall into classes, and the
// SAME SpMV kernel on the SAME matrix and x takes 1.25-1.29 ms or 1.46 ms (nlpkkt240 twin) depending only on whether y sits in
// a block of the same class as the value array (y is 2.6 % of the traffic: ~100 ns per written line, a DRAM row conflict per
// write-back); x and the index arrays move the time by 1-5 % the same way. Offsets inside a block do not matter. Which physical
// memory an allocation gets is the driver's choice and differs from process to process; a process that allocates its handle and
// then its vectors gets them side by side, usually in one block — the "slow timing state" round 1 could not explain.
// Synthetic write probes rank the blocks differently from the SpMV kernel, so vectors are placed by timing the handle's OWN
// kernel on candidates taken from deeper and deeper in the pool: earlier candidates and 16 GiB of ballast per step stay allocated
// during the walk so that the driver has to move on, and everything but the winner is returned at the end.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <vector>

#include "handle.hpp"

namespace spmv {

namespace {

constexpr size_t PLACE_MIN_BYTES = (size_t) 8 << 20;       // smaller problems run out of the caches
constexpr size_t BALLAST_STEP = (size_t) 16 << 30;         // blocks are 32 GiB: two candidates per block
constexpr size_t WALK_LIMIT = (size_t) 160 << 30;
constexpr size_t KEEP_FREE = (size_t) 8 << 30;             // never take the last of the pool for ballast
constexpr double CONTRAST = 1.04;                          // classes differ by 12-15 % (the two fast ones by 3 %), repeats by < 0.3 %

int
setting()
{
	static const int s = getenv("SPMV_MI355X_PLACEMENT") ? atoi(getenv("SPMV_MI355X_PLACEMENT")) : 1;
	return s;
}

// average microseconds of the handle's kernel reading x and writing y
double
kernel_us(spmv_mi355x_matrix * A, const void * x, void * y)
{
	double ms = 0;
	if (spmv_mi355x_time_device(A, x, y, 2, A->stream, &ms))         // warm-up: first touch of the candidate
		return -1.0;
	if (spmv_mi355x_time_device(A, x, y, 4, A->stream, &ms))
		return -1.0;
	return ms * 1e3;
}

struct Walk {
	double t_first = 0, t_chosen = 0, seconds = 0;
	int tries = 1;
	bool known = false;
};

// `first` (already allocated, `bytes` long) or a better-placed replacement of it in *chosen; measure(candidate) = kernel
// microseconds with the candidate in use. The caller moves contents and frees `first` when it lost.
template <typename Measure>
int
walk(spmv_mi355x_matrix * A, void * first, size_t bytes, Measure measure, void ** chosen, Walk & w)
{
	const auto c0 = std::chrono::steady_clock::now();
	*chosen = first;
	w.t_first = w.t_chosen = measure(first);
	if (w.t_first < 0)
		return 1;
	if (w.t_first < 20.0)                  // an (almost) empty handle: nothing a placement could change, nothing to measure it with
	{
		w.known = true;
		return 0;
	}
	std::vector<void *> held;              // ballast and rejected candidates, returned at the end
	w.known = A->place_fast_us > 0 && w.t_first <= A->place_fast_us * CONTRAST;
	if (!w.known && A->place_only_us > 0 && A->place_fast_us == 0)
	{
		if (w.t_first * CONTRAST < A->place_only_us)       // faster than everything a whole walk saw
		{
			A->place_fast_us = w.t_first;
			w.known = true;
		}
		else if (w.t_first <= A->place_only_us * CONTRAST)
			w.known = true;
	}
	if (!w.known)
	{
		bool contrast = false;
		for (size_t walked = 0; walked < WALK_LIMIT; walked += BALLAST_STEP)
		{
			size_t free_b = 0, total_b = 0;
			if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < BALLAST_STEP + bytes + KEEP_FREE)
				break;
			void * ballast = nullptr, * cand = nullptr;
			if (hipMalloc(&ballast, BALLAST_STEP) != hipSuccess)
			{
				(void) hipGetLastError();
				break;
			}
			held.push_back(ballast);
			if (hipMalloc(&cand, bytes) != hipSuccess)
			{
				(void) hipGetLastError();
				break;
			}
			(void) hipMemset(cand, 0, bytes);
			(void) hipDeviceSynchronize();
			const double t = measure(cand);
			w.tries++;
			if (t > 0 && t * CONTRAST < w.t_chosen)
			{
				if (*chosen != first)
					held.push_back(*chosen);
				*chosen = cand;                  // what was held so far shares a block class with the value stream
				w.t_chosen = t;
				A->place_fast_us = t;
				contrast = true;
				break;
			}
			held.push_back(cand);
			if (t > w.t_chosen * CONTRAST)          // what is held so far is well placed
			{
				A->place_fast_us = w.t_chosen;
				contrast = true;
				break;
			}
		}
		if (!contrast && A->place_fast_us == 0)
			A->place_only_us = w.t_first;
	}
	for (void * p : held)
		(void) hipFree(p);
	w.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
	return 0;
}

void
report(const char * what, size_t bytes, const Walk & w)
{
	if (setting() >= 2)
		fprintf(stderr, "[spmv_mi355x] placed %s (%.0f MiB): %.1f us per SpMV on the first candidate, %.1f us on the chosen one, %d candidate(s)%s, %.0f ms\n",
				what, (double) bytes / (1 << 20), w.t_first, w.t_chosen, w.tries, w.known ? " (rate already known)" : "", w.seconds * 1e3);
}

// Diagnostic (SPMV_MI355X_PLACEMENT=4): the kernel time with the value array at arena + a*16 GiB and y at arena + (b*16 + 8) GiB
// of one 160 GiB allocation, a, b = 0..9 — the table of profiles/r02_placement.md §4
int
placement_map(spmv_mi355x_matrix * A)
{
	size_t vsize = 0;
	if (!A->d_val || hipMemPtrGetInfo(A->d_val, &vsize) != hipSuccess || vsize > ((size_t) 8 << 30))
	{
		(void) hipGetLastError();
		return 0;
	}
	const size_t G = (size_t) 1 << 30;
	const int NA = 10;
	void * arena = nullptr;
	HIP_TRY(hipMalloc(&arena, NA * 16 * G));
	void * val0 = A->d_val;
	fprintf(stderr, "[spmv_mi355x] rows: value array (%.1f GiB) at a*16 GiB; columns: y at b*16+8 GiB; kernel us\n", (double) vsize / G);
	for (int a = 0; a < NA; a++)
	{
		void * v = (char *) arena + (size_t) a * 16 * G;
		HIP_TRY(hipMemcpy(v, val0, vsize, hipMemcpyDeviceToDevice));
		HIP_TRY(hipDeviceSynchronize());
		A->d_val = v;
		fprintf(stderr, "[spmv_mi355x] a=%d:", a);
		for (int b = 0; b < NA; b++)
			fprintf(stderr, " %6.0f", kernel_us(A, A->d_x, (char *) arena + ((size_t) b * 16 + 8) * G));
		fprintf(stderr, "\n");
	}
	A->d_val = val0;
	HIP_TRY(hipFree(arena));
	return 0;
}

}   // namespace

// Coordinate descent over WHERE the handle's arrays live: every array of 16 MiB .. 8 GiB (the value stream first, then y, x, index
// bytes, row permutation, ...; larger arrays stay and the others are placed against them) is tried at up to ten sites taken 16 GiB apart
// (a D2D copy and six launches per trial) and stays at the site where the handle's kernel ran fastest, if that beats where it
// was by 2 %. Sites that end up unused, the ballast between them and the originals of moved arrays are returned.
int
tune_placement(spmv_mi355x_matrix * A)
{
	if (setting() == 0 || A->placement_off || !A->d_x || !A->d_y || (size_t) (A->m + 64) * A->vbytes < PLACE_MIN_BYTES)
		return 0;
	const auto c0 = std::chrono::steady_clock::now();
	struct Slot { void ** p; const char * name; size_t size, off; };
	// every device array a kernel READS that can reach 16 MiB, plus the handle's own vectors (keep in step with handle.hpp; an array
	// missing here simply stays where it is)
	Slot all[] = {{&A->d_y, "y", 0, 0}, {&A->d_x, "x", 0, 0}, {&A->d_val, "val", 0, 0}, {(void **) &A->d_sell_idx, "sell_idx", 0, 0},
	              {(void **) &A->d_col, "col", 0, 0}, {(void **) &A->d_row_of_sorted, "row_of_sorted", 0, 0}, {(void **) &A->d_coob_ent, "coob_ent", 0, 0},
	              {(void **) &A->d_row_ptr, "row_ptr", 0, 0}, {(void **) &A->d_col16, "col16", 0, 0}, {(void **) &A->d_sell_desc, "sell_desc", 0, 0},
	              {(void **) &A->d_slice_ptr, "slice_ptr", 0, 0}, {(void **) &A->d_rowind, "rowind", 0, 0}};
	size_t largest = 0;
	for (Slot & sl : all)
		if (*sl.p)
		{
			if (hipMemPtrGetInfo(*sl.p, &sl.size) != hipSuccess)
			{
				(void) hipGetLastError();
				sl.size = 0;
			}
			largest = std::max(largest, sl.size);
		}
	std::vector<Slot *> movable;
	size_t site_bytes = 0;
	bool anchored = false;
	// arrays above the cap stay where they are (the others are placed against them). 8 GiB: the headline's 5.7 GiB value array takes part
	// (worth 1-2 %: 1.264-1.270 against 1.280-1.294 ms); the sites and the ballast between them still add up to ~160 GiB
	static const size_t cap = getenv("SPMV_MI355X_PLACEMENT_CAP_GIB") ? (size_t) atol(getenv("SPMV_MI355X_PLACEMENT_CAP_GIB")) << 30 : (size_t) 8 << 30;
	for (Slot & sl : all)
	{
		if (sl.size == largest && !anchored && sl.p != &A->d_y && sl.p != &A->d_x && sl.size > cap)
		{
			anchored = true;                        // the big stream everything else is placed against
			continue;
		}
		if (sl.size >= ((size_t) 16 << 20) && sl.size <= cap)
		{
			sl.off = site_bytes;
			site_bytes += (sl.size + ((size_t) 2 << 20) - 1) / ((size_t) 2 << 20) * ((size_t) 2 << 20);
			movable.push_back(&sl);
		}
	}
	if (movable.empty())
		return 0;
	std::stable_sort(movable.begin(), movable.end(), [&](const Slot * a, const Slot * b) { return (a->size == largest) > (b->size == largest); });
	const size_t ballast_bytes = site_bytes + ((size_t) 1 << 30) < BALLAST_STEP ? BALLAST_STEP - site_bytes : (size_t) 1 << 30;
	// a site = one buffer per movable array, allocated back to back behind its ballast (so they share a block of HBM); every buffer
	// is an allocation of its own: what an array does not move into is returned one by one at the end
	std::vector<void *> ballast;
	std::vector<std::vector<void *>> sites;
	for (int s = 0; s < 10; s++)
	{
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < ballast_bytes + site_bytes + KEEP_FREE)
			break;
		void * b = nullptr;
		if (hipMalloc(&b, ballast_bytes) != hipSuccess)
		{
			(void) hipGetLastError();
			break;
		}
		ballast.push_back(b);
		std::vector<void *> bufs(movable.size(), nullptr);
		bool ok = true;
		for (size_t k = 0; k < movable.size() && ok; k++)
			if (hipMalloc(&bufs[k], movable[k]->size) != hipSuccess)
			{
				(void) hipGetLastError();
				ok = false;
			}
		if (!ok)
		{
			for (void * q : bufs)
				if (q)
					(void) hipFree(q);
			break;
		}
		sites.push_back(bufs);
	}
	const double ms_alloc = std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count() * 1e3;
	std::vector<void *> original(movable.size());
	for (size_t k = 0; k < movable.size(); k++)
		original[k] = *movable[k]->p;
	double t_cur = kernel_us(A, A->d_x, A->d_y);
	const double t_start = t_cur;
	int rc = t_cur < 0;
	const std::vector<std::vector<void *>> all_sites = sites;          // for the clean-up at the end
	if (t_cur >= 0 && t_cur < 20.0)
		sites.clear();                     // launch-bound: differences between sites drown in the noise
	// Two sweeps: what is best for y depends on where x and the index arrays end up (and the other way round)
	for (int sweep = 0, moved_any = 1; sweep < 2 && moved_any && !rc && !sites.empty(); sweep++)
	{
		moved_any = 0;
		for (size_t k = 0; k < movable.size() && !rc; k++)
		{
			Slot * sl = movable[k];
			void * const orig = *sl->p;
			int best = -1;
			double t_best = t_cur;
			if (setting() >= 2)
				fprintf(stderr, "[spmv_mi355x] %-14s %5.0f MiB: %7.1f us where it is; at the sites:", sl->name, (double) sl->size / (1 << 20), t_cur);
			for (size_t s = 0; s < sites.size() && !rc; s++)
			{
				void * dst = sites[s][k];
				if (dst == orig)                          // second sweep: the site it already lives at
				{
					if (setting() >= 2)
						fprintf(stderr, " =");
					continue;
				}
				// on the stream the trial launches use, and finished before they start: a kernel that read a half-copied index
				// array would gather x out of bounds
				if (hipMemcpyAsync(dst, orig, sl->size, hipMemcpyDeviceToDevice, A->stream) != hipSuccess || hipStreamSynchronize(A->stream) != hipSuccess)
				{
					set_error("placement: device copy failed: %s", hipGetErrorString(hipGetLastError()));
					rc = 1;
					break;
				}
				*sl->p = dst;
				const double t = kernel_us(A, A->d_x, A->d_y);
				if (t < 0)
					rc = 1;
				if (setting() >= 2)
					fprintf(stderr, " %.0f", t);
				if (t > 0 && t < t_best * 0.98)
				{
					best = (int) s;
					t_best = t;
				}
			}
			if (best >= 0 && !rc)
			{
				// every site holds a faithful copy: the arrays the kernel reads never change, y is zeroed below
				*sl->p = sites[(size_t) best][k];
				t_cur = t_best;
				moved_any = 1;
			}
			else
				*sl->p = orig;
			if (setting() >= 2)
				fprintf(stderr, " -> %s, %.1f us\n", best >= 0 ? "moved" : "stays", t_cur);
		}
	}
	if (!rc)
		HIP_TRY(hipMemset(A->d_y, 0, (size_t) (A->m + 64) * A->vbytes));
	HIP_TRY(hipDeviceSynchronize());
	for (void * p : ballast)
		(void) hipFree(p);
	// every site buffer an array did not end up in goes back, and so does the original of an array that moved: each array is again
	// an allocation of its own, owned through the handle's pointer as before
	for (size_t k = 0; k < movable.size(); k++)
	{
		void * const now = *movable[k]->p;
		for (const auto & bufs : all_sites)
			if (bufs[k] != now)
				(void) hipFree(bufs[k]);
		if (original[k] != now)
			(void) hipFree(original[k]);
	}
	A->place_fast_us = t_cur;                      // later output vectors are held against this
	if (setting() >= 2)
		fprintf(stderr, "[spmv_mi355x] placement: %.1f -> %.1f us per SpMV, %zu sites, %.0f ms (%.0f ms of it allocating the sites)\n", t_start, t_cur,
				all_sites.size(), std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count() * 1e3, ms_alloc);
	if (!rc && setting() == 4)
		return placement_map(A);
	return rc;
}

// allocation of `bytes` for a vector the handle's SpMV writes (A->d_x must exist). Zero-filled.
int
dev_alloc_output(spmv_mi355x_matrix * A, void ** out, size_t bytes)
{
	if (dev_alloc_bytes(out, bytes))
		return 1;
	HIP_TRY(hipMemset(*out, 0, std::max<size_t>(bytes, 8)));
	HIP_TRY(hipDeviceSynchronize());
	const size_t need = (size_t) (A->m + 64) * A->vbytes;
	if (setting() == 0 || A->placement_off || bytes < PLACE_MIN_BYTES || bytes < need || !A->d_x)
		return 0;
	void * first = *out, * chosen = nullptr;
	Walk w;
	if (walk(A, first, bytes, [&](void * c) { return kernel_us(A, A->d_x, c); }, &chosen, w))
		return 1;
	if (chosen != first)
		HIP_TRY(hipFree(first));
	HIP_TRY(hipMemset(chosen, 0, bytes));
	HIP_TRY(hipDeviceSynchronize());
	*out = chosen;
	report("an output vector", bytes, w);
	return 0;
}

}   // namespace spmv

extern "C" {

int
spmv_mi355x_output_alloc(spmv_mi355x_matrix * A, size_t bytes, void ** out)
{
	if (!A || !out)
	{
		spmv::set_error("output_alloc: NULL argument");
		return 1;
	}
	*out = nullptr;
	if (spmv::ensure_x(A))                   // sets the device, creates the handle's stream and its own x
		return 1;
	return spmv::dev_alloc_output(A, out, bytes);
}

int
spmv_mi355x_output_free(void * p)
{
	if (p)
		HIP_TRY(hipFree(p));
	return 0;
}

}
