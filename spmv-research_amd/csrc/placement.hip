// Placement of the vectors beside the matrix in HBM: two to four process-wide VECTOR POOLS per device, in blocks of different class.
//
// Measured on MI355X (profiles/r02_placement.md): the 288 GiB of a device behave as 32 GiB blocks that fall into classes, and the
// SAME SpMV kernel on the SAME matrix and x takes 1.25-1.29 ms or 1.46 ms (nlpkkt240 twin) depending only on whether y sits in
// a block of the same class as the value array (y is 2.6 % of the traffic: ~100 ns per written line, a DRAM row conflict per
// write-back); x moves the time by 1-5 % the same way. Offsets inside a block do not matter. Which physical memory an allocation
// gets is the driver's choice; a process that allocates its handle and then its vectors gets them side by side, usually in one
// block — the "slow timing state" round 1 could not explain. Synthetic write probes rank the blocks differently from the SpMV
// kernel, so classes are told apart by timing a handle's OWN kernel.
//
// Round 2 searched per handle (ten sites, ~165 GiB of ballast, ~600 launches, every handle again). Round 3: ONE walk per process and
// device, made by the first handle that asks (opts.placement = 1 or SPMV_MI355X_PLACEMENT >= 1; OFF by default): candidates for y
// are taken from deeper and deeper in the driver's pool, 16 GiB of ballast apart; a candidate under which the handle's kernel runs
// 1.5 % faster or slower than under every candidate kept so far is in a block of another class and is KEPT as one more of the device's
// vector pools (1-4 GiB each, at most four). The walk ends three candidates after the last new class once it has two (boxes differ:
// the second class turned up 16 GiB in on some, 128 GiB in on others — profiles/r03_placement_walk.txt — and some show a third), or
// at opts.placement_budget_gib of ballast (default 160, never the last 8 GiB of the device); the ballast goes back. From then on a
// vector of any handle of the process is a slice of one of the pools: one trial per pool (6 launches each) decides which. A scope
// guard returns ballast and rejected candidates on every way out. Other processes on the same GPU: the walk's free-memory
// check (hipMemGetInfo, then hipMalloc) is racy between processes, and a neighbour's kernels run a few % slower for the seconds
// the driver takes to clear the returned ballast — one more reason why this is opt-in.
// Level 3 (SPMV_MI355X_PLACEMENT=3 / opts.placement = 3; spmv_mi355x_place_arrays for a caller's vectors) adds round 2's search over the handle's
// matrix arrays — worth 1-5 %: most where the driver laid the value array across two regions, so that no pool is good for y — and places the vectors again
// when an array moved.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "handle.hpp"

namespace spmv {

namespace {

constexpr size_t PLACE_MIN_BYTES = (size_t) 8 << 20;       // smaller problems run out of the caches
constexpr size_t WALK_STEP = (size_t) 16 << 30;            // blocks are 32 GiB: two candidates per block
constexpr size_t KEEP_FREE = (size_t) 8 << 30;             // never take the last of the device for ballast
constexpr size_t POOL_ALIGN = (size_t) 2 << 20;
constexpr double CONTRAST = 1.015;                         // classes differ by 12-15 % (the two fast ones by 2-3 %), repeats by < 0.3 %
constexpr int MAX_DEV = 64;
constexpr int MAX_POOLS = 4;
constexpr int WALK_ON = 3;                                 // candidates tried after the last new class before the walk ends

// SPMV_MI355X_PLACEMENT: unset = what the handle's opts say; 0 = off; 1 = pools; 2 = pools + log; 3 = + search over the matrix arrays;
// 4 = 3 + the (value array block) x (y block) table of profiles/r02_placement.md §4
int
env_level()
{
	static const int s = getenv("SPMV_MI355X_PLACEMENT") ? atoi(getenv("SPMV_MI355X_PLACEMENT")) : -1;
	return s;
}

int
level_of(const spmv_mi355x_matrix * A)
{
	const int e = env_level();
	return e >= 0 ? e : A->placement_level;
}

bool
verbose()
{
	return env_level() >= 2;
}

// returns everything it still holds when it goes out of scope
struct Held {
	std::vector<void *> v;
	~Held() { release(); }
	void release()
	{
		for (void * p : v)
			if (p)
				(void) hipFree(p);
		v.clear();
	}
};

struct Pool {
	char * base = nullptr;
	size_t size = 0;
	std::vector<std::pair<size_t, size_t>> free_list;      // (offset, bytes), sorted by offset, coalesced
	std::vector<std::pair<size_t, size_t>> live;           // (offset, bytes) handed out
};

struct DevPools {
	int state = 0;                 // 0 = no walk yet, 1 = pools of different class, 2 = walked: no contrast seen (plain allocations)
	int npools = 0;
	Pool pool[MAX_POOLS];
	double walk_us[MAX_POOLS] = {0, 0, 0, 0};   // the walking handle's kernel with y in pool k
	long walked_gib = 0;           // ballast the walk held at its deepest
	int candidates = 0;
};

DevPools g_dev[MAX_DEV];
std::mutex g_mu;

void *
pool_alloc(Pool & p, size_t bytes)
{
	bytes = (bytes + POOL_ALIGN - 1) / POOL_ALIGN * POOL_ALIGN;
	for (size_t i = 0; i < p.free_list.size(); i++)
		if (p.free_list[i].second >= bytes)
		{
			const size_t off = p.free_list[i].first;
			p.free_list[i].first += bytes;
			p.free_list[i].second -= bytes;
			if (p.free_list[i].second == 0)
				p.free_list.erase(p.free_list.begin() + (long) i);
			p.live.emplace_back(off, bytes);
			return p.base + off;
		}
	return nullptr;
}

bool
pool_free(Pool & p, void * ptr)
{
	if (!p.base || (char *) ptr < p.base || (char *) ptr >= p.base + p.size)
		return false;
	const size_t off = (size_t) ((char *) ptr - p.base);
	for (size_t i = 0; i < p.live.size(); i++)
		if (p.live[i].first == off)
		{
			const std::pair<size_t, size_t> blk = p.live[i];
			p.live.erase(p.live.begin() + (long) i);
			auto it = std::lower_bound(p.free_list.begin(), p.free_list.end(), blk);
			it = p.free_list.insert(it, blk);
			if (it + 1 != p.free_list.end() && it->first + it->second == (it + 1)->first)
			{
				it->second += (it + 1)->second;
				p.free_list.erase(it + 1);
			}
			if (it != p.free_list.begin() && (it - 1)->first + (it - 1)->second == it->first)
			{
				(it - 1)->second += it->second;
				p.free_list.erase(it);
			}
			return true;
		}
	return true;       // inside the pool but not a live block: nothing to do (a double free)
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

size_t
budget_bytes(const spmv_mi355x_matrix * A)
{
	long gib = A->placement_budget_gib > 0 ? A->placement_budget_gib : 160;
	if (const char * e = getenv("SPMV_MI355X_PLACEMENT_BUDGET_GIB"))
		if (atol(e) > 0)
			gib = atol(e);
	return (size_t) gib << 30;
}

// The one walk of a device: find two candidates of different block class under A's kernel (y = candidate), keep them as the pools.
int
build_pools(spmv_mi355x_matrix * A, DevPools & dp)
{
	const auto c0 = std::chrono::steady_clock::now();
	dp.state = 2;
	const size_t vec = std::max((size_t) (A->m + 64), (size_t) std::max<long>(A->n, 1)) * A->vbytes;
	const size_t pool_bytes = std::min<size_t>(std::max<size_t>((6 * vec + POOL_ALIGN - 1) / POOL_ALIGN * POOL_ALIGN, (size_t) 1 << 30), (size_t) 4 << 30);
	const size_t budget = budget_bytes(A);
	Held held;
	void * cand0 = nullptr;
	if (hipMalloc(&cand0, pool_bytes) != hipSuccess)
	{
		(void) hipGetLastError();
		return 0;                                  // no room for pools: plain allocations
	}
	held.v.push_back(cand0);
	HIP_TRY(hipMemsetAsync(cand0, 0, pool_bytes, A->stream));
	HIP_TRY(hipStreamSynchronize(A->stream));
	const double t0 = kernel_us(A, A->d_x, cand0);
	if (t0 < 0)
		return 1;
	if (t0 < 20.0)                                 // launch-bound: differences between blocks drown in the noise
		return 0;
	void * kept[MAX_POOLS] = {cand0, nullptr, nullptr, nullptr};
	double t_kept[MAX_POOLS] = {t0, 0, 0, 0};
	int nk = 1, tries = 1, since_new = 0;
	for (size_t walked = 0; walked + WALK_STEP <= budget && nk < MAX_POOLS && !(nk >= 2 && since_new >= WALK_ON); walked += WALK_STEP)
	{
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < WALK_STEP + pool_bytes + KEEP_FREE)
			break;
		void * ballast = nullptr, * cand = nullptr;
		if (hipMalloc(&ballast, WALK_STEP) != hipSuccess)
		{
			(void) hipGetLastError();
			break;
		}
		held.v.push_back(ballast);
		if (hipMalloc(&cand, pool_bytes) != hipSuccess)
		{
			(void) hipGetLastError();
			break;
		}
		held.v.push_back(cand);
		HIP_TRY(hipMemsetAsync(cand, 0, pool_bytes, A->stream));
		HIP_TRY(hipStreamSynchronize(A->stream));
		const double t = kernel_us(A, A->d_x, cand);
		tries++;
		if (t < 0)
			return 1;
		bool fresh = true;
		for (int k = 0; k < nk; k++)
			fresh = fresh && (t * CONTRAST < t_kept[k] || t > t_kept[k] * CONTRAST);
		if (verbose())
			fprintf(stderr, "[spmv_mi355x] placement walk: %.0f GiB in, %.1f us per SpMV%s\n", (double) (walked + WALK_STEP) / (1 << 30), t, fresh ? " (a new class: kept)" : "");
		since_new++;
		if (fresh)
		{
			kept[nk] = cand;
			t_kept[nk++] = t;
			since_new = 0;
		}
	}
	dp.candidates = tries;
	dp.walked_gib = (long) ((size_t) (tries - 1) * (WALK_STEP >> 30));
	if (nk >= 2)
	{
		// keep them: take them out of the guard's hands
		for (void *& p : held.v)
			for (int k = 0; k < nk; k++)
				if (p && p == kept[k])
					p = nullptr;
		dp.npools = nk;
		for (int k = 0; k < nk; k++)
		{
			Pool & p = dp.pool[k];
			p.base = (char *) kept[k];
			p.size = pool_bytes;
			p.free_list.assign(1, std::make_pair((size_t) 0, pool_bytes));
			p.live.clear();
			dp.walk_us[k] = t_kept[k];
		}
		dp.state = 1;
	}
	else
		dp.walk_us[0] = t0;
	held.release();
	if (verbose())
	{
		fprintf(stderr, "[spmv_mi355x] placement: device %d, %d candidate(s), %s, pools of %.0f MiB, %.0f ms\n", A->device, tries,
				nk >= 2 ? "block classes found" : "no contrast inside the budget: plain allocations", (double) pool_bytes / (1 << 20),
				std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count() * 1e3);
		if (nk >= 2)
		{
			fprintf(stderr, "[spmv_mi355x] placement: us per SpMV of %s with y in pool 0..%d:", A->format_name, nk - 1);
			for (int k = 0; k < nk; k++)
				fprintf(stderr, " %.1f", t_kept[k]);
			fprintf(stderr, "\n");
		}
	}
	return 0;
}

int
plain_alloc(void ** out, size_t bytes)
{
	if (dev_alloc_bytes(out, bytes))
		return 1;
	HIP_TRY(hipMemset(*out, 0, std::max<size_t>(bytes, 8)));
	HIP_TRY(hipDeviceSynchronize());
	return 0;
}

// Diagnostic (level 4): the kernel time with the value array at arena + a*16 GiB and y at arena + (b*16 + 8) GiB of one 160 GiB
// allocation, a, b = 0..9 — the table of profiles/r02_placement.md §4
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
	Held held;
	void * arena = nullptr;
	HIP_TRY(hipMalloc(&arena, NA * 16 * G));
	held.v.push_back(arena);
	void * val0 = A->d_val;
	fprintf(stderr, "[spmv_mi355x] rows: value array (%.1f GiB) at a*16 GiB; columns: y at b*16+8 GiB; kernel us\n", (double) vsize / G);
	for (int a = 0; a < NA; a++)
	{
		void * v = (char *) arena + (size_t) a * 16 * G;
		if (hipMemcpy(v, val0, vsize, hipMemcpyDeviceToDevice) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
		{
			A->d_val = val0;
			set_error("placement map: device copy failed");
			return 1;
		}
		A->d_val = v;
		fprintf(stderr, "[spmv_mi355x] a=%d:", a);
		for (int b = 0; b < NA; b++)
			fprintf(stderr, " %6.0f", kernel_us(A, A->d_x, (char *) arena + ((size_t) b * 16 + 8) * G));
		fprintf(stderr, "\n");
	}
	A->d_val = val0;
	return 0;
}

// Level 3: coordinate descent over where the handle's MATRIX arrays live (round 2's search, minus the vectors, which the pools place):
// every array of 16 MiB .. 8 GiB is tried at up to ten sites taken 16 GiB apart (a D2D copy and six launches per trial) and stays
// where the handle's kernel ran fastest, if that beats where it was by 2 %. Sites that end up unused, the ballast between them and the
// originals of moved arrays are returned — by the guard on every way out.
int
search_arrays(spmv_mi355x_matrix * A, const void * x, void * y, bool * moved_out)
{
	*moved_out = false;
	const auto c0 = std::chrono::steady_clock::now();
	struct Slot { void ** p; const char * name; size_t size; };
	Slot all[] = {{&A->d_val, "val", 0}, {(void **) &A->d_sell_idx, "sell_idx", 0}, {(void **) &A->d_col, "col", 0},
	              {(void **) &A->d_row_of_sorted, "row_of_sorted", 0}, {(void **) &A->d_coob_ent, "coob_ent", 0}, {(void **) &A->d_row_ptr, "row_ptr", 0},
	              {(void **) &A->d_col16, "col16", 0}, {(void **) &A->d_sell_desc, "sell_desc", 0}, {(void **) &A->d_slice_ptr, "slice_ptr", 0},
	              {(void **) &A->d_rowind, "rowind", 0}};
	const size_t cap = (size_t) 8 << 30;
	std::vector<Slot *> movable;
	size_t site_bytes = 0;
	for (Slot & sl : all)
		if (*sl.p)
		{
			if (hipMemPtrGetInfo(*sl.p, &sl.size) != hipSuccess)
			{
				(void) hipGetLastError();
				sl.size = 0;
			}
			if (sl.size >= ((size_t) 16 << 20) && sl.size <= cap)
			{
				site_bytes += (sl.size + POOL_ALIGN - 1) / POOL_ALIGN * POOL_ALIGN;
				movable.push_back(&sl);
			}
		}
	if (movable.empty())
		return 0;
	std::stable_sort(movable.begin(), movable.end(), [](const Slot * a, const Slot * b) { return a->size > b->size; });
	const size_t ballast_bytes = site_bytes + ((size_t) 1 << 30) < WALK_STEP ? WALK_STEP - site_bytes : (size_t) 1 << 30;
	const size_t budget = budget_bytes(A);
	Held ballast, rejected;
	std::vector<std::vector<void *>> sites;
	for (size_t s = 0, used = 0; s < 10 && used + ballast_bytes + site_bytes <= budget + WALK_STEP; s++, used += ballast_bytes + site_bytes)
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
		ballast.v.push_back(b);
		std::vector<void *> bufs(movable.size(), nullptr);
		bool ok = true;
		for (size_t k = 0; k < movable.size() && ok; k++)
			if (hipMalloc(&bufs[k], movable[k]->size) != hipSuccess)
			{
				(void) hipGetLastError();
				ok = false;
			}
		for (void * q : bufs)
			if (q)
				rejected.v.push_back(q);               // the guard owns every site buffer until an array moves in
		if (!ok)
			break;
		sites.push_back(bufs);
	}
	double t_cur = kernel_us(A, x, y);
	const double t_start = t_cur;
	if (t_cur < 0)
		return 1;
	if (t_cur < 20.0)
		return 0;
	for (int sweep = 0, moved_any = 1; sweep < 2 && moved_any && !sites.empty(); sweep++)
	{
		moved_any = 0;
		for (size_t k = 0; k < movable.size(); k++)
		{
			Slot * sl = movable[k];
			void * const orig = *sl->p;
			int best = -1;
			double t_best = t_cur;
			if (verbose())
				fprintf(stderr, "[spmv_mi355x] %-14s %5.0f MiB: %7.1f us where it is; at the sites:", sl->name, (double) sl->size / (1 << 20), t_cur);
			for (size_t s = 0; s < sites.size(); s++)
			{
				void * dst = sites[s][k];
				if (dst == orig)
					continue;
				// on the stream the trial launches use, and finished before they start: a kernel that read a half-copied index
				// array would gather x out of bounds
				if (hipMemcpyAsync(dst, orig, sl->size, hipMemcpyDeviceToDevice, A->stream) != hipSuccess || hipStreamSynchronize(A->stream) != hipSuccess)
				{
					*sl->p = orig;
					set_error("placement: device copy failed: %s", hipGetErrorString(hipGetLastError()));
					return 1;
				}
				*sl->p = dst;
				const double t = kernel_us(A, x, y);
				*sl->p = orig;
				if (t < 0)
					return 1;
				if (verbose())
					fprintf(stderr, " %.0f", t);
				if (t < t_best)
				{
					best = (int) s;
					t_best = t;
				}
			}
			if (best >= 0 && t_best >= t_cur * 0.98)
				best = -1;                                 // the fastest site is not worth a move
			if (best >= 0)
			{
				// the array moves in: the site buffer leaves the guard, the old home enters it
				void * const now = sites[(size_t) best][k];
				for (void *& q : rejected.v)
					if (q == now)
						q = orig;
				*sl->p = now;
				t_cur = t_best;
				moved_any = 1;
				*moved_out = true;
			}
			if (verbose())
				fprintf(stderr, " -> %s, %.1f us\n", best >= 0 ? "moved" : "stays", t_cur);
		}
	}
	HIP_TRY(hipDeviceSynchronize());
	if (verbose())
		fprintf(stderr, "[spmv_mi355x] array search: %.1f -> %.1f us per SpMV, %zu sites, %.0f ms\n", t_start, t_cur, sites.size(),
				std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count() * 1e3);
	return 0;
}

}   // namespace

// A zero-filled device vector of `bytes` for handle A: one its SpMV WRITES (is_output: timed as y against A's x) or one it READS (timed
// as x against A's y). With placement on and the vector large enough it is a slice of the device's pool in which A's kernel runs faster.
int
place_vector(spmv_mi355x_matrix * A, void ** out, size_t bytes, bool is_output)
{
	*out = nullptr;
	const size_t need = is_output ? (size_t) (A->m + 64) * A->vbytes : (size_t) std::max<long>(A->n, 1) * A->vbytes;
	const void * other = is_output ? (const void *) A->d_x : (const void *) A->d_y;
	if (level_of(A) < 1 || bytes < PLACE_MIN_BYTES || bytes < need || !other || A->device < 0 || A->device >= MAX_DEV)
		return plain_alloc(out, bytes);
	std::lock_guard<std::mutex> lock(g_mu);
	DevPools & dp = g_dev[A->device];
	if (dp.state == 0)
	{
		if (!is_output || !A->d_x)
			return plain_alloc(out, bytes);            // the walk is made with a y candidate
		if (build_pools(A, dp))
			return 1;
	}
	if (dp.state != 1)
		return plain_alloc(out, bytes);
	void * c[MAX_POOLS] = {nullptr, nullptr, nullptr, nullptr};
	bool room = true;
	for (int k = 0; k < dp.npools; k++)
		room = (c[k] = pool_alloc(dp.pool[k], bytes)) != nullptr && room;
	if (!room)
	{
		for (int k = 0; k < dp.npools; k++)
			if (c[k])
				pool_free(dp.pool[k], c[k]);
		return plain_alloc(out, bytes);                // pools exhausted
	}
	double t[MAX_POOLS] = {0, 0, 0, 0};
	int win = 0;
	for (int k = 0; k < dp.npools; k++)
	{
		HIP_TRY(hipMemsetAsync(c[k], 0, bytes, A->stream));
		HIP_TRY(hipStreamSynchronize(A->stream));
		t[k] = is_output ? kernel_us(A, A->d_x, c[k]) : kernel_us(A, c[k], A->d_y);
		if (t[k] < 0)
			return 1;
		if (t[k] < t[win])
			win = k;
	}
	for (int k = 0; k < dp.npools; k++)
		if (k != win)
			pool_free(dp.pool[k], c[k]);
	if (is_output)
	{
		HIP_TRY(hipMemsetAsync(c[win], 0, bytes, A->stream));      // the trials wrote into it
		HIP_TRY(hipStreamSynchronize(A->stream));
	}
	*out = c[win];
	if (verbose())
	{
		fprintf(stderr, "[spmv_mi355x] placed %s of %s (%.0f MiB): us per SpMV in pool 0..%d:", is_output ? "an output vector" : "an input vector", A->format_name,
				(double) bytes / (1 << 20), dp.npools - 1);
		for (int k = 0; k < dp.npools; k++)
			fprintf(stderr, " %.1f", t[k]);
		fprintf(stderr, " -> pool %d\n", win);
	}
	return 0;
}

// frees what place_vector (or a plain allocation) returned
int
vector_free(void * p)
{
	if (!p)
		return 0;
	{
		std::lock_guard<std::mutex> lock(g_mu);
		for (DevPools & dp : g_dev)
			if (dp.state == 1)
				for (Pool & pl : dp.pool)
					if (pool_free(pl, p))
						return 0;
	}
	HIP_TRY(hipFree(p));
	return 0;
}

// is p a slice of one of the device's pools?
static bool
in_pools(const spmv_mi355x_matrix * A, const void * p)
{
	std::lock_guard<std::mutex> lock(g_mu);
	const DevPools & dp = g_dev[std::min(std::max(A->device, 0), MAX_DEV - 1)];
	bool pooled = false;
	for (const Pool & pl : dp.pool)
		pooled = pooled || (dp.state == 1 && pl.base && (const char *) p >= pl.base && (const char *) p < pl.base + pl.size);
	return pooled;
}

// x (zero-filled, not yet written by the caller) into the pool where A's kernel reads it fastest; stays where it is when there are no pools
static int
place_own_x(spmv_mi355x_matrix * A)
{
	void * x2 = nullptr;
	const size_t xb = (size_t) std::max<long>(A->n, 1) * A->vbytes;
	if (place_vector(A, &x2, xb, false))
		return 1;
	// x2 is a pool slice only when the pools exist and x is large enough; otherwise it is one more plain allocation: keep the first
	if (in_pools(A, x2))
	{
		if (vector_free(A->d_x))
			return 1;
		A->d_x = x2;
		return 0;
	}
	return vector_free(x2);
}

// Once per handle, when its own x / y pair is first needed (A->d_x exists as a plain, zeroed allocation): y into the best pool, then x;
// level 3: then the matrix arrays, and when one of them moved, y and x once more (the best block for a vector is one of another class
// than the value array's: profiles/r03_placement_walk.txt).
int
tune_placement(spmv_mi355x_matrix * A)
{
	if (place_vector(A, &A->d_y, (size_t) (A->m + 64) * A->vbytes, true))
		return 1;
	if (level_of(A) < 1)
		return 0;
	if (place_own_x(A))
		return 1;
	if (level_of(A) >= 3)
	{
		bool moved = false;
		if (search_arrays(A, A->d_x, A->d_y, &moved))
			return 1;
		if (moved && in_pools(A, A->d_y))
		{
			void * y2 = nullptr;
			if (place_vector(A, &y2, (size_t) (A->m + 64) * A->vbytes, true) || vector_free(A->d_y))
				return 1;
			A->d_y = y2;
			if (place_own_x(A))
				return 1;
		}
	}
	if (level_of(A) >= 4)
		return placement_map(A);
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
	if (spmv::ensure_x(A))                   // sets the device, creates the handle's own x
		return 1;
	return spmv::place_vector(A, out, bytes, true);
}

int
spmv_mi355x_input_alloc(spmv_mi355x_matrix * A, size_t bytes, void ** out)
{
	if (!A || !out)
	{
		spmv::set_error("input_alloc: NULL argument");
		return 1;
	}
	*out = nullptr;
	if (!spmv_mi355x_y_device(A))            // the handle's own pair first: an input vector is timed against the handle's y
		return 1;
	return spmv::place_vector(A, out, bytes, false);
}

int
spmv_mi355x_place_arrays(spmv_mi355x_matrix * A, const void * x_dev, void * y_dev)
{
	if (!A || !x_dev || !y_dev)
	{
		spmv::set_error("place_arrays: NULL argument");
		return 1;
	}
	if (spmv::ensure_x(A))                   // sets the device, creates the stream the trials run on
		return 1;
	bool moved = false;
	return spmv::search_arrays(A, x_dev, y_dev, &moved);
}

int
spmv_mi355x_output_free(void * p)
{
	return spmv::vector_free(p);
}

int
spmv_mi355x_placement_info(int device, int * state_out, int * candidates_out, long * walked_gib_out, int * pools_out, double us_out[4])
{
	if (device < 0 || device >= spmv::MAX_DEV)
	{
		spmv::set_error("placement_info: device %d", device);
		return 1;
	}
	std::lock_guard<std::mutex> lock(spmv::g_mu);
	const spmv::DevPools & dp = spmv::g_dev[device];
	if (state_out) *state_out = dp.state;
	if (candidates_out) *candidates_out = dp.candidates;
	if (walked_gib_out) *walked_gib_out = dp.walked_gib;
	if (pools_out) *pools_out = dp.npools;
	if (us_out)
		for (int k = 0; k < spmv::MAX_POOLS; k++)
			us_out[k] = dp.walk_us[k];
	return 0;
}

int
spmv_mi355x_placement_release(int device)
{
	std::lock_guard<std::mutex> lock(spmv::g_mu);
	for (int d = 0; d < spmv::MAX_DEV; d++)
	{
		if (device >= 0 && d != device)
			continue;
		spmv::DevPools & dp = spmv::g_dev[d];
		if (dp.state != 1)
		{
			dp.state = 0;
			continue;
		}
		size_t live = 0;
		for (const spmv::Pool & pl : dp.pool)
			live += pl.live.size();
		if (live)
		{
			spmv::set_error("placement_release: device %d still has %zu vector(s) in its pools", d, live);
			return 1;
		}
		int cur = -1;
		(void) hipGetDevice(&cur);
		(void) hipSetDevice(d);
		for (spmv::Pool & pl : dp.pool)
		{
			(void) hipFree(pl.base);
			pl = spmv::Pool();
		}
		if (cur >= 0)
			(void) hipSetDevice(cur);
		dp.npools = 0;
		dp.state = 0;
	}
	return 0;
}

}
