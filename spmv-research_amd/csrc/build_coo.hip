// csr_to_format() for COO: the reference's row-sorted expansion (mkl_coo.cpp:79-90: rowind[j] = i for j in row i) done on the
// GPU, and the column-blocked layout for graph matrices (kernels_coo.hip).
#include "handle.hpp"

namespace spmv {

// Column-blocked layout (opts.col_blocks; described in kernels_coo.hip). Built on the host: row ranges balanced by non-zeros
// (COO: equal shares of the entries, as coo_kernel's waves) or by rows + non-zeros (merge path: equal shares of the merge
// diagonal, merge.cpp:277-287), cut at chunk boundaries — by rows alone when a share would overflow a workgroup's LDS;
// chunks of 16 rows dealt to the 32 workgroups of a range so that all get the same number of entries; rows longer than a
// workgroup's fair share / 8 are split over the 32 workgroups; per workgroup a sort by (column, LDS slot) and a cut into full
// batches whose wave instructions have a base column each.
int
build_blocked_layout(spmv_mi355x_matrix * A, const int * rp, const int * ci, const double * va, int col_blocks, bool merge_balance)
{
	const long lm = A->m, lnnz = A->nnz, n = A->n;
	const char * pf = A->f32 ? "f" : "d";
	const int WGS = coo_blocked_wgs_per_range(), RND = coo_blocked_chunk_rows(), KMAX = coo_blocked_max_long_rows();
	const long cap = coo_blocked_rows_cap(A->f32);                 // chunk rows per workgroup (KMAX more slots are set aside for split rows)
	const long range_cap = cap * WGS;                              // rows per range
	const long P = std::max<long>(1, (lm + NUM_XCD * range_cap - 1) / (NUM_XCD * range_cap));
	const long NR = NUM_XCD * P, NT = NR * WGS;
	// ---- row ranges
	std::vector<int> range_row((size_t) NR + 1, 0);
	bool fits = true;
	for (long r = 1; r < NR; r++)
	{
		long row;
		if (merge_balance)
		{
			// first row whose merge-path coordinate (rows before it + entries before it) reaches the r-th share of the diagonal
			const long tgt = (long) ((double) (lm + lnnz) * r / NR);
			long lo = 0, hi = lm;
			while (lo < hi)
			{
				const long mid = (lo + hi) / 2;
				if ((long) rp[mid] + mid >= tgt)
					hi = mid;
				else
					lo = mid + 1;
			}
			row = lo;
		}
		else
		{
			const long tgt = (long) ((double) lnnz * r / NR);
			row = std::lower_bound(rp, rp + lm + 1, (int) std::min<long>(tgt, 0x7fffffffL)) - rp;
		}
		row = std::min<long>(lm, (row + RND / 2) / RND * RND);
		range_row[r] = (int) std::max<long>(row, range_row[r - 1]);
	}
	range_row[NR] = (int) lm;
	for (long r = 0; r < NR; r++)
		fits = fits && range_row[r + 1] - range_row[r] <= range_cap;
	if (!fits)
	{
		const long per = ((lm + NR - 1) / NR + RND - 1) / RND * RND;
		for (long r = 0; r <= NR; r++)
			range_row[r] = (int) std::min<long>(lm, r * per);
	}
	double v0 = 0;
	const bool uniform = values_uniform(A, va, lnnz, &v0);
	// ---- rows are dealt to the 32 workgroups of their range in chunks of CH consecutive rows (16: a chunk's y is one 128-byte line)
	long CH = RND;
	if (const char * e = getenv("SPMV_MI355X_COOB_CHUNK"))
		if (atol(e) >= RND && atol(e) % RND == 0 && atol(e) <= cap)
			CH = atol(e);
	// ---- rows to split over the 32 workgroups of their range (each takes a contiguous piece of the row's entries): a row longer than
	// 1/8 of a workgroup's fair share of the entries would unbalance the workgroup that owns it (the merge-path remedy, merge.cpp:302-318).
	// A split row needs one LDS slot in EVERY workgroup of the range; KMAX slots are set aside for them, the longest rows go first.
	// (Splitting more rows — every hub of a power-law graph — was tried and LOSES: 197 -> 250 us on the soc-LiveJournal1 twin with 1 400
	// split rows per range although the workgroups' shares get more even; a workgroup then has a dense stretch of columns of its own,
	// the 32 workgroups of an XCD drift apart in their sweep over x and their common window no longer fits the L2.)
	long long_min = std::max<long>(256, lnnz / NT / 8);
	if (const char * e = getenv("SPMV_MI355X_COOB_LONG_MIN"))      // tests: split ordinary rows of small matrices too
		if (atol(e) >= 1)
			long_min = atol(e);
	std::vector<int> range_long((size_t) NR + 1, 0), long_row;
	for (long r = 0; r < NR; r++)
	{
		const long r0 = range_row[r], r1 = range_row[r + 1];
		std::vector<std::pair<int, int>> cand;                    // (-length, row): longest first, ties in row order
		for (long row = r0; row < r1; row++)
			if (rp[row + 1] - rp[row] >= long_min)
				cand.emplace_back(-(rp[row + 1] - rp[row]), (int) row);
		if ((long) cand.size() > KMAX)
		{
			std::partial_sort(cand.begin(), cand.begin() + KMAX, cand.end());
			cand.resize((size_t) KMAX);
		}
		const size_t at = long_row.size();
		for (const auto & c : cand)
			long_row.push_back(c.second);
		std::sort(long_row.begin() + at, long_row.end());
		range_long[r + 1] = (int) long_row.size();
	}
	const long NL = (long) long_row.size();
	auto is_long = [&](long r, long row) {
		return std::binary_search(long_row.begin() + range_long[r], long_row.begin() + range_long[r + 1], (int) row);
	};
	// piece of a split row that workgroup j sums: entries [len*j/32, len*(j+1)/32)
	auto piece = [&](long row, long j, long & a, long & b) {
		const long len = rp[row + 1] - rp[row];
		a = rp[row] + len * j / WGS;
		b = rp[row] + len * (j + 1) / WGS;
	};
	// ---- the deal: longest-processing-time-first over a range's chunks, weighted by their entries (split rows excluded: every workgroup
	// gets a 32nd of those anyway), at most `cap / CH` chunks per workgroup. Every workgroup of the launch then has the same number of
	// entries to within a chunk's weight — a round-robin deal left the fullest workgroup of the soc-LiveJournal1 twin 7.7 % above the
	// mean, and the launch takes as long as its fullest workgroup. A workgroup's chunks stay in row order (LDS slot = position).
	std::vector<int> chunk_ptr((size_t) NT + 1, 0);
	std::vector<int> chunk_row;                                   // first row of every chunk, workgroup after workgroup
	{
		std::vector<std::vector<int>> mine((size_t) NT);
		#pragma omp parallel for num_threads(std::min<int>(spmv::host_threads(), (int) NR)) schedule(dynamic, 1)
		for (long r = 0; r < NR; r++)
		{
			const long r0 = range_row[r], r1 = range_row[r + 1];
			const long chunks = (r1 - r0 + CH - 1) / CH;
			const long nlong = range_long[r + 1] - range_long[r];
			const long cap_chunks = std::max<long>((chunks + WGS - 1) / WGS, cap / CH);
			std::vector<std::pair<long, int>> w((size_t) chunks);     // (-entries, chunk)
			for (long c = 0; c < chunks; c++)
			{
				const long a = r0 + c * CH, bnd = std::min(r1, a + CH);
				long cnt = rp[bnd] - rp[a];
				if (nlong)
					for (long row = a; row < bnd; row++)
						if (is_long(r, row))
							cnt -= rp[row + 1] - rp[row];
				w[(size_t) c] = std::make_pair(-cnt, (int) c);
			}
			std::sort(w.begin(), w.end());
			// min-heap of (entries so far, workgroup); a workgroup that holds cap_chunks chunks leaves it
			std::vector<std::pair<long, int>> heap;
			for (int j = 0; j < WGS; j++)
				heap.emplace_back(0L, j);
			auto cmp = [](const std::pair<long, int> & p, const std::pair<long, int> & q) { return p > q; };
			std::make_heap(heap.begin(), heap.end(), cmp);
			for (const auto & c : w)
			{
				std::pop_heap(heap.begin(), heap.end(), cmp);
				std::pair<long, int> top = heap.back();
				heap.pop_back();
				std::vector<int> & lst = mine[(size_t) (r * WGS + top.second)];
				lst.push_back(c.second);
				top.first += -c.first;
				if ((long) lst.size() < cap_chunks)
				{
					heap.push_back(top);
					std::push_heap(heap.begin(), heap.end(), cmp);
				}
			}
			for (int j = 0; j < WGS; j++)
				std::sort(mine[(size_t) (r * WGS + j)].begin(), mine[(size_t) (r * WGS + j)].end());
		}
		for (long t = 0; t < NT; t++)
			chunk_ptr[(size_t) t + 1] = chunk_ptr[(size_t) t] + (int) mine[(size_t) t].size();
		chunk_row.resize((size_t) chunk_ptr[(size_t) NT] + 1, 0);
		for (long t = 0; t < NT; t++)
			for (size_t k = 0; k < mine[(size_t) t].size(); k++)
				chunk_row[(size_t) chunk_ptr[(size_t) t] + k] = (int) (range_row[t / WGS] + (long) mine[(size_t) t][k] * CH);
	}
	// ---- entries of every workgroup: (column, LDS slot[, value]) sorted by column, cut into batches of BATCH entries. A batch has a
	// base column (its first entry's) and holds entries whose column - base fits SPAN (17 bits, or ceil(n / col_blocks) columns when
	// col_blocks > 0); what is left of a batch is padding: column offset 0, a spare LDS slot, value 0.
	const int SLOT_BITS = coo_blocked_slot_bits(), SPARE = coo_blocked_spare_slots();
	const long BATCH = coo_blocked_batch_entries(uniform);
	const long NW = 1024 / 64, KPL = BATCH / 1024;                 // waves per workgroup, entries per lane and batch
	const long SPAN = col_blocks > 0 ? std::max<long>(1, std::min<long>((n + col_blocks - 1) / col_blocks, 1L << (32 - SLOT_BITS))) : 1L << (32 - SLOT_BITS);
	std::vector<int> wg_rows((size_t) NT, 0);
	for (long t = 0; t < NT; t++)
		wg_rows[(size_t) t] = (int) ((chunk_ptr[(size_t) t + 1] - chunk_ptr[(size_t) t]) * CH + (range_long[(size_t) (t / WGS) + 1] - range_long[(size_t) (t / WGS)]));
	const long GHOST = 2;           // the kernel loads entries and base columns two batches ahead without clamping: two all-zero batches behind the last
	std::vector<int> batch_ptr((size_t) NT + 1, 0);
	const bool on_device = A->convert_on_device && NT <= (1L << 17) && SLOT_BITS == 15;
	if (on_device)
	{
		// the GPU builder (convert_coo.hip): the same bytes without the host's sorts
		if (blocked_entries_convert_device(A->f32, uniform, lm, lnnz, rp, ci, va, NR, WGS, CH, range_row, range_long, long_row, chunk_ptr, chunk_row, wg_rows, SPAN, BATCH,
				SLOT_BITS, SPARE, GHOST, batch_ptr, &A->d_coob_ent, &A->d_val, &A->d_coob_batch_base))
			return 1;
	}
	struct Ent { unsigned long long key; double v; };
	std::vector<std::vector<unsigned>> wg_ent((size_t) NT);
	std::vector<std::vector<double>> wg_val((size_t) NT);
	std::vector<std::vector<int>> wg_base((size_t) NT);
	long placed = on_device ? lnnz : 0;
	#pragma omp parallel num_threads(spmv::host_threads()) reduction(+ : placed)
	{
		std::vector<Ent> tmp;
		#pragma omp for schedule(dynamic, 4)
		for (long t = 0; t < (on_device ? 0 : NT); t++)
		{
			const long r = t / WGS, j = t % WGS;
			const long r1 = range_row[r + 1];
			const long mine = chunk_ptr[(size_t) t + 1] - chunk_ptr[(size_t) t];
			const int * cr = chunk_row.data() + chunk_ptr[(size_t) t];
			const long nlong = range_long[r + 1] - range_long[r];
			const unsigned spare0 = (unsigned) (mine * CH + nlong);          // first spare slot
			// the workgroup's entries as (first, last, LDS slot) spans: its chunk rows, then its pieces of the split rows
			tmp.clear();
			auto span = [&](long a, long b, unsigned slot) {
				for (long e = a; e < b; e++)
					tmp.push_back(Ent{((unsigned long long) (unsigned) ci[e] << 32) | slot, uniform ? 0.0 : va[e]});
			};
			for (long c = 0; c < mine; c++)
			{
				const long a = cr[c], bnd = std::min(r1, a + CH);
				for (long row = a; row < bnd; row++)
					if (!(nlong && is_long(r, row)))
						span((long) rp[row], (long) rp[row + 1], (unsigned) (c * CH + (row - a)));
			}
			for (long k = 0; k < nlong; k++)
			{
				long a, b;
				piece(long_row[(size_t) (range_long[r] + k)], j, a, b);
				span(a, b, (unsigned) (mine * CH + k));
			}
			placed += (long) tmp.size();
			std::stable_sort(tmp.begin(), tmp.end(), [](const Ent & p, const Ent & q) { return p.key < q.key; });
			std::vector<unsigned> & E = wg_ent[(size_t) t];
			std::vector<double> & V = wg_val[(size_t) t];
			std::vector<int> & Bs = wg_base[(size_t) t];
			// The sorted entries are cut into GROUPS of up to 64 = what one wave instruction gathers; a group has its own base column
			// (table [batch][wave][u]) and takes sorted entries while column - base < SPAN. Position p of a batch is entry u = p / 1024
			// of lane p % 1024, so group (u, wave) of batch b sits at b * BATCH + u * 1024 + wave * 64.
			// The sorted groups fill the batches in order: batch b takes groups 128 b .. 128 b + 127 (K = 8), (u, wave) = (0, 0), (0, 1), ...
			// (Tried and dropped: K cursors, each walking one K-th of the sorted list, so that a batch touches K short stretches of x
			// instead of one long one where the entries are sparse — 206-216 against 190 us on the soc-LiveJournal1 twin: every batch then
			// has a slow sub-batch and the waves wait for it at the barrier.)
			struct Group { size_t first; int count; int base; };
			std::vector<Group> groups;
			for (size_t k = 0; k < tmp.size();)
			{
				const long base = (long) (tmp[k].key >> 32);
				int fill = 0;
				const size_t first = k;
				for (; fill < 64 && k < tmp.size() && (long) (tmp[k].key >> 32) - base < SPAN; fill++, k++)
					;
				groups.push_back(Group{first, fill, (int) base});
			}
			const long GPB = NW * KPL;                                  // groups per batch
			long nbt = ((long) groups.size() + GPB - 1) / GPB;
			nbt = (nbt + 2) / 3 * 3;                                    // the kernel's loop is unrolled over three rotating register sets
			E.assign((size_t) (nbt * BATCH), 0u);
			if (!uniform)
				V.assign((size_t) (nbt * BATCH), 0.0);
			Bs.assign((size_t) (nbt * GPB), 0);
			int lastbase = groups.empty() ? 0 : groups[0].base;
			for (long slot = 0; slot < nbt * GPB; slot++)
			{
				const long bt = slot / GPB, u = (slot % GPB) / NW, w = slot % NW;
				const size_t at = (size_t) (bt * BATCH + u * 1024 + w * 64);
				int fill = 0;
				if (slot < (long) groups.size())
				{
					const Group & g = groups[(size_t) slot];
					lastbase = g.base;
					for (; fill < g.count; fill++)
					{
						const Ent & e = tmp[g.first + (size_t) fill];
						E[at + (size_t) fill] = (unsigned) (((long) (e.key >> 32) - g.base) << SLOT_BITS) | (unsigned) (e.key & 0xffffffffu);
						if (!uniform)
							V[at + (size_t) fill] = e.v;
					}
				}
				Bs[(size_t) ((bt * NW + w) * KPL + u)] = lastbase;
				for (; fill < 64; fill++)
					E[at + (size_t) fill] = spare0 + (unsigned) (fill % SPARE);       // padding: column offset 0, a spare LDS slot, value 0
			}
		}
	}
	if (placed != lnnz)
	{
		set_error("column-blocked layout: %ld entries placed, %ld expected", placed, lnnz);
		return 1;
	}
	int max_rows = 0;
	for (long t = 0; t < NT; t++)
	{
		if (!on_device)
			batch_ptr[(size_t) t + 1] = batch_ptr[(size_t) t] + (int) (wg_base[(size_t) t].size() / (size_t) (NW * KPL));
		max_rows = std::max(max_rows, wg_rows[(size_t) t]);
	}
	const long NB = batch_ptr[(size_t) NT];
	if (getenv("SPMV_MI355X_COOB_DEBUG"))
		for (long r = 0; r < NR; r++)
		{
			long mn = 1L << 40, mx = 0, sum = 0, ent_r = rp[range_row[r + 1]] - rp[range_row[r]];
			for (long t = r * WGS; t < (r + 1) * WGS; t++)
			{
				const long b = batch_ptr[(size_t) t + 1] - batch_ptr[(size_t) t];
				mn = std::min(mn, b), mx = std::max(mx, b), sum += b;
			}
			fprintf(stderr, "coob range %ld: rows %d..%d, %ld entries, %d split rows, batches per workgroup min %ld mean %.1f max %ld, fill %.3f, rows per workgroup %d..%d\n", r,
					range_row[r], range_row[r + 1], ent_r, range_long[r + 1] - range_long[r], mn, (double) sum / WGS, mx, (double) ent_r / ((double) sum * BATCH),
					*std::min_element(wg_rows.begin() + r * WGS, wg_rows.begin() + (r + 1) * WGS), *std::max_element(wg_rows.begin() + r * WGS, wg_rows.begin() + (r + 1) * WGS));
		}
	std::vector<int> batch_base(on_device ? 0 : (size_t) ((NB + GHOST) * NW * KPL), 0);
	std::vector<unsigned> ent(on_device ? 0 : (size_t) ((NB + GHOST) * BATCH), 0u);
	std::vector<double> pval(uniform || on_device ? 0 : (size_t) ((NB + GHOST) * BATCH), 0.0);
	#pragma omp parallel for num_threads(spmv::host_threads()) schedule(dynamic, 4)
	for (long t = 0; t < (on_device ? 0 : NT); t++)
	{
		const size_t b0 = (size_t) batch_ptr[(size_t) t];
		std::copy(wg_base[(size_t) t].begin(), wg_base[(size_t) t].end(), batch_base.begin() + b0 * (size_t) (NW * KPL));
		std::copy(wg_ent[(size_t) t].begin(), wg_ent[(size_t) t].end(), ent.begin() + b0 * BATCH);
		if (!uniform)
			std::copy(wg_val[(size_t) t].begin(), wg_val[(size_t) t].end(), pval.begin() + b0 * BATCH);
		std::vector<unsigned>().swap(wg_ent[(size_t) t]);
		std::vector<double>().swap(wg_val[(size_t) t]);
	}
	const long lext = (NB + GHOST) * BATCH;        // stored entries, padding and the two ghost batches included
	if (upload_ints(wg_rows.data(), wg_rows.size(), &A->d_coob_wg_rows) || upload_ints(range_row.data(), range_row.size(), &A->d_coob_range_row) ||
	    upload_ints(chunk_ptr.data(), chunk_ptr.size(), &A->d_coob_chunk_ptr) || upload_ints(chunk_row.data(), chunk_row.size(), &A->d_coob_chunk_row) ||
	    upload_ints(batch_ptr.data(), batch_ptr.size(), &A->d_coob_batch_ptr) ||
	    upload_ints(range_long.data(), range_long.size(), &A->d_coob_range_long) || upload_ints(long_row.data(), long_row.size(), &A->d_coob_long_row) ||
	    dev_alloc_bytes(&A->d_coob_carry, (size_t) std::max<long>(NL, 1) * WGS * A->vbytes))
		return 1;
	if (!on_device && (upload_ints(batch_base.data(), batch_base.size(), &A->d_coob_batch_base) || upload_bytes(ent.data(), (size_t) lext * 4, 64, (void **) &A->d_coob_ent)))
		return 1;
	if (!uniform && !on_device)
	{
		if (A->f32)
		{
			std::vector<float> pf32((size_t) std::max<long>(lext, 1));
			#pragma omp parallel for num_threads(spmv::host_threads())
			for (long e = 0; e < lext; e++)
				pf32[(size_t) e] = (float) pval[(size_t) e];
			if (upload_bytes(pf32.data(), (size_t) lext * 4, 64, &A->d_val))
				return 1;
		}
		else if (upload_bytes(pval.data(), (size_t) lext * 8, 64, &A->d_val))
			return 1;
	}
	if (uniform)
	{
		A->cfg.unit = 1;
		A->cfg.unit_value = v0;
	}
	A->coob_ranges = (int) NR;
	A->coob_batches = NB;
	A->coob_chunk_rows = (int) CH;
	A->coob_chunks = (long) chunk_row.size();
	A->coob_num_long = (int) NL;
	A->coob_lds = (int) (((long) (std::max(max_rows, 1) + SPARE) * 8 + 15) / 16 * 16);      // fp64 slots for both precisions
	A->cfg.map = xcd_map_uniform(1, 0);
	A->mem_footprint = (double) lext * (4 + (uniform ? 0 : A->vbytes)) + (double) (NT + 1) * 4 + (double) NB * NW * KPL * 4 + (double) NT * 8 + (double) chunk_row.size() * 4 + (double) NR * 8 + NL * (4.0 + WGS * A->vbytes);
	snprintf(A->format_name, sizeof(A->format_name), "MI355X_%s_r%ld_k%ld_%s%ld%s%s_%s", merge_balance ? "MERGEB" : "COOB", NR, BATCH / 1024, col_blocks > 0 ? "w" : "b",
			col_blocks > 0 ? SPAN : NB, NL ? "_split" : "", uniform ? "_unit" : "", pf);
	snprintf(A->kernel_name, sizeof(A->kernel_name), "coo_blocked_kernel");
	A->kernel_block = 1024;
	return 0;
}

int
build_coo_family(spmv_mi355x_matrix * A, const spmv_mi355x_opts & o, const int * rp, const int * ci, const double * va)
{
	const long lm = A->m, lnnz = A->nnz;
	const char * pf = A->f32 ? "f" : "d";
	if (o.col_blocks != 0)
		return build_blocked_layout(A, rp, ci, va, o.col_blocks, false);
	if (upload_ints(rp, (size_t) lm + 1, &A->d_row_ptr) || upload_ints(ci, (size_t) lnnz, &A->d_col) ||
	    upload_values(A, va, (size_t) lnnz, &A->d_val) || dev_alloc(&A->d_rowind, (size_t) lnnz))
		return 1;
	if (launch_expand_rows(A->d_row_ptr, (int) lm, A->d_rowind, nullptr))
		return 1;
	if (hipDeviceSynchronize() != hipSuccess)
	{
		set_error("COO row expansion failed");
		return 1;
	}
	(void) hipFree(A->d_row_ptr);          // COO keeps (rowind, colind, val) only: mkl_coo.cpp:65
	A->d_row_ptr = nullptr;
	const int per_wave = coo_wave_items(o.merge_items);
	A->coo_k = per_wave / WAVE;
	A->coo_num_waves = (int) ((lnnz + per_wave - 1) / per_wave);
	if (dev_alloc(&A->d_carry_row, (size_t) A->coo_num_waves) || dev_alloc_bytes(&A->d_carry_val, (size_t) A->coo_num_waves * A->vbytes))
		return 1;
	A->cfg.map = xcd_map_uniform((unsigned) ((A->coo_num_waves + coo_waves_per_tile() - 1) / coo_waves_per_tile()), resolve_remap(A->remap, 0));
	A->mem_footprint = (double) lnnz * (A->vbytes + 8);
	snprintf(A->format_name, sizeof(A->format_name), "MI355X_COO_k%d_%s", A->coo_k, pf);
	snprintf(A->kernel_name, sizeof(A->kernel_name), "coo_kernel");
	return 0;
}

}  // namespace spmv
