// kernels_count.hip -- closed-vocabulary all-occurrence counting.
//
// annealing::ExtractCandidates + Trie::AddString (/root/reference/src/tokenizer/
// GetTokens.cpp:21-27, Trie.cpp:114-128) restricted to a given candidate set: for every
// code-point position of every document and every window of 1 .. max_cp code points starting
// there, the window's string gets +1 if it is a candidate.  On the byte trie that is: start a
// walk at every lead byte, stop before the (max_cp+1)-th lead byte or at the document end, and
// count every node with an id whose next byte starts a code point.  No lower-casing on this
// path (the candidate extraction never lower-cases).
//
// Every position is independent, so there is nothing to speculate or repair: a workgroup pulls
// 8 KiB tiles of the corpus from an atomic counter, stages tile + halo in LDS, and its 1,024
// threads each take every 1,024th byte.  Counts go to u32 bins in LDS (the whole id range when
// V + 3 fits, else one id range per pass over the corpus), flushed once per workgroup to a
// slab row that slab_reduce_kernel adds up.
#include "kernels.h"

#include <algorithm>

namespace tokscore {

namespace {

#define LDS_AS __attribute__((address_space(3)))
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr uint32_t kCountTile = 8192;            // corpus bytes owned by one tile
constexpr uint32_t kCountThreads = 1024;
constexpr uint32_t kSlotsPerThread = kCountTile / kCountThreads;
constexpr uint32_t kNoLim = 0x7FFFFFFFu;

__device__ __forceinline__ bool is_lead(uint32_t c) { return (c & 0xC0u) != 0x80u; } // utf8/core.h:134-147

// next document start strictly after pos (tile relative), searched up to `stop`
__device__ __forceinline__ uint32_t next_start(const LDS_AS uint32_t *bm, uint32_t pos, uint32_t stop) {
	const uint32_t i = pos + 1;
	uint32_t w = i >> 5;
	uint32_t word = bm[w] >> (i & 31);
	if (word) return i + __builtin_ctz(word);
	for (w = w + 1; (w << 5) < stop; w++) {
		word = bm[w];
		if (word) return (w << 5) + __builtin_ctz(word);
	}
	return kNoLim;
}

__global__ void __launch_bounds__(1024) count_all_kernel(TrieDev T, CorpusDev C, uint32_t max_cp, uint32_t halo,
                                                         uint32_t id_lo, uint32_t nbins, uint32_t n_ids,
                                                         uint32_t *tile_counter, uint32_t ntiles, uint32_t *slab,
                                                         uint32_t with_positions, uint64_t readable_end) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	__shared__ uint32_t s_tile;
	__shared__ unsigned long long s_first;
	__shared__ uint32_t s_bounds;
	__shared__ uint32_t s_positions;
	const uint32_t tid = threadIdx.x;
	const uint32_t cap = kCountTile + halo;          // bytes staged per tile
	const uint32_t n_vec = (cap + 15) / 16;
	const uint32_t bm_words = (cap + 31) / 32 + 1;
	LDS_AS unsigned char *lds = (LDS_AS unsigned char *)smem;
	LDS_AS u32x2 *ltrie = (LDS_AS u32x2 *)lds;
	LDS_AS uint8_t *tile = (LDS_AS uint8_t *)(lds + T.lds_slots * 8);
	LDS_AS uint32_t *bitmap = (LDS_AS uint32_t *)(lds + T.lds_slots * 8 + n_vec * 16);
	LDS_AS uint32_t *bins = bitmap + bm_words;
	const u32x2 *gtrie = reinterpret_cast<const u32x2 *>(T.slots);

	for (uint32_t i = tid; i < T.lds_slots; i += kCountThreads) ltrie[i] = gtrie[i];
	for (uint32_t i = tid; i < nbins; i += kCountThreads) bins[i] = 0;
	if (tid == 0) s_positions = 0;
	uint32_t positions = 0;

	for (;;) {
		__syncthreads(); // everyone is done with the previous tile (and with s_tile)
		if (tid == 0) {
			s_tile = atomicAdd(tile_counter, 1u);
			s_bounds = 0;
		}
		__syncthreads();
		const uint32_t t = s_tile;
		if (t >= ntiles) break;
		const uint64_t a0 = (uint64_t)t * kCountTile;
		const uint64_t left = C.total - a0;
		const uint32_t hi = left < kCountTile ? (uint32_t)left : kCountTile; // positions owned: [0, hi)
		// ---- tile + halo -> LDS, bytes as they are ----
		for (uint32_t v = tid; v < n_vec; v += kCountThreads) {
			const uint64_t a = a0 + (uint64_t)v * 16;
			u32x4 x = {0, 0, 0, 0};
			if (a + 16 <= readable_end) x = *reinterpret_cast<const u32x4 *>(C.bytes + a);
			((LDS_AS u32x4 *)tile)[v] = x;
		}
		// ---- document starts after the tile's first byte -> bitmap (a window is clipped at the
		// next one, GetTokens.cpp:25: min(parsed.size() - i, max_token_length)) ----
		for (uint32_t w = tid; w < bm_words; w += kCountThreads) bitmap[w] = 0;
		if (tid == 0) {
			uint64_t lo = 0, up = C.D + 1; // first offset index with off > a0
			while (lo < up) {
				const uint64_t mid = (lo + up) >> 1;
				if (C.off[mid] > a0) up = mid;
				else lo = mid + 1;
			}
			s_first = lo;
		}
		__syncthreads();
		for (uint64_t idx = s_first + tid; idx <= C.D; idx += kCountThreads) {
			const uint64_t o = C.off[idx];
			if (o >= a0 + cap) break;
			const uint32_t b = (uint32_t)(o - a0);
			atomicOr((uint32_t *)&bitmap[b >> 5], 1u << (b & 31));
			s_bounds = 1;
		}
		__syncthreads();
		const bool has_bounds = s_bounds != 0;

		// ---- one flat loop per wave.  A lane keeps TWO walks in flight (its even and its odd
		// positions): the positions are independent, so the second walk's byte and slot loads
		// overlap the first one's -- the loop is bound by the latency of those dependent loads.
		// A walk that ended picks its next position in the same pass.
		struct Walk {
			uint32_t j, cp, nb, ncp, lim;
			bool on;
		};
		Walk w[2];
#pragma unroll
		for (int k = 0; k < 2; k++) {
			w[k].j = (uint32_t)k;
			w[k].cp = 0;
			w[k].nb = 0;
			w[k].ncp = 0;
			w[k].lim = kNoLim;
			w[k].on = false;
		}
		while (__builtin_amdgcn_ballot_w64(w[0].on || w[1].on || w[0].j < kSlotsPerThread || w[1].j < kSlotsPerThread)) {
			uint32_t c[2], slot[2];
			bool go[2];
			u32x2 e[2];
			// next position for the walks that are idle
#pragma unroll
			for (int k = 0; k < 2; k++) {
				if (!w[k].on && w[k].j < kSlotsPerThread) {
					const uint32_t p = tid + kCountThreads * w[k].j;
					w[k].j += 2;
					if (p < hi && is_lead(tile[p])) {
						positions++;
						w[k].on = true;
						w[k].cp = p;
						w[k].nb = T.root_base;
						w[k].ncp = 0;
						w[k].lim = has_bounds ? next_start(bitmap, p, p + halo) : kNoLim;
					}
				}
			}
			// bytes, then slots, of both walks before either is looked at
#pragma unroll
			for (int k = 0; k < 2; k++) c[k] = w[k].on ? (uint32_t)tile[w[k].cp] : 0u;
#pragma unroll
			for (int k = 0; k < 2; k++) {
				const bool lead = is_lead(c[k]);
				go[k] = w[k].on && w[k].cp < w[k].lim && !(lead && w[k].ncp == max_cp);
				w[k].ncp += (go[k] && lead) ? 1u : 0u;
				slot[k] = w[k].nb + c[k];
			}
#pragma unroll
			for (int k = 0; k < 2; k++) {
				e[k] = u32x2{0u, 0u};
				if (go[k]) {
					if (slot[k] < T.lds_slots) e[k] = ltrie[slot[k]];
					else e[k] = gtrie[slot[k]];
				}
			}
#pragma unroll
			for (int k = 0; k < 2; k++) {
				bool more = false;
				if (go[k] && (e[k].x & kSlotKeyMask) == (c[k] | kSlotValid)) {
					w[k].cp++;
					w[k].nb = e[k].x >> kSlotBaseShift;
					// a candidate is a string of whole code points: it ends where one starts
					if (e[k].y != 0 && (w[k].cp >= w[k].lim || is_lead(tile[w[k].cp]))) {
						const uint32_t b = e[k].y - id_lo;
						if (b < nbins) atomicAdd((uint32_t *)&bins[b], 1u);
					}
					more = w[k].nb != 0;
				}
				w[k].on = more;
			}
		}
	}
	// ---- bins -> this workgroup's slab row ----
	if (with_positions) {
		for (int o = 32; o > 0; o >>= 1) positions += __shfl_down(positions, o, 64);
		if ((tid & 63) == 0 && positions) atomicAdd(&s_positions, positions);
	}
	__syncthreads();
	uint32_t *row = slab + (size_t)blockIdx.x * (n_ids + 1);
	for (uint32_t i = tid; i < nbins; i += kCountThreads) row[id_lo + i] = bins[i];
	if (with_positions && tid == 0) row[n_ids] = s_positions;
}

} // namespace

// LDS split for the counting kernel: bins for as many ids as fit beside the tile, the rest
// for the hottest trie slots.  Returns false when not even one id range fits.
bool plan_count(const TrieDev &T, uint64_t max_len_bytes, uint32_t max_cp, uint32_t n_ids, CountPlan *plan) {
	const uint32_t lds_total = 160 * 1024;
	const uint64_t walk = std::min<uint64_t>(max_len_bytes, 4ull * max_cp);
	if (walk > 4096) return false;
	const uint32_t halo = (uint32_t)((walk + 16 + 15) & ~15ull);
	const uint32_t cap = kCountTile + halo;
	const uint32_t n_vec = (cap + 15) / 16;
	const uint32_t bm_words = (cap + 31) / 32 + 1;
	const uint32_t fixed = n_vec * 16 + bm_words * 4 + 1024; // + static shared / slack
	const uint32_t min_trie = 256 * 8;
	if (fixed + min_trie + 4096 > lds_total) return false;
	uint32_t nbins = std::min<uint32_t>(n_ids, (lds_total - fixed - min_trie) / 4);
	uint32_t lds_slots = (lds_total - fixed - nbins * 4) / 8;
	lds_slots = std::min<uint32_t>(lds_slots, T.nslots);
	plan->halo = halo;
	plan->nbins = nbins;
	plan->passes = (n_ids + nbins - 1) / nbins;
	plan->lds_slots = lds_slots;
	plan->lds_bytes = lds_slots * 8 + n_vec * 16 + bm_words * 4 + nbins * 4;
	return true;
}

uint32_t count_tiles(uint64_t total_bytes) { return (uint32_t)((total_bytes + kCountTile - 1) / kCountTile); }

void launch_count_all(const TrieDev &T, const CorpusDev &C, const CountPlan &plan, uint32_t max_cp, uint32_t n_ids,
                      uint32_t blocks, uint32_t *tile_counters, uint32_t *slab, hipStream_t stream) {
	TrieDev Tl = T;
	Tl.lds_slots = plan.lds_slots;
	const uint64_t readable_end = (C.total + 256) & ~15ull;
	const uint32_t ntiles = count_tiles(C.total);
	(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&count_all_kernel),
	                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes);
	for (uint32_t p = 0; p < plan.passes; p++) {
		const uint32_t id_lo = p * plan.nbins;
		const uint32_t nb = std::min<uint32_t>(plan.nbins, n_ids - id_lo);
		hipLaunchKernelGGL(count_all_kernel, dim3(blocks), dim3(kCountThreads), plan.lds_bytes, stream, Tl, C, max_cp,
		                   plan.halo, id_lo, nb, n_ids, tile_counters + p, ntiles, slab, p == 0 ? 1u : 0u, readable_end);
	}
}

} // namespace tokscore
