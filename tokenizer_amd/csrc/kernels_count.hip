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
// Every position is independent, so there is nothing to speculate or repair, and every wave
// is a worker of its own: it pulls 1 KiB tiles of the corpus from a queue (16 heads, one per
// wave slot of the workgroups; the first tile of a wave is static), stages tile + halo and the
// tile's document-start bitmap in its own LDS region, and its 64 lanes each take every 64th
// byte.  The next tile's bytes and offsets are loaded into registers and the tile after that
// is pulled while the current one is walked, so a wave never waits for the queue or for the
// corpus.  Counts go to u32 bins in LDS shared by the workgroup (the whole id range when V + 3
// fits, else one id range per pass over the corpus), flushed once per workgroup to a slab row
// that slab_reduce_kernel adds up.
#include "kernels.h"

#include <algorithm>

namespace tokscore {

namespace {

constexpr uint32_t kCountTile = 1024;            // corpus bytes owned by one tile (one wave)
constexpr uint32_t kCountThreads = 1024;
constexpr uint32_t kCountWaves = kCountThreads / 64;
constexpr uint32_t kPosPerLane = kCountTile / 64;
constexpr uint32_t kCountHeadStride = 32;        // queue heads one 128-byte line apart
constexpr uint32_t kNoLim = 0x7FFFFFFFu;
constexpr int kPreVec = 2;                       // register prefetch: 2 x 16 B per lane = 2 KiB

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

// first[t] = index of the first document offset strictly after tile t's first byte (a window
// is clipped at the next document start, GetTokens.cpp:25: min(parsed.size() - i, max_token_length))
__global__ void __launch_bounds__(256) tile_first_kernel(CorpusDev C, uint32_t ntiles, uint64_t *first) {
	const uint32_t t = blockIdx.x * 256 + threadIdx.x;
	if (t >= ntiles) return;
	const uint64_t a0 = (uint64_t)t * kCountTile;
	uint64_t lo = 0, up = C.D + 1;
	while (lo < up) {
		const uint64_t mid = (lo + up) >> 1;
		if (C.off[mid] > a0) up = mid;
		else lo = mid + 1;
	}
	first[t] = lo;
}

template <int K>
__global__ void __launch_bounds__(1024) count_all_kernel(TrieDev T, CorpusDev C, uint32_t max_cp, uint32_t halo,
                                                         uint32_t id_lo, uint32_t nbins, uint32_t n_ids,
                                                         uint32_t *heads, uint32_t ntiles, const uint64_t *tile_first,
                                                         uint32_t *slab, uint32_t with_positions, uint64_t readable_end) {
	extern __shared__ __attribute__((aligned(16))) unsigned char count_smem[];
	__shared__ uint32_t s_positions;
	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63;
	const uint32_t wave = tid >> 6;
	const uint32_t cap = kCountTile + halo;          // bytes staged per tile (halo is a multiple of 16)
	const uint32_t n_vec = cap / 16;
	const uint32_t bm_words = cap / 32 + 1;
	const uint32_t region = n_vec * 16 + ((bm_words * 4 + 15) & ~15u);
	LDS_AS unsigned char *lds = (LDS_AS unsigned char *)count_smem;
	LDS_AS u32x2 *ltrie = (LDS_AS u32x2 *)lds;
	LDS_AS uint8_t *tile = (LDS_AS uint8_t *)(lds + T.lds_slots * 8 + wave * region);
	LDS_AS uint32_t *bitmap = (LDS_AS uint32_t *)(tile + n_vec * 16);
	LDS_AS uint32_t *bins = (LDS_AS uint32_t *)(lds + T.lds_slots * 8 + kCountWaves * region);
	const u32x2 *gtrie = reinterpret_cast<const u32x2 *>(T.slots);

	for (uint32_t i = tid; i < T.lds_slots; i += kCountThreads) ltrie[i] = gtrie[i];
	for (uint32_t i = tid; i < nbins; i += kCountThreads) bins[i] = 0;
	if (tid == 0) s_positions = 0;
	__syncthreads();
	uint32_t positions = 0;
	const bool reg_prefetch = n_vec <= 64 * kPreVec;

	// tiles of this wave's queue head: t = k * kCountWaves + wave; k = blockIdx.x is static,
	// the heads start at gridDim.x
	uint32_t *head = heads + wave * kCountHeadStride;
	auto pull = [&]() -> uint32_t {
		uint32_t k = 0;
		if (lane == 0) k = atomicAdd(head, 1u);
		k = __builtin_amdgcn_readfirstlane(k);
		return k < 0x08000000u ? k * kCountWaves + wave : 0xFFFFFFFFu;
	};
	u32x4 pre[kPreVec];
	uint64_t pre_scan = 0, pre_off = ~0ull;
	auto prefetch = [&](uint32_t t) {
		const uint64_t base = (uint64_t)t * kCountTile;
		if (reg_prefetch) {
#pragma unroll
			for (int i = 0; i < kPreVec; i++) {
				const uint32_t v = lane + 64 * i;
				const uint64_t a = base + (uint64_t)v * 16;
				u32x4 x = {0, 0, 0, 0};
				if (v < n_vec && a + 16 <= readable_end) x = *reinterpret_cast<const u32x4 *>(C.bytes + a);
				pre[i] = x;
			}
		}
		pre_scan = tile_first[t];
		const uint64_t idx = pre_scan + lane;
		pre_off = idx <= C.D ? C.off[idx] : ~0ull;
	};

	uint32_t t0 = blockIdx.x * kCountWaves + wave;
	uint32_t t1 = 0xFFFFFFFFu;
	if (t0 < ntiles) {
		prefetch(t0);
		t1 = pull();
	}
	while (t0 < ntiles) {
		const uint64_t a0 = (uint64_t)t0 * kCountTile;
		const uint64_t left = C.total - a0;
		const uint32_t hi = left < kCountTile ? (uint32_t)left : kCountTile; // positions owned: [0, hi)
		wave_sync(); // the previous tile's walks are over
		// ---- tile + halo -> LDS, bytes as they are ----
		if (reg_prefetch) {
#pragma unroll
			for (int i = 0; i < kPreVec; i++) {
				const uint32_t v = lane + 64 * i;
				if (v < n_vec) ((LDS_AS u32x4 *)tile)[v] = pre[i];
			}
		} else {
			for (uint32_t v = lane; v < n_vec; v += 64) {
				const uint64_t a = a0 + (uint64_t)v * 16;
				u32x4 x = {0, 0, 0, 0};
				if (a + 16 <= readable_end) x = *reinterpret_cast<const u32x4 *>(C.bytes + a);
				((LDS_AS u32x4 *)tile)[v] = x;
			}
		}
		// ---- document starts after the tile's first byte -> bitmap; the offsets are sorted, so
		// the ones inside the staged bytes are a run starting at tile_first[t0] ----
		bool has_bounds = false; // wave uniform
		{
			uint64_t scan = pre_scan;
			uint64_t o = pre_off;
			for (;;) {
				const bool in_tile = o < a0 + cap; // o > a0 by the choice of tile_first
				const unsigned long long in_mask = __ballot(in_tile);
				if (in_mask && !has_bounds) {
					for (uint32_t w = lane; w < bm_words; w += 64) bitmap[w] = 0;
					wave_sync();
					has_bounds = true;
				}
				if (in_tile) {
					const uint32_t b = (uint32_t)(o - a0);
					atomicOr((uint32_t *)&bitmap[b >> 5], 1u << (b & 31));
				}
				if (in_mask != ~0ull) break;
				scan += 64;
				const uint64_t idx = scan + lane;
				o = idx <= C.D ? C.off[idx] : ~0ull;
			}
		}
		// ---- the next tile's bytes into registers, the one after that from the queue ----
		uint32_t t2 = t1;
		if (t1 < ntiles) {
			prefetch(t1);
			t2 = pull();
		}
		wave_sync();

		// ---- one flat loop.  A lane keeps K walks in flight (positions lane + 64 * (k, k + K, ..)):
		// the positions are independent, so the walks' byte and slot loads overlap -- the loop is
		// bound by the latency of those dependent loads.  A walk that ended picks its next
		// position in the same pass.
		struct Walk {
			uint32_t j, cp, nb, ncp, lim;
			bool on;
		};
		Walk w[K];
#pragma unroll
		for (int k = 0; k < K; k++) {
			w[k].j = (uint32_t)k;
			w[k].cp = 0;
			w[k].nb = 0;
			w[k].ncp = 0;
			w[k].lim = kNoLim;
			w[k].on = false;
		}
		for (;;) {
			bool busy = false;
#pragma unroll
			for (int k = 0; k < K; k++) busy = busy || w[k].on || w[k].j < kPosPerLane;
			if (!__builtin_amdgcn_ballot_w64(busy)) break;
			uint32_t c[K], slot[K];
			bool go[K];
			u32x2 e[K];
			// next position for the walks that are idle
#pragma unroll
			for (int k = 0; k < K; k++) {
				if (!w[k].on && w[k].j < kPosPerLane) {
					const uint32_t p = lane + 64 * w[k].j;
					w[k].j += K;
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
			// bytes, then slots, of all walks before any is looked at
#pragma unroll
			for (int k = 0; k < K; k++) c[k] = w[k].on ? (uint32_t)tile[w[k].cp] : 0u;
#pragma unroll
			for (int k = 0; k < K; k++) {
				const bool lead = is_lead(c[k]);
				go[k] = w[k].on && w[k].cp < w[k].lim && !(lead && w[k].ncp == max_cp);
				w[k].ncp += (go[k] && lead) ? 1u : 0u;
				slot[k] = w[k].nb + c[k];
			}
#pragma unroll
			for (int k = 0; k < K; k++) {
				e[k] = u32x2{0u, 0u};
				if (go[k]) {
					if (slot[k] < T.lds_slots) e[k] = ltrie[slot[k]];
					else e[k] = gtrie[slot[k]];
				}
			}
#pragma unroll
			for (int k = 0; k < K; k++) {
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
		t0 = t1;
		t1 = t2;
	}
	// ---- bins -> this workgroup's slab row ----
	if (with_positions) {
		for (int o = 32; o > 0; o >>= 1) positions += __shfl_down(positions, o, 64);
		if (lane == 0 && positions) atomicAdd(&s_positions, positions);
	}
	__syncthreads();
	uint32_t *row = slab + (size_t)blockIdx.x * (n_ids + 1);
	for (uint32_t i = tid; i < nbins; i += kCountThreads) row[id_lo + i] = bins[i];
	if (with_positions && tid == 0) row[n_ids] = s_positions;
}

} // namespace

// LDS split for the counting kernel: bins for as many ids as fit beside the waves' tiles, the
// rest for the hottest trie slots.  Returns false when not even one id range fits.
bool plan_count(const TrieDev &T, uint64_t max_len_bytes, uint32_t max_cp, uint32_t n_ids, CountPlan *plan) {
	const uint32_t lds_total = 160 * 1024;
	const uint64_t walk = std::min<uint64_t>(max_len_bytes, 4ull * max_cp);
	if (walk > 4096) return false;
	const uint32_t halo = (uint32_t)((walk + 16 + 15) & ~15ull);
	const uint32_t cap = kCountTile + halo;
	const uint32_t n_vec = cap / 16;
	const uint32_t bm_words = cap / 32 + 1;
	const uint32_t region = n_vec * 16 + ((bm_words * 4 + 15) & ~15u);
	const uint32_t fixed = kCountWaves * region + 1024; // + static shared / slack
	const uint32_t min_trie = 256 * 8;
	if (fixed + min_trie + 4096 > lds_total) return false;
	uint32_t nbins = std::min<uint32_t>(n_ids, (lds_total - fixed - min_trie) / 4);
	uint32_t lds_slots = (lds_total - fixed - nbins * 4) / 8;
	lds_slots = std::min<uint32_t>(lds_slots, T.nslots);
	plan->halo = halo;
	plan->nbins = nbins;
	plan->passes = (n_ids + nbins - 1) / nbins;
	plan->lds_slots = lds_slots;
	plan->lds_bytes = lds_slots * 8 + kCountWaves * region + nbins * 4;
	return true;
}

uint32_t count_tiles(uint64_t total_bytes) { return (uint32_t)((total_bytes + kCountTile - 1) / kCountTile); }
uint32_t count_blocks(uint64_t total_bytes, uint32_t num_cus) {
	const uint32_t want = (count_tiles(total_bytes) + kCountWaves - 1) / kCountWaves;
	return std::max<uint32_t>(1, std::min<uint32_t>(num_cus, want));
}
size_t count_head_words(const CountPlan &plan) { return (size_t)plan.passes * kCountWaves * kCountHeadStride; }

hipError_t launch_count_all(const TrieDev &T, const CorpusDev &C, const CountPlan &plan, uint32_t max_cp, uint32_t n_ids,
                            uint32_t blocks, uint32_t walks, uint32_t *heads, uint64_t *tile_first, uint32_t *slab,
                            hipStream_t stream) {
	TrieDev Tl = T;
	Tl.lds_slots = plan.lds_slots;
	const uint64_t readable_end = (C.total + 256) & ~15ull;
	const uint32_t ntiles = count_tiles(C.total);
	// every head starts behind the static first tiles (k = blockIdx.x)
	hipError_t e = hipMemsetD32Async((hipDeviceptr_t)heads, (int)blocks, count_head_words(plan), stream);
	if (e != hipSuccess) return e;
	if (ntiles) hipLaunchKernelGGL(tile_first_kernel, dim3((ntiles + 255) / 256), dim3(256), 0, stream, C, ntiles, tile_first);
	auto kernel = walks == 4 ? &count_all_kernel<4> : &count_all_kernel<2>;
	e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
	                        (int)plan.lds_bytes);
	if (e != hipSuccess) return e;
	for (uint32_t p = 0; p < plan.passes; p++) {
		const uint32_t id_lo = p * plan.nbins;
		const uint32_t nb = std::min<uint32_t>(plan.nbins, n_ids - id_lo);
		hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kCountThreads), plan.lds_bytes, stream, Tl, C, max_cp, plan.halo,
		                   id_lo, nb, n_ids, heads + (size_t)p * kCountWaves * kCountHeadStride, ntiles, tile_first, slab,
		                   p == 0 ? 1u : 0u, readable_end);
	}
	return hipGetLastError();
}

} // namespace tokscore
