// kernels_stream.hip -- the hot path since round 2: exact greedy segmentation with one
// independent byte stream per lane and the whole trie in LDS.
//
// What changed against the LDS-tile scan (kernels_tile.hip, still the path for vocabularies
// whose trie does not fit): the serial dependency of SolutionFile::Tokenize
// (/root/reference/src/files/SolutionFile.cpp:89-97) is no longer broken inside a wave by
// speculation and repair.  Instead every LANE owns a contiguous range of the corpus and walks one
// chain from its first token to its last.  Full evaluation: the corpus is cut into equal ranges, one
// per lane of the grid, and lane g starts at the entry it was found to have in the PREVIOUS
// evaluation (entry[g] = overshoot of the chain coming in from range g - 1; zero before the first).
// It writes its own exit overshoot; wherever exit[g - 1] != entry[g] -- everywhere in the first
// evaluation of a plan, near the occurrences of a toggled token afterwards, nowhere while the
// vocabulary stays as it is -- fixup_parallel_stream_kernel dual-walks the true chain (+1) against
// the counted one (-1) until they meet, and stores the true entry for the next evaluation.
// Incremental re-score: lanes take the streams of the selected chunks, which begin and end at
// guaranteed token starts (snap_chunks_kernel: a position no trie path reaches across, or a
// document start) and need no entries.  What the corpus needs is done once per corpus + vocabulary
// (at plan time, prep_corpus_kernel) instead of per evaluation:
//   * lower-casing (SolutionFile.cpp:87-88);
//   * document ends: a separator (code 0) is stored after every document.  Documents hold no NUL
//     (DataFile.cpp:27-29; checked, else the tile path is used), the stream image of the trie has
//     no edge labelled 0 below the root, so no match crosses a document end (the window clip of
//     SolutionFile.cpp:90) and the hot loop has no limit compare at all; the separator itself is
//     consumed as a one-byte token whose slot nobody counts;
//   * bytes become 7-bit codes (trie.h: StreamImage): the bytes that label a trie edge keep a code
//     of their own, all others share one -- that frees the label's eighth bit for "a token ends
//     here", which is what lets a trie slot be 3 bytes.
// A lane keeps the next bytes of its range in a small ring in LDS (64 or 128 bytes, refilled 16
// bytes at a time, the load of one round lands while the next round's lookups run), the trie is
// the 3-byte-per-slot stream image (trie.h: StreamImage), entirely in LDS, and a token is logged
// as the 16-bit index of the slot it ended on; hist_slots_kernel bins slot indices and
// slab_reduce_map_kernel turns slots into ids once per evaluation.
#include "kernels.h"
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace tokscore {

namespace {

// ---- prepared corpus ---------------------------------------------------------------------
// prep[off[d] + d .. off[d+1] + d) = document d lower-cased, prep[off[d+1] + d] = 0.  One thread
// per 16 output bytes.  flags[0] != 0 afterwards: some document holds a NUL byte.
__global__ void __launch_bounds__(256) prep_corpus_kernel(CorpusDev C, ByteCodes M, uint8_t *prep, uint64_t prep_total,
                                                          uint32_t *flags, uint32_t lower) {
	__shared__ uint8_t s_code[256];
	s_code[threadIdx.x] = M.code[threadIdx.x]; // blockDim.x == 256
	__syncthreads();
	const uint64_t nvec = (prep_total + 15) / 16;
	for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t q0 = v * 16;
		// document of output byte q0: the last d with off[d] + d <= q0
		uint64_t lo = 0, hi = C.D; // answer in [0, D - 1]; D >= 1 here
		while (lo + 1 < hi) {
			const uint64_t mid = (lo + hi) >> 1;
			if (C.off[mid] + mid <= q0) lo = mid;
			else hi = mid;
		}
		uint64_t d = lo;
		uint64_t dend = C.off[d + 1] + d; // position of document d's separator
		uint32_t w[4] = {0, 0, 0, 0};
		bool nul = false;
		for (uint32_t k = 0; k < 16; k++) {
			const uint64_t q = q0 + k;
			if (q >= prep_total) break;
			while (q > dend) {
				d++;
				dend = C.off[d + 1] + d;
			}
			uint32_t c = 0;
			if (q < dend) {
				c = C.bytes[q - d];
				nul = nul || c == 0;
				c = s_code[lower ? lower8(c) : c]; // (the counting path matches bytes as they are)
			}
			w[k >> 2] |= c << (8 * (k & 3));
		}
		*reinterpret_cast<u32x4 *>(prep + q0) = u32x4{w[0], w[1], w[2], w[3]};
		if (nul) flags[0] = 1u;
	}
}

// ---- the scan ------------------------------------------------------------------------------
constexpr uint32_t kBatchEntries = kStreamBatch * 512; // u16 entries of one stream batch (32 KiB); fill[batch] = entries written
constexpr uint32_t kLogEntries = 512;               // u16 entries per wave log block = one stream block (1 KiB)
constexpr uint32_t kLogHigh = kLogEntries - 64;     // flush above this fill
constexpr uint32_t kRound = 16;                     // lookups between two ring refills

struct WaveLog {
	LDS_AS uint16_t *cells;
	uint32_t cur;   // wave uniform: entries used
	uint32_t batch; // wave uniform
	uint32_t used;  // wave uniform: blocks written in the current batch
	__device__ __forceinline__ void clear(uint32_t lane) {
		LDS_AS uint32_t *w = (LDS_AS uint32_t *)cells;
		for (uint32_t i = lane; i < kLogEntries / 2; i += 64) w[i] = 0xFFFFFFFFu;
		cur = 0;
	}
	__device__ __forceinline__ void init(LDS_AS uint16_t *c, uint32_t lane) {
		cells = c;
		batch = 0xFFFFFFFFu;
		used = kStreamBatch;
		clear(lane);
	}
	// wave-uniform call: the lanes with `pred` append `val`
	__device__ __forceinline__ void put(bool pred, uint32_t val) {
		const unsigned long long m = __builtin_amdgcn_ballot_w64(pred);
		if (pred) {
			const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
			cells[cur + rank] = (uint16_t)val;
		}
		cur += (uint32_t)__builtin_popcountll(m);
	}
	__device__ __forceinline__ void flush(const StreamDev &S, uint32_t lane) {
		constexpr uint32_t vecs = kLogEntries * 2 / 16;
		if (used == kStreamBatch) {
			if (lane == 0 && batch < S.capacity_batches) S.fill[batch] = kBatchEntries;
			uint32_t b = 0;
			if (lane == 0) b = atomicAdd(&S.cursor[0], 1u);
			batch = __builtin_amdgcn_readfirstlane(b);
			used = 0;
			// cannot happen: the stream is sized for one entry per prepared byte; reported all the same
			if (batch >= S.capacity_batches && lane == 0) S.cursor[1] = 1u;
		}
		wave_sync();
		if (batch < S.capacity_batches) {
			const LDS_AS u32x4 *src = (const LDS_AS u32x4 *)cells;
			u32x4 *dst = reinterpret_cast<u32x4 *>(S.blocks) + ((size_t)batch * kStreamBatch + used) * vecs;
			for (uint32_t i = lane; i < vecs; i += 64) dst[i] = src[i];
		}
		used++;
		wave_sync();
		clear(lane);
		wave_sync();
	}
	__device__ __forceinline__ void finish(const StreamDev &S, uint32_t lane) {
		if (cur != 0) flush(S, lane);
		if (lane == 0 && batch < S.capacity_batches) S.fill[batch] = used * kLogEntries;
	}
};

// scan_ranges_kernel's log: no LDS block at all.  A wave owns one stream batch at a time and the
// lanes that complete a token in a step store their 16-bit slot indices straight to it, at
// consecutive positions (rank among them by mbcnt): one partial-line store per step that the L2
// merges, one LDS instruction per lookup fewer, and a dense stream (no null padding to read back).
struct DirectLog {
	uint16_t *base;  // wave uniform: the current batch
	uint32_t cur;    // wave uniform: bytes written to it
	uint32_t batch;  // wave uniform
	static constexpr uint32_t kHighBytes = kBatchEntries * 2 - 128; // a step adds at most 64 entries
	__device__ __forceinline__ void open(const StreamDev &S, uint32_t lane) {
		uint32_t b = 0;
		if (lane == 0) b = atomicAdd(&S.cursor[0], 1u);
		batch = __builtin_amdgcn_readfirstlane(b);
		if (batch >= S.capacity_batches) {
			// cannot happen (the stream is sized for one entry per prepared byte); reported all the same
			if (lane == 0) S.cursor[1] = 1u;
			batch = 0;
		}
		base = reinterpret_cast<uint16_t *>(S.blocks) + (size_t)batch * kBatchEntries;
		cur = 0;
	}
	// the wave's first batch is its own number: four thousand waves asking one counter at the same
	// moment would wait for it for tens of microseconds (launch_prepare starts the counter behind them)
	__device__ __forceinline__ void open_own(const StreamDev &S, uint32_t wave_id) {
		batch = wave_id;
		base = reinterpret_cast<uint16_t *>(S.blocks) + (size_t)batch * kBatchEntries;
		cur = 0;
	}
	__device__ __forceinline__ void close(const StreamDev &S, uint32_t lane) {
		if (lane == 0) S.fill[batch] = cur / 2;
	}
	__device__ __forceinline__ void roll_if_full(const StreamDev &S, uint32_t lane, uint32_t high = kHighBytes) {
		if (cur > high) {
			close(S, lane);
			open(S, lane);
		}
	}
	__device__ __forceinline__ void put(bool pred, uint32_t val) { // wave-uniform call
		const unsigned long long m = __builtin_amdgcn_ballot_w64(pred);
		if (pred) {
			const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
			base[cur / 2 + rank] = (uint16_t)val;
		}
		cur += 2 * (uint32_t)__builtin_popcountll(m);
	}
};

// One lookup for the lanes in `act` (see the file comment).  Walker state per lane:
//   cp    prepared offset of the byte looked up next
//   nb    base of the current trie node
//   lend  offset just past the longest token found so far from the current start
//   lslot the slot that token ended on
// The root row is dense and all token ends, so the first lookup of a token always sets lend /
// lslot; they need no reset between tokens.
// Ring invariant (RING >= 2 * kRound + 14 + max_len, plan_stream_scan): at the start of a round
// fill - cp > kRound and lend >= fill - RING, where [fill - RING, fill) is what the ring holds.
template <uint32_t RING>
__global__ void __launch_bounds__(1024) scan_stream_kernel(StreamTrieDev T, StreamWork W, StreamDev Qd) {
	extern __shared__ __attribute__((aligned(16))) unsigned char stream_smem[];
	LDS_AS unsigned char *lds = (LDS_AS unsigned char *)stream_smem;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = uni32(threadIdx.x >> 6);
	const uint32_t waves = blockDim.x >> 6;

	// ---- stage the trie image: lt[] then base[] (both padded to 16 bytes by the host) ----
	const uint32_t label_bytes = (T.nslots + 15) & ~15u;
	const uint32_t info_bytes = (T.nslots * 2 + 15) & ~15u;
	{
		LDS_AS u32x4 *dst = (LDS_AS u32x4 *)lds;
		const u32x4 *src = reinterpret_cast<const u32x4 *>(T.lt);
		for (uint32_t i = threadIdx.x; i < label_bytes / 16; i += blockDim.x) dst[i] = src[i];
		dst = (LDS_AS u32x4 *)(lds + label_bytes);
		src = reinterpret_cast<const u32x4 *>(T.base);
		for (uint32_t i = threadIdx.x; i < info_bytes / 16; i += blockDim.x) dst[i] = src[i];
	}
	__syncthreads();
	const LDS_AS uint8_t *l_lt = (const LDS_AS uint8_t *)lds;
	const LDS_AS uint16_t *l_base = (const LDS_AS uint16_t *)(lds + label_bytes);

	LDS_AS unsigned char *wbase = lds + label_bytes + info_bytes + wave * (64 * RING + kLogEntries * 2);
	LDS_AS unsigned char *ring = wbase + lane * RING;
	WaveLog log;
	log.init((LDS_AS uint16_t *)(wbase + 64 * RING), lane);

	const uint32_t total_lanes = gridDim.x * waves * 64;
	const uint32_t gl = (blockIdx.x * waves + wave) * 64 + lane;
	uint32_t n_items, my_items = 0;
	if (W.tg_items) {
		// incremental re-score with the occurrence index: the streams of the kToggleLists lists
		// (launch_find_toggle_streams), lane q walks list q (total_lanes is a multiple of 64)
		static_assert(kToggleLists == 64, "one list per lane");
		my_items = W.tg_count[lane] < W.tg_cap ? W.tg_count[lane] : W.tg_cap;
		uint32_t most = my_items;
		for (int o = 32; o > 0; o >>= 1) most = max(most, (uint32_t)__shfl_xor((int)most, o, 64));
		n_items = most * 64;
	} else if (W.sel_list) n_items = W.sel_count[0] * W.streams_per_chunk; // incremental re-score without the index: the selected chunks' streams
	else n_items = W.nranges;                                       // full evaluation: equal ranges

	for (uint32_t item = gl; __builtin_amdgcn_ballot_w64(item < n_items) != 0; item += total_lanes) {
		uint32_t beg = 0, end = 0;
		if (W.tg_items) {
			if ((item >> 6) < my_items) {
				const uint32_t strm = W.tg_items[(size_t)lane * W.tg_cap + (item >> 6)].stream;
				beg = W.pbeg[strm];
				end = W.pbeg[strm + 1];
			}
		} else if (item < n_items) {
			if (W.sel_list) {
				uint32_t s0, s1;
				const uint32_t chunk = W.sel_list[item / W.streams_per_chunk];
				s0 = W.chunk_first[chunk] + item % W.streams_per_chunk;
				s1 = s0 + 1;
				if (s0 >= W.chunk_first[chunk + 1]) s0 = s1 = W.nstreams; // fewer streams in this chunk
				if (s0 > W.nstreams) s0 = W.nstreams;
				if (s1 > W.nstreams) s1 = W.nstreams;
				beg = W.pbeg[s0];
				end = W.pbeg[s1];
			} else {
				beg = item * W.range_bytes;
				end = beg + W.range_bytes < W.prep_total ? beg + W.range_bytes : W.prep_total;
				beg += W.entry[item]; // (may lie at or past `end`: the chain jumps the whole range)
			}
		}
		const bool has_range = !W.tg_items && !W.sel_list && item < n_items;
		bool active = beg < end;
		uint32_t cp = beg, lend = beg, lslot = 0, nb = 0; // (the root's base is 0)
		uint32_t fill = beg & ~15u;
		wave_sync();
		if (active) {
#pragma unroll
			for (uint32_t k = 0; k < RING / 16; k++) {
				const u32x4 x = *reinterpret_cast<const u32x4 *>(W.prep + fill);
				*(LDS_AS u32x4 *)(ring + (fill & (RING - 1))) = x;
				fill += 16;
			}
		}
		wave_sync();
		bool req = false;
		u32x4 data = {0, 0, 0, 0};
		unsigned long long act = __builtin_amdgcn_ballot_w64(active);
		while (act != 0) {
			// ---- ring refill: what was asked for a round ago lands, the next 16 bytes are asked for ----
			if (req) {
				*(LDS_AS u32x4 *)(ring + (fill & (RING - 1))) = data;
				fill += 16;
			}
			req = active && (fill - lend) <= RING - 16;
			if (req) data = *reinterpret_cast<const u32x4 *>(W.prep + fill);
			wave_sync();
			// ---- kRound lookups ----
			for (uint32_t it = 0; it < kRound; it++) {
				bool done = false;
				if (active) {
					const uint32_t c = ring[cp & (RING - 1)];
					const uint32_t t = nb + c;
					const uint32_t x = l_lt[t] ^ c; // 0x80: hit on a token end, 0: hit, else miss
					const uint32_t b = l_base[t];
					const uint32_t cp1 = cp + 1;
					const bool term = x == 0x80u;
					lend = term ? cp1 : lend;
					lslot = term ? t : lslot;
					nb = (x & 0x7Fu) == 0 ? b : 0u;
					const bool cont = nb != 0;
					cp = cont ? cp1 : lend;
					done = !cont;
				}
				log.put(done, lslot);
				if (done) active = cp < end; // (nb is 0 = the root again)
				if (log.cur > kLogHigh) log.flush(Qd, lane);
				act = __builtin_amdgcn_ballot_w64(active);
				if (act == 0) break;
			}
		}
		if (has_range) W.exit_a[item] = cp - end; // overshoot of the range's last token (cp is its end)
	}
	log.finish(Qd, lane);
}

// ---- batched move evaluation: which streams? -------------------------------------------------
// pairs[pair_off[m] ..] = the chunks move m's string occurs in (its row of the occurrence index),
// in any order; what the row's count promised and the bitmap does not hold is marked invalid.
__global__ void __launch_bounds__(1024) list_toggle_pairs_kernel(ToggleBatch B) {
	__shared__ uint32_t s_n;
	const uint32_t m = blockIdx.x;
	const uint32_t *row = B.index + (size_t)B.rows[m] * B.words_per_row;
	const uint32_t base = B.pair_off[m], cap = B.pair_off[m + 1] - base;
	if (threadIdx.x == 0) s_n = 0;
	__syncthreads();
	for (uint32_t j = threadIdx.x; j < B.words_per_row; j += blockDim.x) {
		uint32_t w = row[j];
		if (!w) continue;
		uint32_t at = atomicAdd(&s_n, (uint32_t)__builtin_popcount(w));
		while (w) {
			if (at < cap) B.pairs[base + at] = uint4{j * 32 + (uint32_t)__builtin_ctz(w), m << 10, 1u, 0u}; // (fill_toggle_pairs_kernel completes it)
			w &= w - 1;
			at++;
		}
	}
	__syncthreads();
	for (uint32_t j = s_n + threadIdx.x; j < cap; j += blockDim.x) B.pairs[base + j] = uint4{0u, m << 10, 0u, 0u};
}

// A pair's record as the search wants it (ToggleBatch::pairs): looked up here, one thread per pair,
// where the latency of the four dependent loads is hidden, not in front of every search wave.
__global__ void __launch_bounds__(256) fill_toggle_pairs_kernel(ToggleBatch B, uint32_t npairs) {
	const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= npairs) return;
	const uint4 r = B.pairs[p];
	if (r.z == 0) return; // nothing there
	const uint32_t s0 = B.chunk_first[r.x], s1 = B.chunk_first[r.x + 1];
	const uint32_t ns = s1 - s0 < 512 ? s1 - s0 : 512; // (plan_stream: at most 8 * 64 streams in a chunk)
	B.pairs[p] = uint4{s0, ns | r.y, B.pbeg[s0], B.pbeg[s1]};
}

// One wave per (move, chunk) pair: the chunk's bytes of the prepared corpus are searched for the
// move's string (in code space; a string with a byte the corpus cannot hold never matches), every
// stream with an occurrence becomes one item.  Stream cuts are guaranteed token starts -- no trie
// path crosses one -- so an occurrence lies inside one stream and a move changes nothing outside
// the streams listed here.  The items go to kToggleLists lists (list = workgroup % kToggleLists,
// one counter each, one atomic per workgroup: a single counter takes about 90 atomics per
// microsecond, and a batch has hundreds of thousands of pairs); lane q of the walk reads list q.
__global__ void __launch_bounds__(1024) find_toggle_streams_kernel(ToggleBatch B, StreamWork W, uint32_t nchunks) {
	__shared__ uint32_t s_bits[16][16]; // per wave: up to 512 streams of one chunk (plan_stream: at most 8 * 64)
	__shared__ uint32_t s_cnt[16], s_at[16];
	__shared__ uint32_t s_hits[16][64];                 // per wave and stream (the first 64 of a chunk): occurrences seen
	__shared__ uint16_t s_off[16][64][kToggleOffsets];  // ... and where, from the stream's begin
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t p = blockIdx.x * 16 + wave;
	const uint32_t npairs = B.pair_off[B.n];
	uint32_t m = 0, s0 = 0, ns = 0;
	if (lane < 16) s_bits[wave][lane] = 0;
	s_hits[wave][lane] = 0;
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	const uint4 pair = p < npairs ? B.pairs[p] : uint4{0u, 0u, 0u, 0u};
	ns = pair.y & 1023u;
	if (ns != 0) {
		m = pair.y >> 10;
		s0 = pair.x;
		const uint32_t a = pair.z, b = pair.w;
		const uint4 mv = B.moves[m];
		const uint32_t L = mv.y, pat = mv.z, msk = mv.w;
		const uint8_t *str = B.codes + mv.x;
		if (L >= 2 && b - a >= L) { // (a one-byte string cannot be toggled as a token end: the root row is all token ends)
			// 16 positions per lane and step, the string's first four bytes compared as one dword at
			// every byte alignment; what matches there is compared on from memory
			for (uint32_t base = (a & ~15u) + lane * 16; base < b; base += 64 * 16) {
				const u32x4 v = *reinterpret_cast<const u32x4 *>(W.prep + base);
				const uint32_t w[5] = {v[0], v[1], v[2], v[3], *reinterpret_cast<const uint32_t *>(W.prep + base + 16)};
				uint32_t hits = 0; // bit 4 j + i: position base + 4 j + i
#pragma unroll
				for (uint32_t j = 0; j < 4; j++) {
#pragma unroll
					for (uint32_t i = 0; i < 4; i++) {
						const uint32_t d = i == 0 ? w[j] : __builtin_amdgcn_alignbyte(w[j + 1], w[j], i);
						hits |= ((d ^ pat) & msk) == 0 ? 1u << (4 * j + i) : 0u;
					}
				}
				while (hits) {
					const uint32_t q = base + (uint32_t)__builtin_ctz(hits);
					hits &= hits - 1;
					if (q < a || q + L > b) continue;
					uint32_t k = 4;
					while (k < L && W.prep[q + k] == str[k]) k++;
					if (k < L) continue;
					// Both chains start a stream together and take the same tokens until one starts
					// where the string occurs: an occurrence inside a token of the vocabulary as it is
					// cannot be where they part
					if (B.starts && !((B.starts[q >> 6] >> (q & 63)) & 1ull)) continue;
					uint32_t l = s0, h = s0 + ns; // last stream with pbeg <= q
					while (h - l > 1) {
						const uint32_t mid = (l + h) >> 1;
						if (W.pbeg[mid] <= q) l = mid;
						else h = mid;
					}
					const uint32_t si = l - s0, rel = q - W.pbeg[l];
					atomicOr(&s_bits[wave][si >> 5], 1u << (si & 31));
					if (si < 64) {
						const uint32_t k = rel < 0xFFFFu ? atomicAdd(&s_hits[wave][si], 1u) : atomicAdd(&s_hits[wave][si], 1000u);
						if (k < kToggleOffsets) s_off[wave][si][k] = (uint16_t)rel;
					}
				}
			}
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
	if (B.seen) {
		// a stream another pair (another move, another wave) listed already is not listed again
		for (uint32_t i0 = 0; i0 < ns; i0 += 64) {
			const uint32_t i = i0 + lane;
			if (i < ns && ((s_bits[wave][i >> 5] >> (i & 31)) & 1u)) {
				const uint32_t sidx = s0 + i, bit = 1u << (sidx & 31);
				if (atomicOr(&B.seen[sidx >> 5], bit) & bit) atomicAnd(&s_bits[wave][i >> 5], ~(1u << (i & 31)));
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
	}
	{
		uint32_t n = lane < 16 ? (uint32_t)__builtin_popcount(s_bits[wave][lane]) : 0u;
		for (int o = 8; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
		if (lane == 0) s_cnt[wave] = n;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t total = 0;
		for (uint32_t k = 0; k < 16; k++) {
			s_at[k] = total;
			total += s_cnt[k];
		}
		const uint32_t first = total ? atomicAdd(&B.n_items[blockIdx.x % kToggleLists], total) : 0u;
		for (uint32_t k = 0; k < 16; k++) s_at[k] += first;
	}
	__syncthreads();
	ToggleItem *list = B.items + (size_t)(blockIdx.x % kToggleLists) * B.cap_items;
	uint32_t at = s_at[wave];
	for (uint32_t i0 = 0; i0 < ns; i0 += 64) {
		const uint32_t i = i0 + lane;
		const bool hit = i < ns && ((s_bits[wave][i >> 5] >> (i & 31)) & 1u);
		const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
		if (hit) {
			const uint32_t r = at + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
			if (r < B.cap_items) {
				ToggleItem it;
				it.stream = s0 + i;
				it.move = m;
				const uint32_t n = i < 64 ? s_hits[wave][i] : 1000u;
				it.n = n <= kToggleOffsets ? (uint16_t)n : kToggleWhole;
#pragma unroll
				for (uint32_t k = 0; k < kToggleOffsets; k++) it.off[k] = i < 64 && k < n ? s_off[wave][i][k] : (uint16_t)0;
				list[r] = it;
			}
		}
		at += (uint32_t)__builtin_popcountll(mask);
	}
}

// ---- batched move evaluation: the walks ---------------------------------------------------------
// n 16-byte pieces from global memory to LDS, by the whole workgroup: all of a thread's loads are under way before its
// first store (a loop of load, wait, store is a round trip per 16 KB: 10 - 15 us of a call of 64 moves, whose kernels
// each stage the trie image for a few dozen microseconds of work)
__device__ __forceinline__ void fill_lds_16(const u32x4 *src, LDS_AS u32x4 *dst, uint32_t n) {
	constexpr int kInFlight = 8;
	for (uint32_t base = threadIdx.x; base < n; base += kInFlight * blockDim.x) {
		u32x4 v[kInFlight];
#pragma unroll
		for (int k = 0; k < kInFlight; k++) {
			const uint32_t i = base + k * blockDim.x;
			v[k] = src[i < n ? i : base];
		}
#pragma unroll
		for (int k = 0; k < kInFlight; k++) {
			const uint32_t i = base + k * blockDim.x;
			if (i < n) dst[i] = v[k];
		}
	}
}

// The lookup loop of scan_stream_kernel, counting tokens per stream instead of logging them, with the
// token-end bit of one slot flipped per lane (the move).  Without W.tg_items the items are all
// streams and the counts go to W.tg_base (the vocabulary as it is: the context keeps them until a
// toggle is applied); with it the items are the (move, stream) pairs of the kToggleLists lists, and
// the difference to the stream's count in W.tg_base is added to W.tg_delta[move].
// A wave owns a contiguous share of the items and a lane takes the next one of them whenever it has
// finished a stream, so a wave's time is its share's bytes over 64 and not the sum of the longest
// stream of every round.  Differences are summed per lane while the move stays the same (shares
// follow the lists' order, which is the moves' order) and across the wave at the end: one address
// takes about 90 atomics per microsecond.
template <uint32_t RING>
__global__ void __launch_bounds__(1024) walk_toggles_kernel(StreamTrieDev T, StreamWork W) {
	extern __shared__ __attribute__((aligned(16))) unsigned char walk_smem[];
	__shared__ uint32_t s_count[kToggleLists];
	LDS_AS unsigned char *lds = (LDS_AS unsigned char *)walk_smem;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = uni32(threadIdx.x >> 6);
	const uint32_t waves = blockDim.x >> 6;
	const bool listed = W.tg_items != nullptr;
	const uint32_t label_bytes = (T.nslots + 15) & ~15u;
	const uint32_t info_bytes = (T.nslots * 2 + 15) & ~15u;
	{
		fill_lds_16(reinterpret_cast<const u32x4 *>(T.lt), (LDS_AS u32x4 *)lds, label_bytes / 16);
		fill_lds_16(reinterpret_cast<const u32x4 *>(T.base), (LDS_AS u32x4 *)(lds + label_bytes), info_bytes / 16);
		if (threadIdx.x < kToggleLists) s_count[threadIdx.x] = listed ? min(W.tg_count[threadIdx.x], W.tg_cap) : 0u;
	}
	__syncthreads();
	const LDS_AS uint8_t *l_lt = (const LDS_AS uint8_t *)lds;
	const LDS_AS uint16_t *l_base = (const LDS_AS uint16_t *)(lds + label_bytes);
	LDS_AS unsigned char *ring = lds + label_bytes + info_bytes + (wave * 64 + lane) * RING;

	// flat item f: stream f, or entry f / kToggleLists of list f % kToggleLists
	uint32_t total = W.nstreams;
	if (listed) {
		uint32_t most = s_count[lane];
		for (int o = 32; o > 0; o >>= 1) most = max(most, (uint32_t)__shfl_xor((int)most, o, 64));
		total = most * kToggleLists;
	}
	const uint32_t total_waves = gridDim.x * waves;
	const uint32_t share = ((total + total_waves - 1) / total_waves + 63) & ~63u;
	const uint64_t lo64 = (uint64_t)(blockIdx.x * waves + wave) * share;
	uint32_t next = uni32(lo64 < total ? (uint32_t)lo64 : total);
	uint32_t stop = uni32(lo64 + share < total ? (uint32_t)(lo64 + share) : total);
	// All streams, longest first (W.tg_order): wave w takes entries w, w + waves, w + 2 waves, .. of
	// that order -- every wave a like mix, and the long streams are under way when the short ones
	// fill the gaps, instead of one 2 KB stream deciding when the launch ends.
	const bool ordered = !listed && W.tg_order != nullptr;
	const uint32_t wave_global = blockIdx.x * waves + wave;
	if (ordered) {
		next = 0;
		stop = uni32((total + total_waves - 1) / total_waves);
	}

	bool active = false, req = false;
	uint32_t beg = 0, end = 0, cp = 0, lend = 0, lslot = 0, nb = 0, fill = 0, ntok = 0;
	uint32_t strm = 0, move = 0, flip_slot = 0xFFFFFFFFu;
	uint32_t acc_move = 0xFFFFFFFFu;
	int acc = 0;
	u32x4 data = {0, 0, 0, 0};
	auto finish = [&]() { // this lane's stream is done
		if (!listed) {
			W.tg_base[strm] = ntok;
			return;
		}
		const int d = (int)ntok - (int)W.tg_base[strm]; // with the move - without
		if (d == 0) return;
		if (move != acc_move) {
			if (acc != 0) atomicAdd(&W.tg_delta[acc_move], (unsigned long long)(long long)acc);
			acc_move = move;
			acc = 0;
		}
		acc += d;
	};
	for (;;) {
		// ---- lanes without a stream take the next items of the wave's share ----
		const unsigned long long idle = __builtin_amdgcn_ballot_w64(!active);
		if (idle != 0 && next < stop) {
			const uint32_t f = next + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
			next = uni32(next + (uint32_t)__builtin_popcountll(idle));
			if (!active && f < stop) {
				bool have = true;
				strm = f;
				if (ordered) {
					const uint64_t at = (uint64_t)f * total_waves + wave_global;
					have = at < total;
					strm = have ? W.tg_order[at] : 0u;
				}
				if (listed) {
					const uint32_t q = f % kToggleLists, idx = f / kToggleLists;
					have = idx < s_count[q];
					if (have) {
						size_t at = (size_t)q * W.tg_cap + idx;
						if (W.tg_index) at = W.tg_index[at];
						const ToggleItem *it = &W.tg_items[at];
						have = W.tg_n ? W.tg_n[at] + 1u > W.tg_whole : it->n == kToggleWhole; // (the others: local_toggles_kernel)
						strm = it->stream;
						move = it->move;
						if (have) flip_slot = W.tg_slot[move];
					}
				}
				if (have) {
					beg = W.pbeg[strm];
					end = W.pbeg[strm + 1];
					ntok = 0;
					cp = lend = beg;
					lslot = 0, nb = 0;
					req = false;
					if (beg < end) {
						active = true;
						fill = beg & ~15u;
#pragma unroll
						for (uint32_t k = 0; k < RING / 16; k++) {
							const u32x4 x = *reinterpret_cast<const u32x4 *>(W.prep + fill);
							*(LDS_AS u32x4 *)(ring + (fill & (RING - 1))) = x;
							fill += 16;
						}
					} else {
						finish(); // an empty stream (a dropped cut)
					}
				}
			}
		}
		wave_sync();
		if (__builtin_amdgcn_ballot_w64(active) == 0) {
			if (next >= stop) break;
			continue;
		}
		// ---- ring refill, as in scan_stream_kernel ----
		if (req) {
			*(LDS_AS u32x4 *)(ring + (fill & (RING - 1))) = data;
			fill += 16;
		}
		req = active && (fill - lend) <= RING - 16;
		if (req) data = *reinterpret_cast<const u32x4 *>(W.prep + fill);
		wave_sync();
		for (uint32_t it = 0; it < kRound; it++) {
			if (active) {
				const uint32_t c = ring[cp & (RING - 1)];
				const uint32_t t = nb + c;
				uint32_t x = l_lt[t] ^ c; // 0x80: hit on a token end, 0: hit, else miss
				if (t == flip_slot) x ^= 0x80u;
				const uint32_t b = l_base[t];
				const uint32_t cp1 = cp + 1;
				const bool term = x == 0x80u;
				lend = term ? cp1 : lend;
				lslot = term ? t : lslot;
				nb = (x & 0x7Fu) == 0 ? b : 0u;
				const bool cont = nb != 0;
				cp = cont ? cp1 : lend;
				if (!cont) {
					ntok += lslot != 0 ? 1u : 0u; // (slot 0: a document separator, no token)
					if (cp >= end) {
						active = false;
						req = false;
						finish();
					}
				}
			}
			if (__builtin_amdgcn_ballot_w64(active) == 0) break;
		}
	}
	if (listed) {
		// what the lanes still hold: one atomic per move and wave
		unsigned long long pend = __builtin_amdgcn_ballot_w64(acc != 0);
		while (pend) {
			const uint32_t leader = (uint32_t)__builtin_ctzll(pend);
			const uint32_t m0 = (uint32_t)__builtin_amdgcn_readlane((int)acc_move, (int)leader);
			const bool mine = acc != 0 && acc_move == m0;
			int sum = mine ? acc : 0;
			for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
			if (lane == leader) atomicAdd(&W.tg_delta[m0], (unsigned long long)(long long)sum);
			pend &= ~__builtin_amdgcn_ballot_w64(mine);
		}
	}
}

// ---- batched move evaluation: the vocabulary as it is ---------------------------------------------
// Tokens per stream (W.tg_base) and where tokens start (W.tg_starts, one bit per byte, zeroed by the caller) under
// the trie as it stands: every stream walked once.  Streams are a few hundred bytes, so what a
// lane spends per stream outside the walk decides: a lane takes the streams L, L + lanes, L + 2
// lanes, .. of the longest-first order (W.tg_order: every lane a like mix) and has the next one's
// cuts and first bytes in registers before the current one ends -- its number, its cuts and its
// bytes are three dependent loads, issued one per round of lookups, each used a round later.
template <uint32_t RING>
__global__ void __launch_bounds__(1024) count_streams_kernel(StreamTrieDev T, StreamWork W) {
	extern __shared__ __attribute__((aligned(16))) unsigned char count_streams_smem[];
	LDS_AS unsigned char *lds = (LDS_AS unsigned char *)count_streams_smem;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = uni32(threadIdx.x >> 6);
	const uint32_t label_bytes = (T.nslots + 15) & ~15u;
	const uint32_t info_bytes = (T.nslots * 2 + 15) & ~15u;
	{
		LDS_AS u32x4 *dst = (LDS_AS u32x4 *)lds;
		const u32x4 *src = reinterpret_cast<const u32x4 *>(T.lt);
		for (uint32_t i = threadIdx.x; i < label_bytes / 16; i += blockDim.x) dst[i] = src[i];
		dst = (LDS_AS u32x4 *)(lds + label_bytes);
		src = reinterpret_cast<const u32x4 *>(T.base);
		for (uint32_t i = threadIdx.x; i < info_bytes / 16; i += blockDim.x) dst[i] = src[i];
	}
	__syncthreads();
	const LDS_AS uint8_t *l_lt = (const LDS_AS uint8_t *)lds;
	const LDS_AS uint16_t *l_base = (const LDS_AS uint16_t *)(lds + label_bytes);
	LDS_AS unsigned char *ring = lds + label_bytes + info_bytes + (wave * 64 + lane) * RING;
	const uint32_t total = W.nstreams;
	const uint32_t lanes = gridDim.x * blockDim.x;
	uint64_t at = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; // position in the order of this lane's next stream

	bool active = false, req = false;
	uint32_t strm = 0, end = 0, cp = 0, lend = 0, lslot = 0, nb = 0, fill = 0, ntok = 0;
	uint32_t stage = 0, n_strm = 0, n_beg = 0, n_end = 0; // the stream after this one: 0 nothing yet, 1 number, 2 cuts, 3 bytes asked for
	u32x4 n_pre[RING / 16];
	u32x4 data = {0, 0, 0, 0};
	// token starts: one bit per prepared byte, collected per 64-byte window in a register and OR-ed
	// into the map when the walk leaves the window (a byte store per token was what the kernel
	// waited for: 26 M scattered partial-line writes)
	unsigned long long wmask = 0;
	uint32_t wword = 0;
	auto mark = [&](uint32_t q) {
		const uint32_t w = q >> 6;
		if (w != wword) {
			if (wmask) atomicOr(&W.tg_starts[wword], wmask);
			wword = w, wmask = 0;
		}
		wmask |= 1ull << (q & 63);
	};
	for (;;) {
		// ---- one step of the next stream's preparation; its loads are used a round from now ----
		if (stage == 2) {
			if (n_beg < n_end) {
#pragma unroll
				for (uint32_t k = 0; k < RING / 16; k++) n_pre[k] = *reinterpret_cast<const u32x4 *>(W.prep + (n_beg & ~15u) + 16 * k);
			}
			stage = 3;
		} else if (stage == 1) {
			n_beg = W.pbeg[n_strm];
			n_end = W.pbeg[n_strm + 1];
			stage = 2;
		} else if (stage == 0 && at < total) {
			n_strm = W.tg_order ? W.tg_order[at] : (uint32_t)at;
			at += lanes;
			stage = 1;
		}
		if (!active && stage == 3) {
			strm = n_strm;
			end = n_end;
			cp = lend = n_beg;
			lslot = 0, nb = 0, ntok = 0;
			req = false;
			stage = 0;
			if (n_beg < n_end) {
				active = true;
				if (W.tg_starts) mark(n_beg);
				fill = n_beg & ~15u;
#pragma unroll
				for (uint32_t k = 0; k < RING / 16; k++) {
					*(LDS_AS u32x4 *)(ring + (fill & (RING - 1))) = n_pre[k];
					fill += 16;
				}
			} else {
				W.tg_base[strm] = 0; // an empty stream (a dropped cut)
			}
		}
		wave_sync();
		if (__builtin_amdgcn_ballot_w64(active) == 0) {
			if (__builtin_amdgcn_ballot_w64(stage != 0 || at < total) == 0) break;
			continue;
		}
		// ---- ring refill and a round of lookups, as in scan_stream_kernel ----
		if (req) {
			*(LDS_AS u32x4 *)(ring + (fill & (RING - 1))) = data;
			fill += 16;
		}
		req = active && (fill - lend) <= RING - 16;
		if (req) data = *reinterpret_cast<const u32x4 *>(W.prep + fill);
		wave_sync();
		for (uint32_t it = 0; it < kRound; it++) {
			if (active) {
				const uint32_t c = ring[cp & (RING - 1)];
				const uint32_t t = nb + c;
				const uint32_t x = l_lt[t] ^ c; // 0x80: hit on a token end, 0: hit, else miss
				const uint32_t b = l_base[t];
				const uint32_t cp1 = cp + 1;
				const bool term = x == 0x80u;
				lend = term ? cp1 : lend;
				lslot = term ? t : lslot;
				nb = (x & 0x7Fu) == 0 ? b : 0u;
				const bool cont = nb != 0;
				cp = cont ? cp1 : lend;
				if (!cont) {
					ntok += lslot != 0 ? 1u : 0u; // (slot 0: a document separator, no token)
					if (cp < end) {
						if (W.tg_starts) mark(cp); // the next token starts here
					} else {
						active = false;
						req = false;
						W.tg_base[strm] = ntok;
						if (wmask) atomicOr(&W.tg_starts[wword], wmask);
						wmask = 0;
					}
				}
			}
			if (__builtin_amdgcn_ballot_w64(active) == 0) break;
		}
	}
}

// ---- batched move evaluation: local walks --------------------------------------------------------
// An item with occurrence offsets.  The chain of the vocabulary as it is (A) and the chain with the
// move's bit flipped (B) take the same tokens from the stream's begin until a token starts where the
// string occurs; from there both are walked, whichever is behind next, until they stand on the same
// position again -- a few dozen bytes, never past the stream's end (its cut is a token start of
// every chain).  What lies between is counted once: occurrences are taken in ascending order and
// those the last walk has passed are skipped.  Bytes come straight from the prepared corpus in
// global memory (the walks are too short for a ring), the trie image from LDS.
// (win / win_at: kLocalWindow bytes of the prepared corpus from win_at on, in LDS: a walk's bytes are mostly there, and a
// byte from global memory costs a round trip of its own on a chain of dependent lookups)
constexpr uint32_t kLocalWindow = 48; // (at most: what LDS has left beside the trie image decides, launch_local_toggles)
__device__ __forceinline__ uint32_t local_token(const LDS_AS uint8_t *l_lt, const LDS_AS uint16_t *l_base, const uint8_t *prep,
                                                uint32_t pos, uint32_t flip_slot, uint32_t &counts, const LDS_AS uint8_t *win, uint32_t win_at,
                                                uint32_t window) {
	uint32_t nb = 0, d = 0, len = 1, slot = 0;
	uint32_t w4 = 0, w4_at = 0xFFFFFFFFu; // four bytes of the window at a time: one LDS round trip less on three steps of four
	do {
		const uint32_t rel = pos + d - win_at;
		uint32_t c;
		if (rel < window) {
			if ((rel & ~3u) != w4_at) {
				w4_at = rel & ~3u;
				w4 = *(const LDS_AS uint32_t *)(win + w4_at);
			}
			c = (w4 >> (8 * (rel & 3u))) & 0xFFu;
		} else {
			c = prep[pos + d];
		}
		const uint32_t t = nb + c;
		uint32_t x = l_lt[t] ^ c;
		if (t == flip_slot) x ^= 0x80u;
		if ((x & 0x7Fu) != 0) break;
		d++;
		if (x == 0x80u) len = d, slot = t;
		nb = l_base[t];
	} while (nb != 0);
	counts += slot != 0 ? 1u : 0u; // (slot 0: a document separator, no token)
	return pos + len;
}

constexpr uint32_t kLocalMoves = 1024;
__global__ void __launch_bounds__(1024) local_toggles_kernel(StreamTrieDev T, StreamWork W, uint32_t window) {
	extern __shared__ __attribute__((aligned(16))) unsigned char local_smem[];
	__shared__ uint32_t s_count[kToggleLists];
	__shared__ int s_delta[kLocalMoves];
	const bool by_block = W.tg_moves != 0 && W.tg_moves <= kLocalMoves;
	if (by_block)
		for (uint32_t i = threadIdx.x; i < W.tg_moves; i += blockDim.x) s_delta[i] = 0;
	LDS_AS unsigned char *lds = (LDS_AS unsigned char *)local_smem;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t label_bytes = (T.nslots + 15) & ~15u;
	const uint32_t info_bytes = (T.nslots * 2 + 15) & ~15u;
	{
		fill_lds_16(reinterpret_cast<const u32x4 *>(T.lt), (LDS_AS u32x4 *)lds, label_bytes / 16);
		fill_lds_16(reinterpret_cast<const u32x4 *>(T.base), (LDS_AS u32x4 *)(lds + label_bytes), info_bytes / 16);
		if (threadIdx.x < kToggleLists) s_count[threadIdx.x] = min(W.tg_count[threadIdx.x], W.tg_cap);
	}
	__syncthreads();
	const LDS_AS uint8_t *l_lt = (const LDS_AS uint8_t *)lds;
	const LDS_AS uint16_t *l_base = (const LDS_AS uint16_t *)(lds + label_bytes);
	LDS_AS uint8_t *win = lds + label_bytes + info_bytes + threadIdx.x * window; // this lane's window of the corpus
	uint32_t most = s_count[lane];
	for (int o = 32; o > 0; o >>= 1) most = max(most, (uint32_t)__shfl_xor((int)most, o, 64));
	const uint32_t total_lanes = gridDim.x * blockDim.x;
	// lane q of every wave reads list q (total_lanes is a multiple of 64)
	for (uint32_t idx = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; __builtin_amdgcn_ballot_w64(idx < most) != 0; idx += total_lanes >> 6) {
		int d = 0;
		uint32_t move = 0;
		if (idx < s_count[lane]) {
			size_t at = (size_t)lane * W.tg_cap + idx;
			if (W.tg_index) at = W.tg_index[at]; // (items made from the one-pass search: the lists hold their places in a table ...
			ToggleItem it = W.tg_items[at];
			if (W.tg_n) { // ... and the count of the occurrences after the first is kept beside them)
				const uint32_t n = W.tg_n[at] + 1u;
				it.n = n > W.tg_whole ? kToggleWhole : (uint16_t)n;
			}
			if (it.n != kToggleWhole && it.n != 0) {
				move = it.move;
				const uint32_t flip_slot = W.tg_slot[move];
				const uint32_t beg = W.pbeg[it.stream];
				uint32_t cur = 0; // offsets below this lie inside what has been walked
				for (uint32_t round = 0; round < it.n; round++) {
					uint32_t best = 0xFFFFFFFFu;
#pragma unroll
					for (uint32_t k = 0; k < kToggleOffsets; k++)
						if (k < it.n && it.off[k] >= cur && it.off[k] < best) best = it.off[k];
					if (best == 0xFFFFFFFFu) break;
					uint32_t na = 0, nb2 = 0;
					uint32_t win_at = (beg + best) & ~15u;
					auto load_window = [&]() {
						const u32x4 *src = reinterpret_cast<const u32x4 *>(W.prep + win_at); // (the prepared corpus is padded)
						LDS_AS u32x4 *dst = (LDS_AS u32x4 *)win;
						u32x4 v[kLocalWindow / 16]; // (all loads, then the stores: one round trip, not one per piece)
#pragma unroll
						for (uint32_t k = 0; k < kLocalWindow / 16; k++)
							if (k * 16 < window) v[k] = src[k];
#pragma unroll
						for (uint32_t k = 0; k < kLocalWindow / 16; k++)
							if (k * 16 < window) dst[k] = v[k];
					};
					load_window();
					if (W.tg_known) {
						// The chain of the vocabulary as it is needs no walk: it stands where W.tg_known has a bit.  Only the
						// chain with the move is walked, until it stands on one of those -- that is where the two meet -- and
						// the difference is its steps minus the bits passed (separators are a step and a bit alike).
						uint32_t bw = 0;
						unsigned long long bits0 = 0, bits1 = 0; // the map from the window's first byte on (128 positions and more)
						auto load_bits = [&]() {
							bw = win_at >> 6;
							bits0 = W.tg_known[bw];
							bits1 = W.tg_known[bw + 1]; // (the map has a word to spare)
						};
						auto word = [&](uint32_t w) { return w == bw ? bits0 : w == bw + 1 ? bits1 : W.tg_known[w]; };
						load_bits();
						const uint32_t q0 = beg + best;
						uint32_t pb = q0, steps = 0;
						do {
							if (window >= 32 && pb + 24 > win_at + window) {
								win_at = pb & ~15u;
								load_window();
								load_bits();
							}
							uint32_t cnt = 0;
							pb = local_token(l_lt, l_base, W.prep, pb, flip_slot, cnt, win, win_at, window);
							steps++;
						} while (pb < W.prep_total && !((word(pb >> 6) >> (pb & 63)) & 1ull));
						uint32_t passed = 0; // bits in [q0, pb)
						for (uint32_t w = q0 >> 6; w <= (pb - 1) >> 6; w++) {
							unsigned long long x = word(w);
							if (w == q0 >> 6) x &= ~0ull << (q0 & 63);
							if (w == (pb - 1) >> 6) x &= ~0ull >> (63 - ((pb - 1) & 63));
							passed += (uint32_t)__builtin_popcountll(x);
						}
						d += (int)steps - (int)passed;
						cur = pb - beg;
						continue;
					}
					// (one call per step, for whichever chain is behind: a branch per chain made every wave run both)
					uint32_t pa = local_token(l_lt, l_base, W.prep, beg + best, 0xFFFFFFFFu, na, win, win_at, window);
					uint32_t pb = beg + best;
					while (pa != pb) {
						const bool a = pa < pb;
						const uint32_t p = a ? pa : pb;
						// (the window follows the chain that is behind: a byte beyond it is a round trip to global memory of its
						// own, and the walks that go on for a hundred bytes were what a wave waited for)
						if (window >= 32 && p + 24 > win_at + window) {
							win_at = p & ~15u;
							load_window();
						}
						uint32_t cnt = 0;
						const uint32_t r = local_token(l_lt, l_base, W.prep, p, a ? 0xFFFFFFFFu : flip_slot, cnt, win, win_at, window);
						if (a) pa = r, na += cnt;
						else pb = r, nb2 += cnt;
					}
					d += (int)nb2 - (int)na;
					cur = pa - beg;
				}
			}
		}
		if (by_block) { // (items in no order of moves: per workgroup first, in LDS)
			if (d != 0) atomicAdd(&s_delta[move], d);
			continue;
		}
		// one atomic per move and wave (neighbouring items mostly belong to one move)
		unsigned long long pend = __builtin_amdgcn_ballot_w64(d != 0);
		while (pend) {
			const uint32_t leader = (uint32_t)__builtin_ctzll(pend);
			const uint32_t m0 = (uint32_t)__builtin_amdgcn_readlane((int)move, (int)leader);
			const bool mine = d != 0 && move == m0;
			int sum = mine ? d : 0;
			for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
			if (lane == leader) atomicAdd(&W.tg_delta[m0], (unsigned long long)(long long)sum);
			pend &= ~__builtin_amdgcn_ballot_w64(mine);
		}
	}
	if (by_block) {
		__syncthreads();
		for (uint32_t i = threadIdx.x; i < W.tg_moves; i += blockDim.x)
			if (s_delta[i] != 0) atomicAdd(&W.tg_delta[i], (unsigned long long)(long long)s_delta[i]);
	}
}

// ---- full evaluation: equal ranges, refilled from the transposed corpus -------------------------
// The ranges of one wave (64 consecutive ones) are stored interleaved in 8-byte pieces, piece k of
// lane l's range at block + (k * 64 + l) * 8 (transpose_ranges_kernel), each range followed by
// `halo` bytes of the next.  The 64 lanes of a wave advance at nearly the same pace, so a refill --
// every lane that has room loads its next piece -- reads a few whole cache lines instead of 64
// different ones, and 8-byte pieces let a lane's LDS ring be 32 bytes (plan_ranges_scan): with the
// 1 KiB log block a wave needs 3 KiB of LDS, and sixteen waves fit beside the trie image.
// Ring invariant, pieces of u = 8 bytes and rounds of K lookups (K <= u): RING >= 2 K + u + max_len - 2.
// At the start of a round, after the piece asked for a round ago has landed, the ring holds
// [fill - RING, fill) with lend >= fill - RING (nothing a walk can come back to is overwritten:
// a piece is only asked for while fill - lend <= RING - u) and fill - cp > K (a round reads bytes
// below cp + K + 1): a lane without a request has fill - cp >= RING - u - max_len + 3, loses at
// most K per round until its next piece lands, and gains u - K >= 0 per round after that.
//
// The loop over rounds is hand-written gfx950 assembly (TOK_RANGES_ASM): the scan is bound by the
// number of instructions a SIMD has to issue per lookup and by the two dependent LDS round trips
// of a lookup (byte, then slot), so what counts is the instruction count, scalar ones included,
// and that the next byte's read is under way while the finished tokens are logged.  A round of K
// lookups is unrolled (K is a template parameter): per lookup 17 vector, 3 LDS, 1 store and 10
// scalar instructions with no loop counter and no branch, plus 18 per round for the refill, the log's
// capacity check and the loop (the compiler's code for the same loop, kept below for TOKSCORE_ASM=0,
// needs about 50 per lookup).
// Conventions: exec = lanes still inside their range (ACT); v24 = the byte at cp, v31 = the byte at
// lend; v25 (slot), v27-v30 belong to the lookup whose reads are in flight; v26, v32-v34, s[38:43] scratch.  The block is entered and left at the start of a
// round: left when no lane is active any more or when the stream batch may not take another round
// (the C++ around it opens the next batch and comes back), with nothing in flight.
// Timing experiments on the round loop (csrc/Makefile `exp`, tools/dbg/ab_libs.sh; never in the product
// library): TOK_EXP_NOSTORE drops the token store (results wrong, the scan 2 % faster: the store is not
// what the loop waits for); TOK_EXP_EXTRA_RT puts one more dependent LDS read in front of every
// lookup (results unchanged, + 11 %: what a round trip on the chain costs).
#ifdef TOK_EXP_NOSTORE
#define TOK_EXP_STORE ""
#else
#define TOK_EXP_STORE "global_store_short v33, %[lslot], %[sbase]\n s_bcnt1_i32_b64 s40, exec\n s_lshl1_add_u32 %[cur], s40, %[cur]\n"
#endif
#ifdef TOK_EXP_EXTRA_RT
#define TOK_EXP_CHAIN "ds_read_u8 v26, v24\n s_waitcnt lgkmcnt(0)\n v_and_b32 v26, 0, v26\n v_or_b32 v24, v24, v26\n"
#else
#define TOK_EXP_CHAIN ""
#endif
#if defined(TOK_EXP_EXTRA_RANDOM)
#define TOK_EXP_READ "ds_read_u8 v32, v26\n" /* one more independent read at a random address */
#elif defined(TOK_EXP_EXTRA_VALU4)
#define TOK_EXP_READ "v_mov_b32 v32, v32\n v_mov_b32 v32, v32\n v_mov_b32 v32, v32\n v_mov_b32 v32, v32\n" /* four more vector instructions */
#elif defined(TOK_EXP_EXTRA_SALU4)
#define TOK_EXP_READ "s_mov_b32 s40, s40\n s_mov_b32 s40, s40\n s_mov_b32 s40, s40\n s_mov_b32 s40, s40\n" /* four more scalar instructions */
#elif defined(TOK_EXP_EXTRA_BCAST)
#define TOK_EXP_READ "v_mov_b32 v32, 0\n ds_read_u8 v32, v32\n" /* ... at one address for all lanes */
#else
#define TOK_EXP_READ ""
#endif
// LDS address of ring byte POS: dword row (POS / 4) mod (RING / 4) of the wave's ring block (256 bytes a
// row: a dword per lane, so whatever positions the lanes are at they read 64 different banks), byte POS mod 4
#define TOK_RING_BYTE_ADDR(DST, POS)                                                         \
	"v_and_b32 " DST ", %[rrows], " POS "\n"                                                 \
	"v_lshl_add_u32 " DST ", " DST ", 6, %[rbase]\n"                                         \
	"v_and_or_b32 " DST ", " POS ", 3, " DST "\n"
#ifdef TOK_EXP_NOPRIO
#define TOK_PRIO_HIGH ""
#define TOK_PRIO_LOW ""
#else
#define TOK_PRIO_HIGH "s_setprio 2\n"
#define TOK_PRIO_LOW "s_setprio 0\n"
#endif
// the reads of one lookup: the slot the byte at cp (v24) leads to from node nb, and the byte at cp + 1
// (v30; beside the slot reads, so that no lookup waits for a byte: the next byte is v30 or v31)
#define TOK_ISSUE_LOOKUP                                                                     \
	"v_add_u32 v25, %[nb], v24\n"                 /* slot t = LDS address of lt[t]     */     \
	"v_lshl_add_u32 v26, v25, 1, %[basea]\n"      /* address of base[t]                */     \
	"ds_read_u8 v27, v25\n"                                                                   \
	"ds_read_u16 v28, v26\n"                                                                  \
	TOK_EXP_READ                                                                              \
	"v_add_u32 v29, 1, %[cp]\n"                                                               \
	TOK_RING_BYTE_ADDR("v30", "v29")                                                          \
	"ds_read_u8 v30, v30\n"
#define TOK_RANGES_ASM(K, HIGH)                                                              \
	asm volatile(                                                                             \
	    "s_mov_b64 s[38:39], exec\n"                                                          \
	    "s_mov_b64 %[act], %[actin]\n"                                                        \
	    "s_mov_b64 %[req], %[reqin]\n"                                                        \
	    "s_mov_b32 %[cur], %[curin]\n"                                                        \
	    "s_mov_b64 exec, %[act]\n"                                                            \
	    TOK_RING_BYTE_ADDR("v25", "%[cp]")                                                    \
	    TOK_RING_BYTE_ADDR("v26", "%[lend]")                                                  \
	    "ds_read_u8 v24, v25\n"                                                               \
	    "ds_read_u8 v31, v26\n"                                                               \
	    "s_waitcnt lgkmcnt(0)\n"                                                              \
	    TOK_ISSUE_LOOKUP                               /* the first lookup's reads          */ \
	    ".Lround%=:\n"                                /* refill, then a round of K lookups */ \
	    "s_cmp_eq_u64 %[act], 0\n"                                                            \
	    "s_cbranch_scc1 .Lout%=\n"                                                            \
	    "s_cmp_gt_u32 %[cur], %[high]\n"              /* the batch may not take another round */ \
	    "s_cbranch_scc1 .Lout%=\n"                                                            \
	    "s_waitcnt vmcnt(4)\n"                        /* (in issue order: the piece came before the round's >= 4 stores) */ \
	    "s_mov_b64 exec, %[req]\n"                    /* the piece asked for a round ago   */ \
	    "v_and_b32 v34, %[rmask], %[fill]\n"          /* (a multiple of 8: dword rows 2 j, 2 j + 1) */ \
	    "v_lshl_add_u32 v34, v34, 6, %[rbase]\n"                                              \
	    "ds_write2st64_b32 v34, %[d0], %[d1] offset1:1\n"                                     \
	    "v_add_u32 %[fill], 8, %[fill]\n"                                                     \
	    "s_mov_b64 exec, %[act]\n"                                                            \
	    "v_sub_u32 v34, %[fill], %[lend]\n"                                                   \
	    "v_cmp_ge_u32 vcc, %[room], v34\n"            /* room for another piece            */ \
	    "s_mov_b64 %[req], vcc\n"                                                             \
	    "s_mov_b64 exec, vcc\n"                                                               \
	    "v_lshl_add_u32 v34, %[fill], 6, %[tb]\n"                                             \
	    "s_waitcnt lgkmcnt(0)\n"                      /* (d0, d1 are free again)           */ \
	    "global_load_dword %[d0], v34, %[tprep]\n"                                            \
	    "global_load_dword %[d1], v34, %[tprep] offset:4\n"                                   \
	    "s_mov_b64 exec, %[act]\n"                                                            \
	    ".rept " #K "\n"                                                                      \
	    TOK_PRIO_HIGH                                  /* the dependent part of a lookup goes before other waves' logging */ \
	    TOK_EXP_CHAIN                                                                         \
	    "s_waitcnt lgkmcnt(0)\n"                                                              \
	    "v_xor_b32 v27, v27, v24\n"                   /* 0x80 hit on a token end, 0 hit    */ \
	    "v_cmp_ne_u32 s[42:43], 0, v28\n"             /* the slot has children             */ \
	    "v_cmp_eq_u32 s[40:41], %[k80], v27\n"        /* token end                         */ \
	    "v_and_b32 v26, 0x7f, v27\n"                                                          \
	    "v_cmp_eq_u32 vcc, 0, v26\n"                  /* hit                               */ \
	    "v_cndmask_b32 %[lend], %[lend], v29, s[40:41]\n" /* (gfx950: two states between a VALU's mask and the VALU reading it) */ \
	    "v_cndmask_b32 %[lslot], %[lslot], v25, s[40:41]\n"                                   \
	    "v_cndmask_b32 v31, v31, v30, s[40:41]\n"     /* v31 = the byte at lend, kept in a register */ \
	    "s_and_b64 s[40:41], vcc, s[42:43]\n"         /* the token goes on                 */ \
	    "v_cndmask_b32 %[cp], %[lend], v29, s[40:41]\n" /* ... or the next token starts at lend */ \
	    "v_cndmask_b32 %[nb], 0, v28, s[40:41]\n"     /* token complete: 0 = the root again */ \
	    "v_cndmask_b32 v24, v31, v30, s[40:41]\n"     /* the byte at the new cp: no read to wait for */ \
	    TOK_ISSUE_LOOKUP                               /* the next lookup's reads, under way while this one's token is logged */ \
	    TOK_PRIO_LOW                                                                          \
	    "s_andn2_b64 exec, exec, s[40:41]\n"          /* lanes whose token is complete     */ \
	    "v_mbcnt_lo_u32_b32 v33, exec_lo, 0\n"                                                \
	    "v_mbcnt_hi_u32_b32 v33, exec_hi, v33\n"      /* rank among them                   */ \
	    "v_lshl_add_u32 v33, v33, 1, %[cur]\n"        /* byte offset in the wave's stream batch */ \
	    TOK_EXP_STORE                                                                         \
	    "v_cmp_ge_u32 vcc, %[cp], %[hi]\n"            /* range done                        */ \
	    "s_andn2_b64 %[act], %[act], vcc\n"                                                   \
	    "s_mov_b64 exec, %[act]\n"                                                            \
	    ".endr\n"                                                                             \
	    "s_branch .Lround%=\n"                                                                \
	    ".Lout%=:\n"                                                                          \
	    "s_waitcnt vmcnt(0) lgkmcnt(0)\n"                                                     \
	    "s_mov_b64 exec, s[38:39]\n"                                                          \
	    : [cp] "+v"(cp), [nb] "+v"(nb), [lend] "+v"(lend), [lslot] "+v"(lslot), [fill] "+v"(fill),      \
	      [d0] "+v"(data0), [d1] "+v"(data1), [act] "=&s"(act_out), [req] "=&s"(req_out), [cur] "=&s"(cur_out)           \
	    : [actin] "s"(uni64(act)), [reqin] "s"(uni64(req)), [curin] "s"(uni32(log.cur)), [sbase] "s"(uni64((unsigned long long)(uintptr_t)log.base)), \
	      [hi] "v"(end), [rbase] "v"(ring_a), [tb] "v"(tb), [rmask] "s"(RING - 1), [rrows] "s"(RING - 4), [basea] "s"(label_bytes), \
	      [high] "s"(HIGH), [k80] "s"(0x80u), [room] "s"(RING - 8), [tprep] "s"(W.tprep)                 \
	    : "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "s38", "s39", "s40", "s41", "s42", "s43", "vcc", "scc", "memory")

// K: lookups per round of the assembly loop (0: the C++ loop, rounds of W.round)
template <uint32_t RING, uint32_t K>
__global__ void __launch_bounds__(1024) scan_ranges_kernel(StreamTrieDev T, StreamWork W, StreamDev Qd, OutInit I) {
	extern __shared__ __attribute__((aligned(16))) unsigned char ranges_smem[];
	LDS_AS unsigned char *lds = (LDS_AS unsigned char *)ranges_smem;
	if (I.out) // (nothing else touches the output vector before this kernel has ended)
		for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < I.V + 5; i += gridDim.x * blockDim.x)
			I.out[i] = i < 2 ? I.D : (i == I.V + 4 ? I.total_bytes : 0ull);
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = uni32(threadIdx.x >> 6);
	const uint32_t waves = blockDim.x >> 6;
	const uint32_t label_bytes = (T.nslots + 15) & ~15u;
	const uint32_t info_bytes = (T.nslots * 2 + 15) & ~15u;
	{
		LDS_AS u32x4 *dst = (LDS_AS u32x4 *)lds;
		const u32x4 *src = reinterpret_cast<const u32x4 *>(T.lt);
		for (uint32_t i = threadIdx.x; i < label_bytes / 16; i += blockDim.x) dst[i] = src[i];
		dst = (LDS_AS u32x4 *)(lds + label_bytes);
		src = reinterpret_cast<const u32x4 *>(T.base);
		for (uint32_t i = threadIdx.x; i < info_bytes / 16; i += blockDim.x) dst[i] = src[i];
	}
	__syncthreads();
	const LDS_AS uint8_t *l_lt = (const LDS_AS uint8_t *)lds;
	const LDS_AS uint16_t *l_base = (const LDS_AS uint16_t *)(lds + label_bytes);
	if (K != 0 && (uint32_t)(uintptr_t)lds != 0) {
		// the assembly takes a slot index for the LDS address of its label (the image starts at LDS
		// address 0: this kernel has no static LDS); reported like a stream overflow, never ignored
		if (threadIdx.x == 0) Qd.cursor[1] = 1u;
		return;
	}
	// The wave's ring block: RING / 4 rows of 64 dwords, row j = dword j of every lane's ring (lane l's
	// at 4 l): a lane's ring bytes are 256 apart per dword, and a read of one byte per lane touches 64
	// different banks wherever the lanes stand.  (With each lane's ring in one piece, lanes 8 apart
	// shared their banks: the read cost as much as a random one, a quarter of the kernel.)
	LDS_AS unsigned char *wbase = lds + ((label_bytes + info_bytes + 127) & ~127u) + wave * (64 * RING);
	LDS_AS unsigned char *ring = wbase + lane * 4;
	auto ring_at = [&](uint32_t pos) { return ring + ((pos & (RING - 4)) << 6) + (pos & 3u); }; // byte pos of the prepared range
	DirectLog log;
	log.open_own(Qd, blockIdx.x * waves + wave);

	const uint32_t total_waves = gridDim.x * waves;
	const uint32_t rows = W.range_bytes + W.halo; // bytes a lane's range has in its block (multiple of 8)
	const uint32_t round = W.round;
	for (uint32_t wb = blockIdx.x * waves + wave; wb * 64 < W.nranges; wb += total_waves) {
		const uint32_t g = wb * 64 + lane;
		// byte x (a multiple of 8) of this lane's range starts the piece at blk + x * 64
		const uint8_t *blk = W.tprep + (size_t)wb * rows * 64 + lane * 8;
		uint32_t end = 0, cp = 0;
		if (g < W.nranges) {
			const uint32_t b = g * W.range_bytes;
			end = b + W.range_bytes < W.prep_total ? W.range_bytes : W.prep_total - b;
			cp = W.entry[g]; // (may lie at or past `end`: the chain jumps the whole range)
		}
		bool active = cp < end;
		uint32_t lend = cp, lslot = 0, nb = 0; // (the root's base is 0)
		uint32_t fill = cp & ~7u;
		wave_sync();
		if (active) {
#pragma unroll
			for (uint32_t k = 0; k < RING / 8; k++) {
				const u32x2 x = *reinterpret_cast<const u32x2 *>(blk + (size_t)fill * 64);
				*(LDS_AS uint32_t *)ring_at(fill) = x.x;
				*(LDS_AS uint32_t *)ring_at(fill + 4) = x.y;
				fill += 8;
			}
		}
		wave_sync();
		if constexpr (K != 0) {
			const uint32_t ring_a = (uint32_t)(uintptr_t)ring;
			const uint32_t tb = (uint32_t)((size_t)wb * rows * 64 + lane * 8); // (the transposed copy is below 4 GiB)
			unsigned long long act = __builtin_amdgcn_ballot_w64(active), req = 0;
			uint32_t data0 = 0, data1 = 0;
			constexpr uint32_t high = kBatchEntries * 2 - 128 * K; // a round adds at most 64 K entries
			while (act != 0) {
				unsigned long long act_out, req_out;
				uint32_t cur_out;
				if constexpr (K == 4) TOK_RANGES_ASM(4, high);
				else if constexpr (K == 5) TOK_RANGES_ASM(5, high);
				else if constexpr (K == 6) TOK_RANGES_ASM(6, high);
				else if constexpr (K == 7) TOK_RANGES_ASM(7, high);
				else TOK_RANGES_ASM(8, high);
				act = act_out, req = req_out;
				log.cur = cur_out;
				log.roll_if_full(Qd, lane, high);
			}
		} else {
			bool req = false;
			u32x2 data = {0, 0};
			unsigned long long act = __builtin_amdgcn_ballot_w64(active);
			while (act != 0) {
				if (req) {
					*(LDS_AS uint32_t *)ring_at(fill) = data.x;
					*(LDS_AS uint32_t *)ring_at(fill + 4) = data.y;
					fill += 8;
				}
				req = active && (fill - lend) <= RING - 8;
				if (req) data = *reinterpret_cast<const u32x2 *>(blk + (size_t)fill * 64);
				wave_sync();
				for (uint32_t it = 0; it < round; it++) {
					bool done = false;
					if (active) {
						const uint32_t c = *ring_at(cp);
						const uint32_t t = nb + c;
						const uint32_t x = l_lt[t] ^ c;
						const uint32_t b = l_base[t];
						const uint32_t cp1 = cp + 1;
						const bool term = x == 0x80u;
						lend = term ? cp1 : lend;
						lslot = term ? t : lslot;
						nb = (x & 0x7Fu) == 0 ? b : 0u;
						const bool cont = nb != 0;
						cp = cont ? cp1 : lend;
						done = !cont;
					}
					log.put(done, lslot);
					if (done) active = cp < end; // (nb is 0 = the root again)
					log.roll_if_full(Qd, lane);
					act = __builtin_amdgcn_ballot_w64(active);
					if (act == 0) break;
				}
			}
		}
		if (g < W.nranges) W.exit_a[g] = cp - end;
	}
	log.close(Qd, lane);
}

// ---- batched moves: token starts and tokens per stream, in ranges form ---------------------------
// What count_streams_kernel leaves behind (W.tg_starts: one bit per prepared byte where a token of the
// vocabulary as it is starts; W.tg_base: tokens per stream, ADDED here, so zeroed by the caller), from
// the ranges a full evaluation has just scanned: W.entry holds the true entry of every range after
// that evaluation's fix-up, so every lane can walk its own range of the transposed corpus exactly as
// scan_ranges_kernel does -- coalesced refills, equal work per lane -- instead of one stream per
// lane from the linear corpus (0.9 ms per 100 MB, a third of an annealing batch; this: 0.2).
// A token belongs to the stream it starts in (stream cuts are token starts of every chain).
template <uint32_t RING>
__global__ void __launch_bounds__(1024) mark_ranges_kernel(StreamTrieDev T, StreamWork W) {
	extern __shared__ __attribute__((aligned(16))) unsigned char mark_smem[];
	LDS_AS unsigned char *lds = (LDS_AS unsigned char *)mark_smem;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = uni32(threadIdx.x >> 6);
	const uint32_t waves = blockDim.x >> 6;
	const uint32_t label_bytes = (T.nslots + 15) & ~15u;
	const uint32_t info_bytes = (T.nslots * 2 + 15) & ~15u;
	{
		fill_lds_16(reinterpret_cast<const u32x4 *>(T.lt), (LDS_AS u32x4 *)lds, label_bytes / 16);
		fill_lds_16(reinterpret_cast<const u32x4 *>(T.base), (LDS_AS u32x4 *)(lds + label_bytes), info_bytes / 16);
	}
	__syncthreads();
	const LDS_AS uint8_t *l_lt = (const LDS_AS uint8_t *)lds;
	const LDS_AS uint16_t *l_base = (const LDS_AS uint16_t *)(lds + label_bytes);
	LDS_AS unsigned char *ring = lds + ((label_bytes + info_bytes + 127) & ~127u) + wave * (64 * RING) + lane * 4;
	auto ring_at = [&](uint32_t pos) { return ring + ((pos & (RING - 4)) << 6) + (pos & 3u); };
	const uint32_t total_waves = gridDim.x * waves;
	const uint32_t rows = W.range_bytes + W.halo;
	const uint32_t round = W.round;
	for (uint32_t wb = blockIdx.x * waves + wave; wb * 64 < W.nranges; wb += total_waves) {
		const uint32_t g = wb * 64 + lane;
		const uint8_t *blk = W.tprep + (size_t)wb * rows * 64 + lane * 8;
		uint32_t end = 0, cp = 0, b0 = 0;
		if (g < W.nranges) {
			b0 = g * W.range_bytes;
			end = b0 + W.range_bytes < W.prep_total ? W.range_bytes : W.prep_total - b0;
			cp = W.entry[g];
		}
		bool active = cp < end;
		uint32_t lend = cp, lslot = 0, nb = 0;
		uint32_t fill = cp & ~7u;
		// the stream the range's first token starts in, and where that stream ends
		uint32_t sidx = 0, cut = 0, ntok = 0;
		unsigned long long wmask = 0;
		uint32_t wword = 0;
		auto mark = [&](uint32_t q) {
			const uint32_t w = q >> 6;
			if (w != wword) {
				if (wmask) atomicOr(&W.tg_starts[wword], wmask);
				wword = w, wmask = 0;
			}
			wmask |= 1ull << (q & 63);
		};
		wave_sync();
		if (active) {
#pragma unroll
			for (uint32_t k = 0; k < RING / 8; k++) {
				const u32x2 x = *reinterpret_cast<const u32x2 *>(blk + (size_t)fill * 64);
				*(LDS_AS uint32_t *)ring_at(fill) = x.x;
				*(LDS_AS uint32_t *)ring_at(fill + 4) = x.y;
				fill += 8;
			}
			const uint32_t q = b0 + cp;
			uint32_t l = 0, h = W.nstreams; // last stream with pbeg <= q (pbeg[0] = 0)
			if (W.block_stream) { // (a table of the stream at every 256th byte: two round trips instead of nineteen)
				l = W.block_stream[q >> 8];
				while (l + 1 < W.nstreams && W.pbeg[l + 1] <= q) l++;
			} else {
				while (h - l > 1) {
					const uint32_t mid = (l + h) >> 1;
					if (W.pbeg[mid] <= q) l = mid;
					else h = mid;
				}
			}
			sidx = l;
			cut = W.pbeg[sidx + 1];
			mark(q);
		}
		wave_sync();
		bool req = false;
		u32x2 data = {0, 0};
		unsigned long long act = __builtin_amdgcn_ballot_w64(active);
		while (act != 0) {
			if (req) {
				*(LDS_AS uint32_t *)ring_at(fill) = data.x;
				*(LDS_AS uint32_t *)ring_at(fill + 4) = data.y;
				fill += 8;
			}
			req = active && (fill - lend) <= RING - 8;
			if (req) data = *reinterpret_cast<const u32x2 *>(blk + (size_t)fill * 64);
			wave_sync();
			for (uint32_t it = 0; it < round; it++) {
				if (active) {
					const uint32_t c = *ring_at(cp);
					const uint32_t t = nb + c;
					const uint32_t x = l_lt[t] ^ c;
					const uint32_t b = l_base[t];
					const uint32_t cp1 = cp + 1;
					const bool term = x == 0x80u;
					lend = term ? cp1 : lend;
					lslot = term ? t : lslot;
					nb = (x & 0x7Fu) == 0 ? b : 0u;
					const bool cont = nb != 0;
					cp = cont ? cp1 : lend;
					if (!cont) {
						ntok += lslot != 0 ? 1u : 0u; // (slot 0: a document separator, no token)
						if (cp < end) {
							const uint32_t q = b0 + cp; // the next token starts here
							if (q >= cut) {
								if (ntok) atomicAdd(&W.tg_base[sidx], ntok);
								ntok = 0;
								do sidx++;
								while (W.pbeg[sidx + 1] <= q);
								cut = W.pbeg[sidx + 1];
							}
							mark(q);
						} else {
							active = false;
							if (ntok) atomicAdd(&W.tg_base[sidx], ntok);
							if (wmask) atomicOr(&W.tg_starts[wword], wmask);
							wmask = 0, ntok = 0;
						}
					}
				}
				act = __builtin_amdgcn_ballot_w64(active);
				if (act == 0) break;
			}
		}
	}
}

// count_ends_kernel's round loop in gfx950 assembly: scan_ranges_kernel's (TOK_RANGES_ASM: same ring
// rows, same refill, same pipelining -- a lookup's reads are issued before the previous one's entry
// is logged, and no lookup waits for a byte: the next byte is the one at cp + 1 (v30) or the one at
// p + 1 (v31, kept from the window's first lookup)) with the rules of a window walk instead of a
// greedy chain.  p = window start, cp = byte looked up, ncp = code points matched, last = deepest slot hit.
// Masks: s[40:41] lead (the byte starts a code point or is a separator), s[42:43] first (cp == p),
// s[44:45] separator, s[46:47] max_cp code points matched, s[48:49] the slot has children,
// s[50:51] skip (no window starts here), s[52:53] stop / skip or stop, vcc label match -> hit -> goes on.
#define TOK_COUNT_ASM(K, HIGH)                                                               \
	asm volatile(                                                                             \
	    "s_mov_b64 s[38:39], exec\n"                                                          \
	    "s_mov_b64 %[act], %[actin]\n"                                                        \
	    "s_mov_b64 %[req], %[reqin]\n"                                                        \
	    "s_mov_b32 %[cur], %[curin]\n"                                                        \
	    "s_mov_b32 %[np], 0\n"                                                                \
	    "s_mov_b64 exec, %[act]\n"                                                            \
	    "v_add_u32 v36, 1, %[p]\n"                                                            \
	    TOK_RING_BYTE_ADDR("v25", "%[cp]")                                                    \
	    TOK_RING_BYTE_ADDR("v26", "v36")                                                      \
	    "ds_read_u8 v24, v25\n"                                                               \
	    "ds_read_u8 v31, v26\n"                                                               \
	    "s_waitcnt lgkmcnt(0)\n"                                                              \
	    TOK_ISSUE_LOOKUP                                                                      \
	    ".Lround%=:\n"                                                                        \
	    "s_cmp_eq_u64 %[act], 0\n"                                                            \
	    "s_cbranch_scc1 .Lout%=\n"                                                            \
	    "s_cmp_gt_u32 %[cur], %[high]\n"                                                      \
	    "s_cbranch_scc1 .Lout%=\n"                                                            \
	    "s_waitcnt vmcnt(4)\n"                        /* (in issue order: the piece came before the round's >= 4 stores) */ \
	    "s_mov_b64 exec, %[req]\n"                                                            \
	    "v_and_b32 v34, %[rmask], %[fill]\n"                                                  \
	    "v_lshl_add_u32 v34, v34, 6, %[rbase]\n"                                              \
	    "ds_write2st64_b32 v34, %[d0], %[d1] offset1:1\n"                                     \
	    "v_add_u32 %[fill], 8, %[fill]\n"                                                     \
	    "s_mov_b64 exec, %[act]\n"                                                            \
	    "v_sub_u32 v34, %[fill], %[p]\n"                                                      \
	    "v_cmp_ge_u32 vcc, %[room], v34\n"            /* room for another piece (nothing before p is read again) */ \
	    "s_mov_b64 %[req], vcc\n"                                                             \
	    "s_mov_b64 exec, vcc\n"                                                               \
	    "v_lshl_add_u32 v34, %[fill], 6, %[tb]\n"                                             \
	    "s_waitcnt lgkmcnt(0)\n"                                                              \
	    "global_load_dword %[d0], v34, %[tprep]\n"                                            \
	    "global_load_dword %[d1], v34, %[tprep] offset:4\n"                                   \
	    "s_mov_b64 exec, %[act]\n"                                                            \
	    ".rept " #K "\n"                                                                      \
	    TOK_PRIO_HIGH                                                                         \
	    "s_waitcnt lgkmcnt(0)\n"                                                              \
	    "v_xor_b32 v27, v27, v24\n"                                                           \
	    "v_cmp_gt_u32 s[40:41], %[fcont], v24\n"      /* lead                              */ \
	    "v_cmp_eq_u32 s[42:43], %[cp], %[p]\n"        /* first                             */ \
	    "v_cmp_eq_u32 s[44:45], 0, v24\n"             /* separator                         */ \
	    "v_cmp_eq_u32 s[46:47], %[maxcp], %[ncp]\n"   /* the window is full                */ \
	    "v_and_b32 v26, 0x7f, v27\n"                                                          \
	    "v_cmp_eq_u32 vcc, 0, v26\n"                  /* the label matches                 */ \
	    "v_cmp_ne_u32 s[48:49], 0, v28\n"             /* the slot has children             */ \
	    "s_orn2_b64 s[50:51], s[44:45], s[40:41]\n"                                           \
	    "s_and_b64 s[50:51], s[50:51], s[42:43]\n"    /* skip = first and (separator or not lead) */ \
	    "s_or_b64 s[52:53], s[44:45], s[46:47]\n"                                             \
	    "s_and_b64 s[52:53], s[52:53], s[40:41]\n"                                            \
	    "s_andn2_b64 s[52:53], s[52:53], s[42:43]\n"  /* stop = not first and lead and (separator or full) */ \
	    "s_or_b64 s[52:53], s[52:53], s[50:51]\n"                                             \
	    "s_andn2_b64 vcc, vcc, s[52:53]\n"            /* hit                               */ \
	    "v_cndmask_b32 v31, v31, v30, s[42:43]\n"     /* v31 = the byte at p + 1           */ \
	    "v_cndmask_b32_e32 %[last], %[last], v25, vcc\n"                                      \
	    "s_and_b64 s[46:47], vcc, s[40:41]\n"         /* hit on a code point's first byte  */ \
	    "s_and_b64 vcc, vcc, s[48:49]\n"              /* the walk goes on                  */ \
	    "v_addc_co_u32 v26, s[54:55], 0, %[ncp], s[46:47]\n"                                  \
	    "v_add_u32 v36, 1, %[p]\n"                                                            \
	    "v_cndmask_b32_e32 %[ncp], 0, v26, vcc\n"                                             \
	    "v_cndmask_b32_e32 %[nb], 0, v28, vcc\n"                                              \
	    "v_cndmask_b32_e32 v24, v31, v30, vcc\n"      /* the next byte looked up           */ \
	    "v_cndmask_b32_e32 %[p], v36, %[p], vcc\n"    /* the window ends: the next one starts one byte on */ \
	    "v_cndmask_b32_e32 %[cp], %[p], v29, vcc\n"                                           \
	    TOK_ISSUE_LOOKUP                                                                      \
	    TOK_PRIO_LOW                                                                          \
	    "s_andn2_b64 exec, exec, vcc\n"               /* lanes whose window (or skipped byte) is done */ \
	    "s_mov_b64 s[56:57], exec\n"                                                          \
	    "s_andn2_b64 exec, exec, s[50:51]\n"          /* ... and has an entry to log       */ \
	    "v_mbcnt_lo_u32_b32 v33, exec_lo, 0\n"                                                \
	    "v_mbcnt_hi_u32_b32 v33, exec_hi, v33\n"                                              \
	    "v_lshl_add_u32 v33, v33, 1, %[cur]\n"                                                \
	    "global_store_short v33, %[last], %[sbase]\n"                                         \
	    "s_bcnt1_i32_b64 s40, exec\n"                                                         \
	    "s_lshl1_add_u32 %[cur], s40, %[cur]\n"                                               \
	    "s_add_u32 %[np], %[np], s40\n"                                                       \
	    "s_mov_b64 exec, s[56:57]\n"                                                          \
	    "v_cmp_ge_u32 vcc, %[p], %[hi]\n"             /* range done                        */ \
	    "s_andn2_b64 %[act], %[act], vcc\n"                                                   \
	    "s_mov_b64 exec, %[act]\n"                                                            \
	    ".endr\n"                                                                             \
	    "s_branch .Lround%=\n"                                                                \
	    ".Lout%=:\n"                                                                          \
	    "s_waitcnt vmcnt(0) lgkmcnt(0)\n"                                                     \
	    "s_mov_b64 exec, s[38:39]\n"                                                          \
	    : [p] "+v"(p), [cp] "+v"(cp), [nb] "+v"(nb), [last] "+v"(last), [ncp] "+v"(ncp), [fill] "+v"(fill),  \
	      [d0] "+v"(data0), [d1] "+v"(data1), [act] "=&s"(act_out), [req] "=&s"(req_out), [cur] "=&s"(cur_out), [np] "=&s"(np_out) \
	    : [actin] "s"(uni64(act)), [reqin] "s"(uni64(req)), [curin] "s"(uni32(log.cur)), [sbase] "s"(uni64((unsigned long long)(uintptr_t)log.base)), \
	      [hi] "v"(end), [rbase] "v"(ring_a), [tb] "v"(tb), [rmask] "s"(RING - 1), [rrows] "s"(RING - 4), [basea] "s"(label_bytes), \
	      [high] "s"(HIGH), [room] "s"(RING - 8), [tprep] "s"(W.tprep), [fcont] "s"(first_cont), [maxcp] "s"(max_cp) \
	    : "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "s38", "s39", "s40", "s41", "s42", "s43", \
	      "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "vcc", "scc", "memory")

// ---- closed-vocabulary occurrence counting (row f1) on the stream image --------------------------
// What ExtractCandidates + Trie::AddString add up (GetTokens.cpp:21-27, Trie.cpp:114-128): every code
// point position of every document starts a window of up to max_cp code points, and every
// vocabulary string that is a prefix of the window gets +1.  All those prefixes lie on ONE trie
// path, so a position only has to say where its walk ENDED -- one stream entry per position instead
// of one increment per prefix (4.2 per position on the bench shard) -- and the host adds every
// slot's ends to its ancestors afterwards (VocabTrie::AccumulateEnds, 38 k nodes).  The corpus is
// the prepared one WITHOUT lower-casing (this path matches bytes as they are), ranges, rings and
// log as in scan_ranges_kernel.  A walk starts at every byte that starts a code point
// (code < first_cont: trie.h) and is no separator; it stops at a miss, a leaf, a separator, or when
// a (max_cp + 1)-th code point would start.  No range needs its neighbour: nothing to repair.
template <uint32_t RING, uint32_t K, bool ASM> // K: lookups per round (W.round), unrolled; ASM: TOK_COUNT_ASM, else the C++ loop
__global__ void __launch_bounds__(1024) count_ends_kernel(StreamTrieDev T, StreamWork W, StreamDev Qd, uint32_t first_cont,
                                                          uint32_t max_cp, unsigned long long *positions, uint32_t counter_at) {
	extern __shared__ __attribute__((aligned(16))) unsigned char count_ends_smem[];
	// (no static LDS: the assembly takes a slot index for the LDS address of its label, so the image
	// starts at LDS address 0 and the workgroup's position counter sits behind the rings)
	uint32_t &s_positions = *reinterpret_cast<uint32_t *>(count_ends_smem + counter_at);
	LDS_AS unsigned char *lds = (LDS_AS unsigned char *)count_ends_smem;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = uni32(threadIdx.x >> 6);
	const uint32_t waves = blockDim.x >> 6;
	const uint32_t label_bytes = (T.nslots + 15) & ~15u;
	const uint32_t info_bytes = (T.nslots * 2 + 15) & ~15u;
	{
		LDS_AS u32x4 *dst = (LDS_AS u32x4 *)lds;
		const u32x4 *src = reinterpret_cast<const u32x4 *>(T.lt);
		for (uint32_t i = threadIdx.x; i < label_bytes / 16; i += blockDim.x) dst[i] = src[i];
		dst = (LDS_AS u32x4 *)(lds + label_bytes);
		src = reinterpret_cast<const u32x4 *>(T.base);
		for (uint32_t i = threadIdx.x; i < info_bytes / 16; i += blockDim.x) dst[i] = src[i];
		if (threadIdx.x == 0) s_positions = 0;
	}
	__syncthreads();
	const LDS_AS uint8_t *l_lt = (const LDS_AS uint8_t *)lds;
	const LDS_AS uint16_t *l_base = (const LDS_AS uint16_t *)(lds + label_bytes);
	if (ASM && (uint32_t)(uintptr_t)lds != 0) { // (as scan_ranges_kernel: reported, never ignored)
		if (threadIdx.x == 0) Qd.cursor[1] = 1u;
		return;
	}
	// ring rows of one dword per lane, as in scan_ranges_kernel
	LDS_AS unsigned char *ring = lds + ((label_bytes + info_bytes + 127) & ~127u) + wave * (64 * RING) + lane * 4;
	auto ring_at = [&](uint32_t pos) { return ring + ((pos & (RING - 4)) << 6) + (pos & 3u); };
	DirectLog log;
	log.open_own(Qd, blockIdx.x * waves + wave);
	const uint32_t total_waves = gridDim.x * waves;
	const uint32_t rows = W.range_bytes + W.halo;
	constexpr uint32_t high = kBatchEntries * 2 - 128 * K; // a round adds at most 64 K entries
	uint32_t npos = 0, npos_wave = 0; // per lane (C++ loop) / per wave (assembly)
	for (uint32_t wb = blockIdx.x * waves + wave; wb * 64 < W.nranges; wb += total_waves) {
		const uint32_t g = wb * 64 + lane;
		const uint8_t *blk = W.tprep + (size_t)wb * rows * 64 + lane * 8;
		uint32_t end = 0;
		if (g < W.nranges) {
			const uint32_t b = g * W.range_bytes;
			end = b + W.range_bytes < W.prep_total ? W.range_bytes : W.prep_total - b;
		}
		bool active = end != 0;
		uint32_t p = 0, d = 0, nb = 0, last = 0, ncp = 0, fill = 0;
		wave_sync();
		if (active) {
#pragma unroll
			for (uint32_t k = 0; k < RING / 8; k++) {
				const u32x2 x = *reinterpret_cast<const u32x2 *>(blk + (size_t)fill * 64);
				*(LDS_AS uint32_t *)ring_at(fill) = x.x;
				*(LDS_AS uint32_t *)ring_at(fill + 4) = x.y;
				fill += 8;
			}
		}
		wave_sync();
		if constexpr (ASM) {
			const uint32_t ring_a = (uint32_t)(uintptr_t)ring;
			const uint32_t tb = (uint32_t)((size_t)wb * rows * 64 + lane * 8); // (the transposed copy is below 4 GiB)
			unsigned long long act = __builtin_amdgcn_ballot_w64(active), req = 0;
			uint32_t data0 = 0, data1 = 0, cp = 0;
			while (act != 0) {
				unsigned long long act_out, req_out;
				uint32_t cur_out, np_out;
				if constexpr (K == 4) TOK_COUNT_ASM(4, high);
				else if constexpr (K == 5) TOK_COUNT_ASM(5, high);
				else if constexpr (K == 6) TOK_COUNT_ASM(6, high);
				else if constexpr (K == 7) TOK_COUNT_ASM(7, high);
				else TOK_COUNT_ASM(8, high);
				act = act_out, req = req_out;
				log.cur = cur_out;
				npos_wave += np_out;
				log.roll_if_full(Qd, lane, high);
			}
			continue;
		}
		bool req = false;
		u32x2 data = {0, 0};
		unsigned long long act = __builtin_amdgcn_ballot_w64(active);
		while (act != 0) {
			if (req) {
				*(LDS_AS uint32_t *)ring_at(fill) = data.x;
				*(LDS_AS uint32_t *)ring_at(fill + 4) = data.y;
				fill += 8;
			}
			req = active && (fill - p) <= RING - 8; // (nothing before p is read again)
			if (req) data = *reinterpret_cast<const u32x2 *>(blk + (size_t)fill * 64);
			wave_sync();
#pragma unroll
			for (uint32_t it = 0; it < K; it++) {
				bool ended = false;
				if (active) {
					const uint32_t c = *ring_at(p + d);
					const bool lead = c < first_cont; // a code point (or the separator) starts here
					if (d == 0 && (!lead || c == 0)) {
						p++; // no window starts inside a code point or on a separator
						active = p < end;
					} else {
						const bool stop = d != 0 && lead && (c == 0 || ncp == max_cp);
						const uint32_t t = nb + c;
						const uint32_t x = l_lt[t] ^ c;
						const uint32_t b = l_base[t];
						const bool hit = !stop && (x & 0x7Fu) == 0;
						last = hit ? t : last;
						d += hit ? 1u : 0u;
						ncp += hit && lead ? 1u : 0u;
						nb = hit ? b : 0u;
						ended = nb == 0; // (the first byte of a window always hits: the root row is dense)
					}
				}
				log.put(ended, last);
				if (ended) {
					npos++;
					p++;
					d = 0, ncp = 0;
					active = p < end;
				}
			}
			log.roll_if_full(Qd, lane, high);
			act = __builtin_amdgcn_ballot_w64(active);
		}
	}
	log.close(Qd, lane);
	for (int o = 32; o > 0; o >>= 1) npos += __shfl_xor(npos, o, 64);
	npos += npos_wave;
	if (lane == 0 && npos) atomicAdd(&s_positions, npos);
	__syncthreads();
	if (threadIdx.x == 0 && s_positions) atomicAdd(positions, (unsigned long long)s_positions);
}

// tprep block wb, piece k, lane l = the 8 bytes prep[(wb * 64 + l) * range_bytes + 8 k ...] (zero past
// what is readable of the linear copy).  One thread per piece.
__global__ void __launch_bounds__(256) transpose_ranges_kernel(const uint8_t *prep, uint64_t prep_readable, uint32_t range_bytes,
                                                               uint32_t halo, uint32_t nblocks, uint8_t *tprep) {
	const uint64_t pieces = ((uint64_t)range_bytes + halo) / 8; // per range
	const uint64_t n = (uint64_t)nblocks * pieces * 64;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t wb = i / (pieces * 64);
		const uint64_t rem = i - wb * pieces * 64;
		const uint64_t k = rem / 64;
		const uint64_t l = rem % 64;
		const uint64_t src = (wb * 64 + l) * range_bytes + 8 * k;
		uint32_t w[2] = {0, 0};
		for (uint32_t j = 0; j < 8; j++) {
			const uint32_t c = src + j < prep_readable ? prep[src + j] : 0u;
			w[j >> 2] |= c << (8 * (j & 3));
		}
		reinterpret_cast<uint2 *>(tprep)[i] = uint2{w[0], w[1]};
	}
}

// ---- range entries: fix-up ------------------------------------------------------------------
// The protocol of kernels_generic.hip's fix-up (fixup_parallel_kernel / fixup_ordered_kernel), on
// the prepared corpus and the stream image read from global memory.  Range g counted the chain
// that starts at g * R + entry[g]; the true chain enters it at g * R + exit[g - 1].  Two greedy
// chains over the same bytes coincide from the first position they share, so the repair walks
// whichever chain is behind -- the true one counted +1, the counted one -1 -- until they meet or
// both have left the range; a repair that never met changes the range's exit, and the ordered
// kernel then redoes the ranges after it one by one.  entry[] is left holding the true entries.
struct FixDev {
	StreamTrieDev T;
	const uint32_t *slot_id;
	const uint8_t *prep;
	uint32_t prep_total, range_bytes, nranges;
	uint32_t *entry, *exit_a, *exit_b;
	uint32_t *changed; // [0] set when some repair moved an exit, [1] workgroups done
	uint32_t epoch;    // 1 .. 65535, another one every evaluation: exit_b[g] counts only when tagged with it
	unsigned long long *out;
	uint32_t total_index;
	uint32_t blocks; // workgroups that take part (the first ones of the launch)
};

// one token of the chain at pos: its length and the slot it ends on (the root row is dense and
// all token ends, so len >= 1; separators stop every walk)
__device__ __forceinline__ void walk_token_image(const FixDev &F, uint32_t pos, uint32_t &len, uint32_t &slot) {
	uint32_t nb = 0, d = 0; // the root's base is 0
	len = 1;
	slot = 0;
	do {
		const uint32_t c = F.prep[pos + d];
		const uint32_t t = nb + c;
		const uint32_t x = F.T.lt[t] ^ c;
		if ((x & 0x7Fu) != 0) break;
		d++;
		if (x == 0x80u) len = d, slot = t;
		nb = F.T.base[t];
	} while (nb != 0);
}

__device__ __forceinline__ void count_slot(const FixDev &F, uint32_t slot, bool add) {
	const uint32_t id = F.slot_id[slot];
	if (id == 0xFFFFFFFFu) return; // separator
	const unsigned long long v = add ? 1ull : ~0ull;
	atomicAdd(&F.out[id], v);
	atomicAdd(&F.out[F.total_index], v);
}

// true when the chains met (exit unchanged); *pa_out = where the true chain left the range
__device__ bool dual_walk_image(const FixDev &F, uint32_t end, uint32_t pa, uint32_t pb, uint32_t *pa_out) {
	while (pa != pb && (pa < end || pb < end)) {
		uint32_t len, slot;
		if (pa < pb) {
			walk_token_image(F, pa, len, slot);
			count_slot(F, slot, true);
			pa += len;
		} else {
			walk_token_image(F, pb, len, slot);
			count_slot(F, slot, false);
			pb += len;
		}
	}
	*pa_out = pa;
	return pa == pb;
}

__device__ void fixup_ordered_pass(const FixDev &F);

// The whole fix-up, for the workgroups blockIdx.x < F.blocks of a launch (every thread of them
// reaches every barrier).  A kernel of its own (fixup_parallel_stream_kernel) or the opening of
// hist_slots_kernel: with an unchanged vocabulary it only compares, and a launch for that costs
// more than the comparing (7 us of a 125 us evaluation).
__device__ void fixup_stream_body(const FixDev &F) {
	__shared__ uint32_t s_last, s_wrote;
	if (threadIdx.x == 0) s_wrote = 0;
	__syncthreads();
	for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < F.nranges; g += F.blocks * blockDim.x) {
		uint32_t ex = F.exit_a[g];
		(void)ex;
		if (g > 0) {
			const uint32_t want = F.exit_a[g - 1];
			const uint32_t have = F.entry[g];
			if (want != have) {
				const uint32_t begin = g * F.range_bytes;
				const uint32_t end = begin + F.range_bytes < F.prep_total ? begin + F.range_bytes : F.prep_total;
				uint32_t pa;
				if (!dual_walk_image(F, end, begin + want, begin + have, &pa)) {
					ex = pa - end;
					atomicOr(&F.changed[0], 1u);
				}
				F.entry[g] = want;
				F.exit_b[g] = F.epoch << 16 | ex; // (an overshoot is shorter than a token: 16 bits)
				s_wrote = 1u;
			}
		}
	}
	// The ordered pass (rare: only when a repair moved an exit) is done by whichever workgroup
	// finishes last, in the same launch: a launch of its own costs more than a normal evaluation's
	// whole fix-up.  changed[1] counts the workgroups that are done.
	// (a fence costs microseconds: only a workgroup that repaired something has anything to publish,
	// and exit_b is only read by the ordered pass, which runs only after some repair)
	__syncthreads();
	if (threadIdx.x == 0) {
		if (s_wrote) __threadfence();
		s_last = atomicAdd(&F.changed[1], 1u) == F.blocks - 1 ? 1u : 0u;
	}
	__syncthreads();
	if (!s_last) return;
	fixup_ordered_pass(F);
}

__global__ void __launch_bounds__(256) fixup_parallel_stream_kernel(FixDev F) { fixup_stream_body(F); }

// ONE workgroup; every thread reaches every barrier.  Does nothing unless a repair moved an exit.
// exit of range g after the repairs so far: exit_b where this evaluation wrote one (published by
// the writer's fence), else what the scan left
__device__ __forceinline__ uint32_t exit_now(const FixDev &F, uint32_t g) {
	const uint32_t b = F.exit_b[g];
	return (b >> 16) == F.epoch ? (b & 0xFFFFu) : F.exit_a[g];
}

__device__ void fixup_ordered_pass(const FixDev &F) {
	__shared__ uint32_t s_first;
	if (__hip_atomic_load(&F.changed[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return; // uniform
	__threadfence(); // the other workgroups' exit_b / entry
	for (uint32_t base = 1; base < F.nranges; base += blockDim.x) {
		for (;;) {
			const uint32_t g = base + threadIdx.x;
			const bool need = g < F.nranges && exit_now(F, g - 1) != F.entry[g];
			if (threadIdx.x == 0) s_first = 0xFFFFFFFFu;
			__syncthreads();
			if (need) atomicMin(&s_first, g);
			__syncthreads();
			const uint32_t first = s_first;
			if (first == 0xFFFFFFFFu) break;
			if (g == first) {
				const uint32_t want = exit_now(F, g - 1);
				const uint32_t have = F.entry[g];
				const uint32_t begin = g * F.range_bytes;
				const uint32_t end = begin + F.range_bytes < F.prep_total ? begin + F.range_bytes : F.prep_total;
				uint32_t pa;
				if (!dual_walk_image(F, end, begin + want, begin + have, &pa)) F.exit_b[g] = F.epoch << 16 | (pa - end);
				F.entry[g] = want;
				__threadfence();
			}
			__syncthreads();
		}
		__syncthreads();
	}
}

// a slot's count goes to the id the slot counts for; entries on slots nobody counts (document
// separators) leave the token total again (main.cpp:41 counts the ids Tokenize returned, not the
// separators this layout adds).  No map: counts per slot.
// pool != nullptr: slot_id[] holds events of the automaton scan (dfa.h): an id with the top bit set is a
// list of ids in pool, each of which the entry counts for (the entry itself was added to the total once)
__device__ __forceinline__ void book_slot(uint32_t s, long long sum, const uint32_t *slot_id, unsigned long long *out,
                                          uint32_t total_index, const uint32_t *pool = nullptr) {
	const uint32_t id = slot_id ? slot_id[s] : s;
	if (id == 0xFFFFFFFFu) atomicAdd(&out[total_index], (unsigned long long)(-sum));
	else if (pool && (id & 0x80000000u)) {
		const uint32_t o = id & 0x7FFFFFFFu, len = pool[o];
		for (uint32_t i = 0; i < len; i++) atomicAdd(&out[pool[o + 1 + i]], (unsigned long long)sum);
		atomicAdd(&out[total_index], (unsigned long long)(sum * ((long long)len - 1)));
	} else atomicAdd(&out[id], (unsigned long long)sum);
}
__host__ __device__ __forceinline__ uint32_t slab_row_stride(uint32_t nslots) { return (nslots + 1) & ~1u; } // int16 units

// ---- token stream (slot indices) -> slab rows ----------------------------------------------
// grid = (slab_blocks, slot ranges).  As hist_kernel (kernels_tile.hip), for u16 slot entries
// (0xFFFF = null, no sign bit: a whole pass either adds or, with flip, subtracts).
// The loop is bound by the instructions it issues (23 per entry with a test and a branch for
// null, a range test and a count of the entries seen: 14 us of the kernel's 20 on the bench
// shard), so with one slot range there is none of that per entry: the index is min(entry,
// width + lane), which sends a null entry to a spare word of the lane's own, and the entries seen
// are the entries looked at minus what the spare words collected.
template <bool ONE_RANGE>
__global__ void __launch_bounds__(1024) hist_slots_kernel(StreamDev Qd, uint32_t nslots, uint32_t range_slots, int16_t *slab,
                                                          uint32_t flip, const uint32_t *slot_id, unsigned long long *out,
                                                          uint32_t total_index, FixDev F, const uint32_t *pool) {
	extern __shared__ __attribute__((aligned(16))) unsigned char hist_smem[];
	uint32_t *bins = reinterpret_cast<uint32_t *>(hist_smem);
	__shared__ int s_total;
	// a full evaluation's fix-up rides along (F.blocks != 0): it and the histogram only meet in `out`,
	// which both add to with atomics
	if (F.blocks != 0 && blockIdx.y == 0 && blockIdx.x < F.blocks) fixup_stream_body(F);
	const uint32_t lo = ONE_RANGE ? 0u : blockIdx.y * range_slots;
	const uint32_t hi = lo + range_slots < nslots ? lo + range_slots : nslots;
	const uint32_t width = hi - lo;
	for (uint32_t i = threadIdx.x; i < width + 64; i += blockDim.x) bins[i] = 0;
	if (threadIdx.x == 0) s_total = 0;
	__syncthreads();
	uint32_t batches = Qd.cursor[0];
	if (batches > Qd.capacity_batches) batches = Qd.capacity_batches;
	constexpr uint32_t vecs_per_block = kLogEntries * 2 / 16; // 64: one 1 KiB row per block
	constexpr int kRowsInFlight = 8;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t n_waves = gridDim.x * (blockDim.x / 64);
	const uint32_t wave_id = uni32(blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6));
	const uint4 *vec = reinterpret_cast<const uint4 *>(Qd.blocks);
	const uint32_t sgn = flip ? 0xFFFFFFFFu : 1u;
	const uint32_t spare = width + lane;
	uint32_t seen = 0; // ONE_RANGE: entries looked at, null or not
	for (uint32_t bt = wave_id; bt < batches; bt += n_waves) {
		uint32_t live = Qd.fill[bt]; // entries written to the batch (what follows them was never written)
		if (live > kBatchEntries) live = kBatchEntries;
		live = uni32(live);
		const uint32_t live_rows = (live + 511) / 512;
		const uint4 *rows = vec + (uint64_t)bt * kStreamBatch * vecs_per_block + lane;
		for (uint32_t r0 = 0; r0 < live_rows; r0 += kRowsInFlight) {
			uint4 v[kRowsInFlight];
#pragma unroll
			for (int u = 0; u < kRowsInFlight; u++) {
				v[u] = uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
				if (r0 + u < live_rows) {
					v[u] = rows[(size_t)(r0 + u) * 64];
					const uint32_t first = (r0 + u) * 512 + lane * 8; // this lane's eight entries of the row
					if (first + 8 > live) { // the batch's last, partly written vector: blank what lies past `live`
						uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
						for (uint32_t q = 0; q < 8; q++)
							if (first + q >= live) w[q >> 1] |= 0xFFFFu << (16 * (q & 1));
						v[u] = uint4{w[0], w[1], w[2], w[3]};
					}
				}
			}
			if constexpr (ONE_RANGE) {
				seen += 8 * kRowsInFlight;
#pragma unroll
				for (int u = 0; u < kRowsInFlight; u++) {
					const uint32_t wds[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
					for (int k = 0; k < 4; k++) {
						const uint32_t e0 = wds[k] & 0xFFFFu, e1 = wds[k] >> 16;
						atomicAdd(&bins[e0 < spare ? e0 : spare], sgn);
						atomicAdd(&bins[e1 < spare ? e1 : spare], sgn);
					}
				}
			} else {
#pragma unroll
				for (int u = 0; u < kRowsInFlight; u++) {
					const uint32_t wds[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
					for (int k = 0; k < 4; k++) {
						if (wds[k] == 0xFFFFFFFFu) continue;
#pragma unroll
						for (int h = 0; h < 2; h++) {
							const uint32_t e = (wds[k] >> (16 * h)) & 0xFFFFu;
							if (e == 0xFFFFu) continue;
							seen++;
							if (e >= lo && e < hi) atomicAdd(&bins[e - lo], sgn);
						}
					}
				}
			}
		}
	}
	if (blockIdx.y == 0) {
		int total = flip ? -(int)seen : (int)seen;
		for (int o = 32; o > 0; o >>= 1) total += __shfl_down(total, o, 64);
		if ((threadIdx.x & 63) == 0 && total != 0) atomicAdd(&s_total, total);
	}
	__syncthreads();
	if (ONE_RANGE && threadIdx.x < 64) { // the null entries were looked at, not seen (flip: both counts are negative)
		int nulls = (int)bins[width + threadIdx.x];
		for (int o = 32; o > 0; o >>= 1) nulls += __shfl_down(nulls, o, 64);
		if (threadIdx.x == 0) s_total -= nulls;
	}
	// The slab row is 16 bits per slot (half the bytes written here and read by the reduction: at
	// 37.8 k slots and 256 workgroups 2 x 19 MB per evaluation).  A bin that does not fit -- one
	// slot with more than 32,767 of a workgroup's entries -- goes to the output directly, as the
	// reduction would have booked it; so does the workgroup's entry count.
	int16_t *row = slab + (size_t)blockIdx.x * slab_row_stride(nslots);
	for (uint32_t i = threadIdx.x; i < width; i += blockDim.x) {
		const int v = (int)bins[i];
		const bool fits = v >= -32768 && v <= 32767;
		row[lo + i] = fits ? (int16_t)v : (int16_t)0;
		if (!fits) book_slot(lo + i, (long long)v, slot_id, out, total_index, pool);
	}
	__syncthreads();
	if (blockIdx.y == 0 && threadIdx.x == 0 && s_total != 0) atomicAdd(&out[total_index], (unsigned long long)(long long)s_total);
}

// Slab rows (int16 partials per slot) -> out: counts[slot_id[s]] += column sums (book_slot).
// One workgroup = 128 columns x 16 row groups: wave w sums the rows r = w, w + 16, ... of its 128
// columns (two to a lane, 256 contiguous bytes per row), the partials meet in LDS, and wave 0 adds
// each column's total to the id its slot counts for -- one scattered 64-bit atomic per slot and
// evaluation.
constexpr uint32_t kMapReduceGroups = 16;
__global__ void __launch_bounds__(1024) slab_reduce_map_kernel(const int16_t *slab, uint32_t rows, uint32_t nslots,
                                                               const uint32_t *slot_id, unsigned long long *out,
                                                               uint32_t total_index, uint32_t *cursor, uint32_t next_first_batches,
                                                               const uint32_t *pool) {
	__shared__ int part[kMapReduceGroups][64][2];
	// (the histogram has read cursor[0], the fix-up is done with its two words: ready for the next evaluation)
	if (next_first_batches != 0 && blockIdx.x == 0 && threadIdx.x == 0) cursor[0] = next_first_batches, cursor[2] = 0u, cursor[3] = 0u;
	const uint32_t lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
	const uint32_t s = blockIdx.x * 128 + lane * 2;
	const size_t stride = slab_row_stride(nslots) / 2; // in 32-bit words
	const uint32_t *slab2 = reinterpret_cast<const uint32_t *>(slab);
	int lo = 0, hi = 0; // (rows x 32,767 fits easily)
	if (s < nslots) {
		uint32_t r = grp;
		for (; r + 3 * kMapReduceGroups < rows; r += 4 * kMapReduceGroups) {
			const uint32_t a0 = slab2[(size_t)r * stride + s / 2];
			const uint32_t a1 = slab2[(size_t)(r + kMapReduceGroups) * stride + s / 2];
			const uint32_t a2 = slab2[(size_t)(r + 2 * kMapReduceGroups) * stride + s / 2];
			const uint32_t a3 = slab2[(size_t)(r + 3 * kMapReduceGroups) * stride + s / 2];
			lo += (int)(int16_t)a0 + (int)(int16_t)a1 + (int)(int16_t)a2 + (int)(int16_t)a3;
			hi += ((int)a0 >> 16) + ((int)a1 >> 16) + ((int)a2 >> 16) + ((int)a3 >> 16);
		}
		for (; r < rows; r += kMapReduceGroups) {
			const uint32_t a = slab2[(size_t)r * stride + s / 2];
			lo += (int)(int16_t)a;
			hi += (int)a >> 16;
		}
	}
	part[grp][lane][0] = lo;
	part[grp][lane][1] = hi;
	__syncthreads();
	if (grp != 0 || s >= nslots) return;
	for (uint32_t g = 1; g < kMapReduceGroups; g++) lo += part[g][lane][0], hi += part[g][lane][1];
	if (lo != 0) book_slot(s, lo, slot_id, out, total_index, pool);
	if (hi != 0 && s + 1 < nslots) book_slot(s + 1, hi, slot_id, out, total_index, pool);
}

} // namespace

// ---- host side -----------------------------------------------------------------------------

void launch_prep_corpus(const CorpusDev &C, const ByteCodes &M, uint8_t *prep, uint64_t prep_total, uint32_t *flags,
                        hipStream_t stream, bool lower) {
	if (prep_total == 0 || C.D == 0) return;
	const uint64_t nvec = (prep_total + 15) / 16;
	const uint32_t blocks = (uint32_t)std::min<uint64_t>((nvec + 255) / 256, 1u << 16);
	hipLaunchKernelGGL(prep_corpus_kernel, dim3(blocks), dim3(256), 0, stream, C, M, prep, prep_total, flags, lower ? 1u : 0u);
}

bool plan_stream_scan(uint32_t nslots, uint64_t max_len, uint32_t num_cus, StreamScanPlan *plan) {
	uint32_t ring = 0;
	if (max_len + 2 * kRound + 14 <= 64) ring = 64;
	else if (max_len + 2 * kRound + 14 <= 128) ring = 128;
	else return false;
	const uint32_t label_bytes = (nslots + 15) & ~15u;
	const uint32_t info_bytes = (nslots * 2 + 15) & ~15u;
	const uint32_t wave_bytes = 64 * ring + kLogEntries * 2;
	const uint32_t lds_total = 160 * 1024 - 64; // hist_slots_kernel's static word and slack
	if (label_bytes + info_bytes + wave_bytes > lds_total) return false;
	uint32_t waves = (lds_total - label_bytes - info_bytes) / wave_bytes;
	if (waves > 16) waves = 16;
	if (const char *s = getenv("TOKSCORE_WAVES")) waves = std::min<uint32_t>(waves, (uint32_t)std::max(1, atoi(s)));
	if (waves < 4 && !getenv("TOKSCORE_WAVES")) return false; // too little of the CU left to hide the LDS latency: tile path
	plan->ring = ring;
	plan->waves_per_block = waves;
	plan->blocks = num_cus;
	plan->lds_bytes = label_bytes + info_bytes + waves * wave_bytes;
	return true;
}

void launch_transpose_ranges(const uint8_t *prep, uint64_t prep_readable, uint32_t range_bytes, uint32_t halo, uint32_t nblocks,
                             uint8_t *tprep, hipStream_t stream) {
	const uint64_t n = (uint64_t)nblocks * ((range_bytes + halo) / 8) * 64;
	if (n == 0) return;
	const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, 1u << 16);
	hipLaunchKernelGGL(transpose_ranges_kernel, dim3(blocks), dim3(256), 0, stream, prep, prep_readable, range_bytes, halo, nblocks,
	                   tprep);
}

bool plan_ranges_scan(uint32_t nslots, uint64_t max_len, uint32_t num_cus, StreamScanPlan *plan) {
	// RING >= 2 K + 8 + max_len - 2 with rounds of K <= 8 lookups (scan_ranges_kernel); a round
	// shorter than 4 spends too much on refills: the next ring size then
	uint32_t ring = 0, round = 0;
	for (uint32_t r : {32u, 64u, 128u}) {
		if (r < 6 + max_len + 2) continue;
		const uint32_t k = std::min<uint32_t>(8, (uint32_t)((r - 6 - max_len) / 2));
		if (k >= 4) {
			ring = r, round = k;
			break;
		}
	}
	if (!ring) return false;
	if (const char *s = getenv("TOKSCORE_RING")) {
		const uint32_t r = (uint32_t)atoi(s);
		if ((r == 64 || r == 128) && r > ring) ring = r, round = std::min<uint32_t>(8, (uint32_t)((r - 6 - max_len) / 2));
	}
	const uint32_t label_bytes = (nslots + 15) & ~15u;
	const uint32_t info_bytes = (nslots * 2 + 15) & ~15u;
	const uint32_t wave_bytes = 64 * ring;
	const uint32_t lds_total = 160 * 1024 - 64;
	const uint32_t image_bytes = (label_bytes + info_bytes + 127) & ~127u;
	if (image_bytes + 4 * wave_bytes > lds_total) return false;
	uint32_t waves = (lds_total - image_bytes) / wave_bytes;
	if (waves > 16) waves = 16;
	if (const char *s = getenv("TOKSCORE_WAVES")) waves = std::min<uint32_t>(waves, (uint32_t)std::max(1, atoi(s)));
	plan->ring = ring;
	plan->round = round;
	plan->halo = (ring + 2 * (uint32_t)max_len + 2 + 7) & ~7u;
	plan->waves_per_block = waves;
	plan->blocks = num_cus;
	plan->lds_bytes = image_bytes + waves * wave_bytes;
	return true;
}

void launch_scan_ranges(const StreamTrieDev &T, const StreamWork &W, const StreamDev &Q, const StreamScanPlan &plan,
                        hipStream_t stream, const OutInit &init, hipEvent_t ev_start, hipEvent_t ev_stop) {
	auto go = [&](auto kernel) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
		                          (int)plan.lds_bytes);
		if (ev_start && ev_stop)
			hipExtLaunchKernelGGL(kernel, dim3(plan.blocks), dim3(64 * plan.waves_per_block), plan.lds_bytes, stream, ev_start, ev_stop, 0u,
			                      T, W, Q, init);
		else hipLaunchKernelGGL(kernel, dim3(plan.blocks), dim3(64 * plan.waves_per_block), plan.lds_bytes, stream, T, W, Q, init);
	};
	static const bool use_asm = [] {
		const char *e = getenv("TOKSCORE_ASM");
		return !(e && atoi(e) == 0);
	}();
	if (use_asm) {
		auto pick = [&](auto ring) {
			constexpr uint32_t R = decltype(ring)::value;
			switch (plan.round) {
			case 4: go(&scan_ranges_kernel<R, 4>); break;
			case 5: go(&scan_ranges_kernel<R, 5>); break;
			case 6: go(&scan_ranges_kernel<R, 6>); break;
			case 7: go(&scan_ranges_kernel<R, 7>); break;
			default: go(&scan_ranges_kernel<R, 8>); break; // (plan_ranges_scan: 4 .. 8)
			}
		};
		if (plan.ring == 32) pick(std::integral_constant<uint32_t, 32>{});
		else if (plan.ring == 64) pick(std::integral_constant<uint32_t, 64>{});
		else pick(std::integral_constant<uint32_t, 128>{});
	} else {
		if (plan.ring == 32) go(&scan_ranges_kernel<32, 0>);
		else if (plan.ring == 64) go(&scan_ranges_kernel<64, 0>);
		else go(&scan_ranges_kernel<128, 0>);
	}
}

void launch_count_ends(const StreamTrieDev &T, const StreamWork &W, const StreamDev &Q, const StreamScanPlan &plan,
                       uint32_t first_cont, uint32_t max_cp, unsigned long long *positions, hipStream_t stream) {
	auto go = [&](auto kernel) {
		// (the workgroup's position counter sits in 16 bytes behind the plan's block)
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
		                          (int)plan.lds_bytes + 16);
		hipLaunchKernelGGL(kernel, dim3(plan.blocks), dim3(64 * plan.waves_per_block), plan.lds_bytes + 16, stream, T, W, Q,
		                   first_cont, max_cp, positions, plan.lds_bytes);
	};
	static const bool use_asm = [] {
		const char *e = getenv("TOKSCORE_ASM");
		return !(e && atoi(e) == 0);
	}();
	auto pick = [&](auto ring) {
		constexpr uint32_t R = decltype(ring)::value;
		if (use_asm) {
			switch (plan.round) {
			case 4: go(&count_ends_kernel<R, 4, true>); break;
			case 5: go(&count_ends_kernel<R, 5, true>); break;
			case 6: go(&count_ends_kernel<R, 6, true>); break;
			case 7: go(&count_ends_kernel<R, 7, true>); break;
			default: go(&count_ends_kernel<R, 8, true>); break; // (plan_ranges_scan: 4 .. 8)
			}
		} else {
			switch (plan.round) {
			case 4: go(&count_ends_kernel<R, 4, false>); break;
			case 5: go(&count_ends_kernel<R, 5, false>); break;
			case 6: go(&count_ends_kernel<R, 6, false>); break;
			case 7: go(&count_ends_kernel<R, 7, false>); break;
			default: go(&count_ends_kernel<R, 8, false>); break;
			}
		}
	};
	if (plan.ring == 32) pick(std::integral_constant<uint32_t, 32>{});
	else if (plan.ring == 64) pick(std::integral_constant<uint32_t, 64>{});
	else pick(std::integral_constant<uint32_t, 128>{});
}

void launch_scan_stream(const StreamTrieDev &T, const StreamWork &W, const StreamDev &Q, const StreamScanPlan &plan,
                        hipStream_t stream) {
	if (plan.ring == 64) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_stream_kernel<64>),
		                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes);
		hipLaunchKernelGGL(scan_stream_kernel<64>, dim3(plan.blocks), dim3(64 * plan.waves_per_block), plan.lds_bytes,
		                   stream, T, W, Q);
	} else {
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_stream_kernel<128>),
		                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes);
		hipLaunchKernelGGL(scan_stream_kernel<128>, dim3(plan.blocks), dim3(64 * plan.waves_per_block), plan.lds_bytes,
		                   stream, T, W, Q);
	}
}

void launch_find_toggle_streams(const ToggleBatch &B, const StreamWork &W, uint32_t npairs, uint32_t nchunks,
                                hipStream_t stream) {
	(void)hipMemsetAsync(B.n_items, 0, kToggleLists * sizeof(uint32_t), stream);
	if (B.n == 0 || npairs == 0) return;
	ToggleBatch Bw = B;
	Bw.chunk_first = W.chunk_first;
	Bw.pbeg = W.pbeg;
	hipLaunchKernelGGL(list_toggle_pairs_kernel, dim3(B.n), dim3(1024), 0, stream, Bw);
	hipLaunchKernelGGL(fill_toggle_pairs_kernel, dim3((npairs + 255) / 256), dim3(256), 0, stream, Bw, npairs);
	hipLaunchKernelGGL(find_toggle_streams_kernel, dim3((npairs + 15) / 16), dim3(1024), 0, stream, Bw, W, nchunks);
}

void launch_mark_ranges(const StreamTrieDev &T, const StreamWork &W, const StreamScanPlan &plan, hipStream_t stream) {
	if (W.nranges == 0 || W.nstreams == 0) return;
	auto go = [&](auto kernel) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
		                          (int)plan.lds_bytes);
		hipLaunchKernelGGL(kernel, dim3(plan.blocks), dim3(64 * plan.waves_per_block), plan.lds_bytes, stream, T, W);
	};
	if (plan.ring == 32) go(&mark_ranges_kernel<32>);
	else if (plan.ring == 64) go(&mark_ranges_kernel<64>);
	else go(&mark_ranges_kernel<128>);
}

void launch_count_streams(const StreamTrieDev &T, const StreamWork &W, const StreamScanPlan &plan, hipStream_t stream) {
	if (W.nstreams == 0) return;
	const uint32_t image = ((T.nslots + 15) & ~15u) + ((T.nslots * 2 + 15) & ~15u);
	const uint32_t lds_total = 160 * 1024 - 1024;
	const uint32_t waves = std::max<uint32_t>(1, std::min<uint32_t>(16, (lds_total - image) / (64 * plan.ring)));
	const uint32_t lds_bytes = image + waves * 64 * plan.ring;
	const uint64_t lanes_per_block = 64ull * waves;
	const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(plan.blocks, (W.nstreams + lanes_per_block - 1) / lanes_per_block));
	auto go = [&](auto kernel) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
		hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64 * waves), lds_bytes, stream, T, W);
	};
	if (plan.ring == 64) go(&count_streams_kernel<64>);
	else go(&count_streams_kernel<128>);
}

void launch_local_toggles(const StreamTrieDev &T, const StreamWork &W, uint32_t num_cus, uint64_t max_items, hipStream_t stream) {
	if (max_items == 0) return;
	const uint32_t image = ((T.nslots + 15) & ~15u) + ((T.nslots * 2 + 15) & ~15u);
	const uint32_t lds_total = 160 * 1024 - 6 * 1024; // (the kernel's static arrays and slack)
	const uint32_t window = image >= lds_total ? 0u : std::min<uint32_t>(kLocalWindow, ((lds_total - image) / 1024) & ~15u);
	const uint32_t lds_bytes = image + 1024 * window;
	const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(num_cus, (max_items + 1023) / 1024));
	(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&local_toggles_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
	                          (int)lds_bytes);
	hipLaunchKernelGGL(local_toggles_kernel, dim3(blocks), dim3(1024), lds_bytes, stream, T, W, window);
}

void launch_walk_toggles(const StreamTrieDev &T, const StreamWork &W, const StreamScanPlan &plan, uint64_t max_items,
                         hipStream_t stream) {
	if (max_items == 0) return;
	// rings only beside the trie image (no log block): as many waves as fit
	const uint32_t image = ((T.nslots + 15) & ~15u) + ((T.nslots * 2 + 15) & ~15u);
	const uint32_t lds_total = 160 * 1024 - 1024; // static LDS and slack
	const uint32_t waves = std::max<uint32_t>(1, std::min<uint32_t>(16, (lds_total - image) / (64 * plan.ring)));
	const uint32_t lds_bytes = image + waves * 64 * plan.ring;
	// every workgroup stages the whole trie image: no more of them than the items can keep busy
	const uint64_t lanes_per_block = 64ull * waves;
	const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(plan.blocks, (max_items + 2 * lanes_per_block - 1) / (2 * lanes_per_block)));
	auto go = [&](auto kernel) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
		hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64 * waves), lds_bytes, stream, T, W);
	};
	if (plan.ring == 64) go(&walk_toggles_kernel<64>);
	else go(&walk_toggles_kernel<128>);
}

namespace {
void histogram_slots(const StreamDev &Q, uint32_t nslots, const uint32_t *slot_id, unsigned long long *out,
                     uint32_t total_index, uint32_t *slab, uint32_t slab_blocks, hipStream_t stream, bool flip, FixDev F,
                     uint32_t next_first_batches = 0, const uint32_t *pool = nullptr) {
	constexpr uint32_t kMaxRange = 40000;
	const uint32_t ranges = (nslots + kMaxRange - 1) / kMaxRange;
	const uint32_t range = (nslots + ranges - 1) / ranges;
	const uint32_t lds = (range + 64) * 4; // + a spare word per lane
	int16_t *rows = reinterpret_cast<int16_t *>(slab); // slab_blocks rows of slab_row_stride(nslots) int16
	auto go = [&](auto kernel) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		hipLaunchKernelGGL(kernel, dim3(slab_blocks, ranges), dim3(1024), lds, stream, Q, nslots, range, rows, flip ? 1u : 0u,
		                   slot_id, out, total_index, F, pool);
	};
	if (F.blocks > slab_blocks) F.blocks = slab_blocks;
	if (ranges == 1) go(&hist_slots_kernel<true>);
	else go(&hist_slots_kernel<false>);
	hipLaunchKernelGGL(slab_reduce_map_kernel, dim3((nslots + 127) / 128), dim3(64 * kMapReduceGroups), 0, stream, rows,
	                   slab_blocks, nslots, slot_id, out, total_index, Q.cursor, next_first_batches, pool);
}
} // namespace

void launch_histogram_slots(const StreamDev &Q, uint32_t nslots, const uint32_t *slot_id, unsigned long long *out,
                            uint32_t total_index, uint32_t *slab, uint32_t slab_blocks, hipStream_t stream, bool flip) {
	histogram_slots(Q, nslots, slot_id, out, total_index, slab, slab_blocks, stream, flip, FixDev{});
}

void launch_histogram_events(const StreamDev &Q, uint32_t nslots, const uint32_t *ev, const uint32_t *pool,
                             unsigned long long *out, uint32_t total_index, uint32_t *slab, uint32_t slab_blocks, hipStream_t stream) {
	histogram_slots(Q, nslots, ev, out, total_index, slab, slab_blocks, stream, false, FixDev{}, 0, pool);
}

// A full evaluation's last two steps in one launch: the fix-up of the ranges (launch_fixup_stream) by
// the first workgroups of the histogram kernel, then the histogram and its reduction.
void launch_fixup_and_histogram_slots(const StreamTrieDev &T, const StreamWork &W, uint32_t *exit_b, uint32_t *changed,
                                      uint32_t epoch, const StreamDev &Q, uint32_t nslots, const uint32_t *slot_id,
                                      unsigned long long *out, uint32_t total_index, uint32_t *slab, uint32_t slab_blocks,
                                      hipStream_t stream, uint32_t next_first_batches) {
	FixDev F{};
	if (W.nranges >= 2) {
		F = FixDev{T, slot_id, W.prep, W.prep_total, W.range_bytes, W.nranges, W.entry, W.exit_a, exit_b, changed, epoch, out, total_index,
		           std::min<uint32_t>((W.nranges + 1023) / 1024, 64)};
	}
	histogram_slots(Q, nslots, slot_id, out, total_index, slab, slab_blocks, stream, false, F, next_first_batches);
}

void launch_fixup_stream(const StreamTrieDev &T, const uint32_t *slot_id, const StreamWork &W, uint32_t *exit_b,
                         uint32_t *changed, uint32_t epoch, unsigned long long *out, uint32_t total_index, hipStream_t stream) {
	if (W.nranges < 2) return;
	// (few workgroups: each ends with an atomic on one word, which takes about 10 ns chip-wide)
	const uint32_t blocks = std::min<uint32_t>((W.nranges + 255) / 256, 64);
	FixDev F{T, slot_id, W.prep, W.prep_total, W.range_bytes, W.nranges, W.entry, W.exit_a, exit_b, changed, epoch, out, total_index, blocks};
	hipLaunchKernelGGL(fixup_parallel_stream_kernel, dim3(blocks), dim3(256), 0, stream, F);
}

uint64_t stream_scan_capacity_batches(uint64_t prep_total, uint32_t waves_total) {
	// one entry per prepared byte at most; a flushed block holds more than kLogHigh entries, every
	// wave ends with one partial block and one partial batch
	const uint64_t blocks = prep_total / (kLogHigh + 1) + waves_total + 1;
	return (blocks + kStreamBatch - 1) / kStreamBatch + waves_total + 16;
}

} // namespace tokscore
