// kernels_generic.hip -- global-memory kernels: the generic scan (any token length, any
// vocabulary the packed trie can hold), the chunk-boundary fix-up, and the finalizer.
//
// Exact parallel greedy segmentation.  The reference's loop is serial in `pos`
// (/root/reference/src/files/SolutionFile.cpp:89-97).  Here every work chunk starts a
// chain at its own first byte (speculating that the previous chunk's last token ended
// exactly on the boundary) and counts the tokens that start inside the chunk.  The true
// chain enters chunk j at begin[j] + exit(j-1).  Two greedy chains over the same bytes
// coincide from the first position they share, and both land exactly on every document
// start (a match window is clipped there), so the repair for a wrong guess is a "dual
// walk": step whichever chain is behind, counting the true chain's tokens +1 and the
// speculative chain's tokens -1, until they meet or both have left the chunk.  If they
// never meet inside the chunk its exit overshoot changes and the next chunk's repair is
// redone in order (fixup_ordered).  All arithmetic is integer; counts are exact.
#include "kernels.h"

#include <algorithm>

namespace tokscore {

// --- generic scan: one lane per chunk -----------------------------------------------
__global__ void __launch_bounds__(256) scan_generic_kernel(TrieDev T, CorpusDev C, ChunkState S,
                                                           unsigned long long *counts) {
	for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < S.nchunks; j += gridDim.x * blockDim.x) {
		const uint64_t begin = S.begin[j];
		const uint64_t end = S.begin[j + 1];
		uint64_t pos = begin;
		uint64_t limit = next_boundary(C, pos);
		while (pos < end) {
			if (pos >= limit) limit = next_boundary(C, pos);
			uint32_t len, id;
			walk_token_global(T, C.bytes, pos, limit, len, id);
			atomicAdd(&counts[id], 1ull);
			pos += len;
		}
		S.entry[j] = 0;
		S.exit_a[j] = (uint32_t)(pos - end);
	}
}

// Dual walk inside chunk [begin, end): chain A (true, counted +1) starts at pa, chain B
// (currently counted, now -1) at pb.  Returns true when they met (exit unchanged).
__device__ bool dual_walk(const TrieDev &T, const CorpusDev &C, uint64_t end, uint64_t pa, uint64_t pb,
                          unsigned long long *counts, uint64_t *pa_out) {
	// no document start lies strictly between the chunk start and either entry, so one
	// clip limit serves both chains until they meet (they meet at the latest there)
	uint64_t limit = next_boundary(C, pa < pb ? pa : pb);
	while (pa != pb && (pa < end || pb < end)) {
		uint32_t len, id;
		if (pa < pb) {
			if (pa >= limit) limit = next_boundary(C, pa);
			walk_token_global(T, C.bytes, pa, limit, len, id);
			atomicAdd(&counts[id], 1ull);
			pa += len;
		} else {
			if (pb >= limit) limit = next_boundary(C, pb);
			walk_token_global(T, C.bytes, pb, limit, len, id);
			atomicAdd(&counts[id], ~0ull); // -1 mod 2^64
			pb += len;
		}
	}
	*pa_out = pa;
	return pa == pb;
}

// --- fix-up, parallel round: every chunk repairs against its predecessor's scan exit --
__global__ void __launch_bounds__(256) fixup_parallel_kernel(TrieDev T, CorpusDev C, ChunkState S,
                                                             unsigned long long *counts) {
	for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < S.nchunks; j += gridDim.x * blockDim.x) {
		uint32_t ex = S.exit_a[j];
		if (j > 0) {
			const uint32_t want = S.exit_a[j - 1];
			const uint32_t have = S.entry[j];
			if (want != have) {
				const uint64_t begin = S.begin[j];
				const uint64_t end = S.begin[j + 1];
				uint64_t pa;
				const bool met = dual_walk(T, C, end, begin + want, begin + have, counts, &pa);
				if (!met) ex = (uint32_t)(pa - end);
				S.entry[j] = want;
			}
		}
		S.exit_b[j] = ex;
	}
}

// --- fix-up, ordered round: one workgroup sweeps the chunks in order and redoes the
// repair wherever the predecessor's exit is not the entry the chunk was repaired for
// (only after a dual walk that never met, e.g. long periodic runs).  Launch with ONE
// block; every thread reaches every barrier.
__global__ void __launch_bounds__(1024) fixup_ordered_kernel(TrieDev T, CorpusDev C, ChunkState S,
                                                             unsigned long long *counts) {
	__shared__ uint32_t s_first;
	for (uint32_t base = 1; base < S.nchunks; base += blockDim.x) {
		for (;;) {
			const uint32_t j = base + threadIdx.x;
			const bool need = j < S.nchunks && S.exit_b[j - 1] != S.entry[j];
			if (threadIdx.x == 0) s_first = 0xFFFFFFFFu;
			__syncthreads();
			if (need) atomicMin(&s_first, j);
			__syncthreads();
			const uint32_t first = s_first;
			if (first == 0xFFFFFFFFu) break; // uniform: s_first is read after the barrier
			if (j == first) {
				const uint32_t want = S.exit_b[j - 1];
				const uint32_t have = S.entry[j];
				const uint64_t begin = S.begin[j];
				const uint64_t end = S.begin[j + 1];
				uint64_t pa;
				const bool met = dual_walk(T, C, end, begin + want, begin + have, counts, &pa);
				if (!met) S.exit_b[j] = (uint32_t)(pa - end);
				S.entry[j] = want;
				__threadfence();
			}
			__syncthreads();
		}
		__syncthreads();
	}
}

// --- chunk boundaries at guaranteed token starts ----------------------------------------
// A position p inside a document is a start of a token of EVERY greedy chain, whatever the
// chain's own start and whichever tokens are enabled, when no trie path that begins before p
// reaches past p: the token that covers byte p-1 cannot extend over p, so it ends there.
// (Trie paths, not ids: a superset of every enabled vocabulary, so toggles keep it valid.)
// One wave per interior boundary moves it forward to the first such position within `window`
// bytes (or to the next document start); a chunk that begins there needs no entry repair.
// reach = max over starts i < p of i + depth(i), depth(i) = length of the longest trie path
// matching the text at i (clipped at the document end, SolutionFile.cpp:90).
__global__ void __launch_bounds__(256) snap_chunks_kernel(TrieDev T, CorpusDev C, const uint64_t *begin_in,
                                                          uint64_t *begin_out, uint32_t *synced, uint32_t nchunks,
                                                          uint32_t max_len, uint32_t window) {
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
	for (uint32_t j = 1 + wave; j < nchunks; j += nwaves) {
		const uint64_t b = begin_in[j];
		const uint64_t dn = next_boundary(C, b); // first document start after b (corpus end if none)
		uint64_t lo = 0, hi = C.D + 1;           // ds = last document start at or before b
		while (lo < hi) {
			const uint64_t mid = (lo + hi) >> 1;
			if (C.off[mid] <= b) lo = mid + 1;
			else hi = mid;
		}
		const uint64_t ds = C.off[lo - 1];
		uint64_t p = b;
		bool ok = ds == b;
		if (!ok) {
			const uint64_t back = (b - ds) < (uint64_t)(max_len - 1) ? (b - ds) : (uint64_t)(max_len - 1);
			const uint64_t last = (dn < b + window) ? dn : b + window - 1; // last candidate position
			uint64_t carry = 0;
			for (uint64_t base = b - back; base <= last; base += 64) {
				const uint64_t i = base + lane;
				uint64_t r = 0;
				if (i < dn) {
					uint32_t nb = T.root_base, d = 0;
					while (nb != 0 && i + d < dn) {
						const uint32_t c = lower8(C.bytes[i + d]);
						const uint2 e = T.slots[nb + c];
						if ((e.x & kSlotKeyMask) != (c | kSlotValid)) break;
						d++;
						nb = e.x >> kSlotBaseShift;
					}
					r = i + d;
				}
				uint64_t incl = r;
				for (uint32_t o = 1; o < 64; o <<= 1) {
					const uint64_t t = __shfl_up(incl, o, 64);
					if (lane >= o && t > incl) incl = t;
				}
				uint64_t excl = __shfl_up(incl, 1, 64);
				if (lane == 0) excl = 0;
				if (carry > excl) excl = carry;
				const bool cand = i >= b && i <= last && excl <= i;
				const unsigned long long m = __ballot(cand);
				if (m) {
					p = base + (uint64_t)__builtin_ctzll(m);
					ok = true;
					break;
				}
				const uint64_t top = __shfl(incl, 63, 64);
				if (top > carry) carry = top;
			}
		}
		if (lane == 0) {
			begin_out[j] = ok ? p : b;
			synced[j] = ok ? 1u : 0u;
		}
	}
}

// --- incremental re-score: which chunks can a toggled token change? -------------------
// Chunk boundaries are token starts of every chain under every subset of enabled tokens
// (snap_chunks_kernel), so a toggle changes the segmentation of exactly the chunks that hold an
// occurrence of the token's string (lower-cased text against the token's bytes as they are,
// SolutionFile.cpp:87-88).  One thread per 16 corpus bytes; the first byte filters, the rest is
// compared from L1/L2.  A superset is fine (occurrences across document starts are not
// excluded): re-scoring an unchanged chunk changes nothing.
__global__ void __launch_bounds__(256) find_token_chunks_kernel(CorpusDev C, const uint64_t *begin, uint32_t nchunks,
                                                                uint64_t chunk_bytes, const uint8_t *tok, uint32_t len,
                                                                uint32_t *sel) { // sel: one bit per chunk
	const uint64_t nvec = (C.total + 15) / 16;
	const uint32_t t0 = tok[0];
	for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t a = v * 16; // the corpus buffer is padded past total: a whole vector is readable
		const uint4 x = *reinterpret_cast<const uint4 *>(C.bytes + a);
		const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
		for (int q = 0; q < 4; q++) {
			// bytes of this dword equal to the token's first byte, four at a time: lower-case
			// (A-Z only), xor with the byte broadcast, exact zero-byte test
			const uint32_t low7 = w[q] & 0x7F7F7F7Fu;
			const uint32_t up = (low7 + 0x3F3F3F3Fu) & ~(low7 + 0x25252525u) & ~w[q] & 0x80808080u;
			const uint32_t y = (w[q] | (up >> 2)) ^ (t0 * 0x01010101u);
			uint32_t hit = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu); // 0x80 where a byte of y is zero
			while (hit) {
				const uint32_t k = (uint32_t)q * 4 + ((uint32_t)__builtin_ctz(hit) >> 3);
				hit &= hit - 1;
				const uint64_t p = a + k;
				if (p + len > C.total) continue;
				uint32_t m = 1;
				while (m < len && lower8(C.bytes[p + m]) == tok[m]) m++;
				if (m < len) continue;
				uint64_t j = p / chunk_bytes;
				if (j >= nchunks) j = nchunks - 1;
				while (p < begin[j]) j--; // the boundary was moved forward past p (chunks without a cut of their own are empty)
				const uint32_t bit = 1u << (j & 31);
				if (!(sel[j >> 5] & bit)) atomicOr(&sel[j >> 5], bit);
			}
		}
	}
}

// chunk bitmap -> list of the selected chunk indices (any order) + their number in list[nchunks]:
// the partial scan's work queue then runs over the list only instead of skipping through every
// chunk of the corpus (one queue pull per chunk costs a wave about a microsecond)
__global__ void __launch_bounds__(1024) compact_sel_kernel(const uint32_t *sel, uint32_t words, uint32_t nchunks,
                                                           uint32_t *list) {
	__shared__ uint32_t s_n;
	if (threadIdx.x == 0) s_n = 0;
	__syncthreads();
	for (uint32_t j = threadIdx.x; j < words; j += blockDim.x) {
		uint32_t w = sel[j];
		if (!w) continue;
		uint32_t at = atomicAdd(&s_n, (uint32_t)__builtin_popcount(w));
		while (w) {
			const uint32_t chunk = j * 32 + (uint32_t)__builtin_ctz(w);
			w &= w - 1;
			if (chunk < nchunks && at < nchunks) list[at] = chunk;
			at++;
		}
	}
	__syncthreads();
	if (threadIdx.x == 0) list[nchunks] = s_n < nchunks ? s_n : nchunks;
}

// --- occurrence index: which chunks does each vocabulary string occur in? ---------------
// One bit per (string, chunk), built once per plan so that an incremental re-score needs no
// search: from every byte position of the lower-cased corpus walk the trie and mark every
// string passed (whatever is enabled: slot_row maps a slot to its string's row).  Document
// starts are not honoured -- a superset of the chunks a toggle can change is all that is needed.
__global__ void __launch_bounds__(256) index_all_kernel(TrieDev T, CorpusDev C, const int32_t *slot_row,
                                                        const uint64_t *begin, uint32_t nchunks, uint64_t chunk_bytes,
                                                        uint32_t words_per_row, uint32_t *index) {
	for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < C.total; p += (uint64_t)gridDim.x * blockDim.x) {
		uint64_t j = p / chunk_bytes;
		if (j >= nchunks) j = nchunks - 1;
		while (p < begin[j]) j--; // (begin[0] = 0; a chunk without a cut of its own is empty: begin[j] = begin[j + 1])
		const uint32_t word = (uint32_t)(j >> 5), bit = 1u << (j & 31);
		uint32_t nb = T.root_base;
		for (uint64_t q = p; nb != 0 && q < C.total; q++) {
			const uint32_t c = lower8(C.bytes[q]);
			const uint32_t t = nb + c;
			const uint2 e = T.slots[t];
			if ((e.x & kSlotKeyMask) != (c | kSlotValid)) break;
			nb = e.x >> kSlotBaseShift;
			const int32_t row = slot_row[t];
			if (row >= 0) {
				uint32_t *w = index + (size_t)row * words_per_row + word;
				if (!(*w & bit)) atomicOr(w, bit);
			}
		}
	}
}

// row_chunks[r] = number of chunks string r occurs in
__global__ void __launch_bounds__(256) index_row_counts_kernel(const uint32_t *index, uint32_t rows, uint32_t words_per_row,
                                                               uint32_t *row_chunks) {
	const uint32_t r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; // one wave per row
	const uint32_t lane = threadIdx.x & 63;
	if (r >= rows) return;
	uint32_t n = 0;
	for (uint32_t w = lane; w < words_per_row; w += 64) n += (uint32_t)__builtin_popcount(index[(size_t)r * words_per_row + w]);
	for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o, 64);
	if (lane == 0) row_chunks[r] = n;
}

// sel = OR of the listed index rows
struct RowList {
	uint32_t n;
	uint32_t row[8];
};
__global__ void __launch_bounds__(256) or_rows_kernel(const uint32_t *index, RowList rows, uint32_t words_per_row,
                                                      uint32_t *sel) {
	const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= words_per_row) return;
	uint32_t v = 0;
	for (uint32_t k = 0; k < rows.n; k++) v |= index[(size_t)rows.row[k] * words_per_row + w];
	sel[w] = v;
}

// --- finalize: START/END once per document, totals (main.cpp:40-41) -------------------
// out = counts[V+3] | total_tokens | total_bytes.  One block.
__global__ void __launch_bounds__(1024) finalize_kernel(unsigned long long *out, uint32_t V, uint64_t D,
                                                        uint64_t total_bytes) {
	__shared__ unsigned long long s_part[16];
	unsigned long long sum = 0;
	for (uint32_t i = 2 + threadIdx.x; i < V + 3; i += blockDim.x) sum += out[i];
	for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
	if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sum;
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned long long t = 0;
		for (uint32_t w = 0; w < (blockDim.x + 63) / 64; w++) t += s_part[w];
		out[0] = D; // <START> (SolutionFile.cpp:85)
		out[1] = D; // <END>   (SolutionFile.cpp:98)
		out[V + 3] = t;
		out[V + 4] = total_bytes;
	}
}

// --- per-document id sequence (the return value of Tokenize) --------------------------
// One wave; lane 0 walks the chain.  Debug / REPL sized inputs only (main.cpp:48-55).
__global__ void tokenize_one_kernel(TrieDev T, const uint8_t *text, uint64_t n, unsigned long long *ids,
                                    uint64_t cap, unsigned long long *n_ids) {
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	uint64_t k = 0;
	if (k < cap) ids[k] = 0; // <START>
	k++;
	uint64_t pos = 0;
	while (pos < n) {
		uint32_t len, id;
		walk_token_global(T, text, pos, n, len, id);
		if (k < cap) ids[k] = (id == T.unknown_id) ? ~0ull : (unsigned long long)id;
		k++;
		pos += len;
	}
	if (k < cap) ids[k] = 1; // <END>
	k++;
	*n_ids = k;
}

// --- host launchers ----------------------------------------------------------------------
static inline uint32_t grid_for(uint32_t n, uint32_t block, uint32_t cap) {
	uint32_t g = (n + block - 1) / block;
	if (g < 1) g = 1;
	return g > cap ? cap : g;
}

void launch_scan_generic(const TrieDev &T, const CorpusDev &C, const ChunkState &S, unsigned long long *counts,
                         hipStream_t stream) {
	if (S.nchunks == 0) return;
	hipLaunchKernelGGL(scan_generic_kernel, dim3(grid_for(S.nchunks, 256, 8192)), dim3(256), 0, stream, T, C, S, counts);
}

void launch_fixup(const TrieDev &T, const CorpusDev &C, const ChunkState &S, unsigned long long *counts,
                  hipStream_t stream) {
	if (S.nchunks == 0) return;
	hipLaunchKernelGGL(fixup_parallel_kernel, dim3(grid_for(S.nchunks, 256, 8192)), dim3(256), 0, stream, T, C, S, counts);
	if (S.nchunks > 1)
		hipLaunchKernelGGL(fixup_ordered_kernel, dim3(1), dim3(1024), 0, stream, T, C, S, counts);
}

void launch_snap_chunks(const TrieDev &T, const CorpusDev &C, const uint64_t *begin_in, uint64_t *begin_out,
                        uint32_t *synced, uint32_t nchunks, uint32_t max_len, uint32_t window, hipStream_t stream) {
	if (nchunks < 2) return;
	const uint32_t waves = nchunks - 1;
	const uint32_t blocks = std::min<uint32_t>((waves + 3) / 4, 4096);
	hipLaunchKernelGGL(snap_chunks_kernel, dim3(blocks), dim3(256), 0, stream, T, C, begin_in, begin_out, synced, nchunks,
	                   max_len, window);
}

void launch_find_token_chunks(const CorpusDev &C, const uint64_t *begin, uint32_t nchunks, uint64_t chunk_bytes,
                              const uint8_t *tok, uint32_t len, uint32_t *sel, hipStream_t stream) {
	if (len == 0 || nchunks == 0 || C.total == 0) return;
	const uint64_t nvec = (C.total + 15) / 16;
	const uint32_t blocks = (uint32_t)std::min<uint64_t>((nvec + 255) / 256, 16384);
	hipLaunchKernelGGL(find_token_chunks_kernel, dim3(blocks), dim3(256), 0, stream, C, begin, nchunks, chunk_bytes, tok,
	                   len, sel);
}

void launch_compact_sel(const uint32_t *sel, uint32_t nchunks, uint32_t *list, hipStream_t stream) {
	hipLaunchKernelGGL(compact_sel_kernel, dim3(1), dim3(1024), 0, stream, sel, (nchunks + 31) / 32, nchunks, list);
}

void launch_index_all(const TrieDev &T, const CorpusDev &C, const int32_t *slot_row, const uint64_t *begin,
                      uint32_t nchunks, uint64_t chunk_bytes, uint32_t rows, uint32_t *index, uint32_t *row_chunks,
                      hipStream_t stream) {
	if (nchunks == 0 || rows == 0) return;
	const uint32_t wpr = (nchunks + 31) / 32;
	if (C.total) {
		const uint32_t blocks = (uint32_t)std::min<uint64_t>((C.total + 255) / 256, 1u << 20);
		hipLaunchKernelGGL(index_all_kernel, dim3(blocks), dim3(256), 0, stream, T, C, slot_row, begin, nchunks, chunk_bytes,
		                   wpr, index);
	}
	hipLaunchKernelGGL(index_row_counts_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, index, rows, wpr, row_chunks);
}

void launch_or_rows(const uint32_t *index, const uint32_t *rows, uint32_t n, uint32_t nchunks, uint32_t *sel,
                    hipStream_t stream) {
	RowList L;
	L.n = n < 8 ? n : 8;
	for (uint32_t k = 0; k < L.n; k++) L.row[k] = rows[k];
	const uint32_t wpr = (nchunks + 31) / 32;
	hipLaunchKernelGGL(or_rows_kernel, dim3((wpr + 255) / 256), dim3(256), 0, stream, index, L, wpr, sel);
}

void launch_finalize(unsigned long long *out, uint32_t V, uint64_t D, uint64_t total_bytes, hipStream_t stream) {
	hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(1024), 0, stream, out, V, D, total_bytes);
}

void launch_tokenize_one(const TrieDev &T, const uint8_t *text, uint64_t n, unsigned long long *ids, uint64_t cap,
                         unsigned long long *n_ids, hipStream_t stream) {
	hipLaunchKernelGGL(tokenize_one_kernel, dim3(1), dim3(64), 0, stream, T, text, n, ids, cap, n_ids);
}

// ---------------------------------------------------------------------------------------------
// Read-only streaming probe ("report against a measured read-only streaming
// kernel on the same box"): what this GPU's memory system delivers to a kernel that does nothing
// but read.  16 bytes per lane and load, four loads in flight per lane, grid-stride.
typedef uint32_t probe_v4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void read_probe_kernel(const probe_v4 *__restrict__ p, uint64_t n16, uint32_t *sink) {
	const uint64_t stride = (uint64_t)gridDim.x * 256;
	uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	probe_v4 a = 0, b = 0, c = 0, d = 0;
	for (; i + 3 * stride < n16; i += 4 * stride) {
		const probe_v4 v0 = __builtin_nontemporal_load(p + i);
		const probe_v4 v1 = __builtin_nontemporal_load(p + i + stride);
		const probe_v4 v2 = __builtin_nontemporal_load(p + i + 2 * stride);
		const probe_v4 v3 = __builtin_nontemporal_load(p + i + 3 * stride);
		a ^= v0, b ^= v1, c ^= v2, d ^= v3;
	}
	for (; i < n16; i += stride) a ^= __builtin_nontemporal_load(p + i);
	a ^= b ^ c ^ d;
	const uint32_t x = a.x + a.y + a.z + a.w;
	if (x == 0x9E3779B9u) sink[0] = x; // keeps the loads alive; the buffer is all zero
}

void launch_read_probe(const void *buf, uint64_t bytes, uint32_t blocks, uint32_t *sink, hipStream_t stream) {
	hipLaunchKernelGGL(read_probe_kernel, dim3(blocks), dim3(256), 0, stream, (const probe_v4 *)buf, bytes / 16, sink);
}

} // namespace tokscore
