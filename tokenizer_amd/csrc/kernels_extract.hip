// kernels_extract.hip -- row f3: open-vocabulary candidate extraction on the
// GPU.  What annealing::GetTokens builds with a pointer trie on the host
// (src/tokenizer/GetTokens.cpp:21-90, src/tokenizer/Trie.cpp:81-152) -- every distinct window of
// 1 .. max_cp code points, in trie pre-order with children sorted by code point, with its
// occurrence count and the indices of the string minus its first / last code point -- from one
// sort of all windows:
//
//   1. starts     S[r] = byte position of code point r (lead bytes, utf8/core.h:134-147; valid
//                 UTF-8, as for the counting kernel), len[r] = code points of its window (max_cp,
//                 clipped at the end of its document, GetTokens.cpp:25)
//   2. sort       the windows as byte strings, shorter first on a tie (zero padded 8-byte words,
//                 LSD radix sort by word); for valid UTF-8 byte order is code point order, so
//                 this is the trie's pre-order with the duplicates still in
//   3. lcp        common code points of each window with its predecessor; window i adds the
//                 nodes of depth lcp[i]+1 .. len[i] to the trie, in that order: a prefix sum of
//                 len - lcp numbers the tokens as BuildToken does (Trie.cpp:81-91)
//   4. runs       per depth d, run_d[j] = the window that created the depth-d node on window j's
//                 path (a max-scan over "lcp < d"); a node's count is the length of its run
//                 (Trie.cpp:114-128: one increment per window through it)
//   5. parents    r parent = the depth d-1 node of the same path; l parent = the depth d-1
//                 node on the path of the window one code point further on (Trie.cpp:92-103)
//
// No trie, no hash table, no atomics; memory is sized for 288 GB (the run table is
// 4 * max_cp bytes per code point of corpus).
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <utility>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "device_common.h"
#include "kernels.h"

namespace tokscore {

namespace {

constexpr uint32_t kNoParent = 0xFFFFFFFFu;
constexpr int kThreads = 256;

__device__ __forceinline__ bool is_start(uint8_t b) { return (b & 0xC0) != 0x80; }

// ---- 1. starts ------------------------------------------------------------------------------

// one flag per byte is not materialised: a block counts the starts of its 16 KiB tile, the tile
// bases are scanned, and the same tiles are walked again to write S
constexpr uint32_t kTile = 16384;

__global__ __launch_bounds__(kThreads) void count_starts_kernel(const uint8_t *bytes, uint64_t total, uint32_t *tile_count,
                                                                 uint32_t *has_nul) {
	const uint64_t lo = (uint64_t)blockIdx.x * kTile;
	const uint64_t hi = lo + kTile < total ? lo + kTile : total;
	uint32_t n = 0, nul = 0;
	for (uint64_t p = lo + threadIdx.x; p < hi; p += kThreads) {
		const uint8_t b = bytes[p];
		n += is_start(b);
		nul |= b == 0;
	}
	__shared__ uint32_t red[kThreads];
	red[threadIdx.x] = n;
	__syncthreads();
	for (int s = kThreads / 2; s > 0; s >>= 1) {
		if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
		__syncthreads();
	}
	if (threadIdx.x == 0) tile_count[blockIdx.x] = red[0];
	if (nul) *has_nul = 1;
}

__global__ __launch_bounds__(kThreads) void fill_starts_kernel(const uint8_t *bytes, uint64_t total, const uint32_t *tile_base,
                                                                uint32_t *S) {
	// thread t owns the contiguous slice [lo + t*64, +64) of the tile, so ranks stay in position order
	constexpr uint32_t kSlice = kTile / kThreads;
	const uint64_t lo = (uint64_t)blockIdx.x * kTile + (uint64_t)threadIdx.x * kSlice;
	const uint64_t hi = lo + kSlice < total ? lo + kSlice : total;
	uint32_t n = 0;
	for (uint64_t p = lo; p < hi; p++) n += is_start(bytes[p]);
	__shared__ uint32_t pre[kThreads];
	pre[threadIdx.x] = n;
	__syncthreads();
	for (int s = 1; s < kThreads; s <<= 1) { // Hillis-Steele inclusive scan
		const uint32_t v = (int)threadIdx.x >= s ? pre[threadIdx.x - s] : 0;
		__syncthreads();
		pre[threadIdx.x] += v;
		__syncthreads();
	}
	uint32_t r = tile_base[blockIdx.x] + pre[threadIdx.x] - n;
	for (uint64_t p = lo; p < hi; p++)
		if (is_start(bytes[p])) S[r++] = (uint32_t)p;
}

// rank of the first code point at or after each document offset (lower bound in S)
__global__ void doc_rank_kernel(const uint32_t *S, uint32_t N, const uint64_t *off, uint64_t D, uint32_t *doc_rank) {
	const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (d > D) return;
	const uint64_t want = off[d];
	uint32_t lo = 0, hi = N;
	while (lo < hi) {
		const uint32_t mid = lo + (hi - lo) / 2;
		if (S[mid] < want) lo = mid + 1;
		else hi = mid;
	}
	doc_rank[d] = lo;
}

// win[r] = position:32 | window bytes:16 | window code points:8 -- everything the later stages need
// to know about window r in one 8-byte read (they visit the windows in sorted, i.e. random, order).
// Code points: from r to the end of r's document, capped at max_cp (GetTokens.cpp:25).
// *max_bytes: the longest window in bytes (how many key words the sort needs).
__device__ __forceinline__ uint32_t win_pos(uint64_t rec) { return (uint32_t)rec; }
__device__ __forceinline__ uint32_t win_bytes(uint64_t rec) { return (uint32_t)(rec >> 32) & 0xFFFFu; }
__device__ __forceinline__ uint32_t win_cps(uint64_t rec) { return (uint32_t)(rec >> 48) & 0xFFu; }

__global__ __launch_bounds__(kThreads) void window_len_kernel(const uint32_t *S, uint32_t N, const uint64_t *off, uint64_t D,
                                                               const uint32_t *doc_rank, uint32_t max_cp, uint64_t *win,
                                                               uint32_t *max_bytes) {
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	// the document of the block's first code point by bisection (last d with off[d] <= p); the
	// other lanes walk on from there (a block of 256 code points rarely spans more than two)
	__shared__ uint64_t first_doc;
	if (threadIdx.x == 0) {
		const uint64_t p = S[(uint64_t)blockIdx.x * blockDim.x];
		uint64_t lo = 0, hi = D;
		while (lo + 1 < hi) {
			const uint64_t mid = lo + (hi - lo) / 2;
			if (off[mid] <= p) lo = mid;
			else hi = mid;
		}
		first_doc = lo;
	}
	__syncthreads();
	uint32_t nbytes = 0;
	if (r < N) {
		const uint64_t p = S[r];
		uint64_t d = first_doc;
		while (d + 1 < D && off[d + 1] <= p) d++;
		const uint32_t left = doc_rank[d + 1] - (uint32_t)r;
		const uint32_t n = left < max_cp ? left : max_cp;
		nbytes = S[r + n] - S[r];
		win[r] = (uint64_t)S[r] | (uint64_t)nbytes << 32 | (uint64_t)n << 48;
	}
	__shared__ uint32_t red[kThreads];
	red[threadIdx.x] = nbytes;
	__syncthreads();
	for (int s = kThreads / 2; s > 0; s >>= 1) {
		if ((int)threadIdx.x < s && red[threadIdx.x + s] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + s];
		__syncthreads();
	}
	// (one word cannot take an atomic per block of a large grid: most blocks see they have nothing to add)
	if (threadIdx.x == 0 && red[0] > *(volatile uint32_t *)max_bytes) atomicMax(max_bytes, red[0]);
}

// ---- 2. sort --------------------------------------------------------------------------------

// word w of a window: its bytes [8w, 8w + 8) clipped at the window end, big endian, zero padded.
// Two aligned 8-byte loads and a funnel shift (the corpus buffer is 256-byte aligned and readable
// past its end).
__device__ __forceinline__ uint64_t window_word(const uint8_t *bytes, uint64_t rec, uint32_t w) {
	const uint32_t n = win_bytes(rec);
	if (8 * w >= n) return 0;
	const uint64_t begin = (uint64_t)win_pos(rec) + 8ull * w;
	const uint64_t *q = (const uint64_t *)(bytes + (begin & ~7ull));
	const uint32_t sh = (uint32_t)(begin & 7) * 8;
	const uint64_t lo = q[0];
	uint64_t v = lo;
	if (sh) v = (lo >> sh) | (q[1] << (64 - sh));
	uint64_t key = __builtin_bswap64(v); // byte k of the word at bits 63-8k
	const uint32_t valid = n - 8 * w;
	if (valid < 8) key &= ~0ull << (8 * (8 - valid));
	return key;
}

__global__ void gather_keys_kernel(const uint8_t *bytes, const uint64_t *win, const uint32_t *perm, uint32_t N, uint32_t w,
                                   uint64_t *keys) {
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < N) keys[i] = window_word(bytes, win[perm[i]], w);
}

__global__ void iota_kernel(uint32_t *v, uint32_t N) {
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < N) v[i] = (uint32_t)i;
}

// ---- 3. lcp ---------------------------------------------------------------------------------

// common code points of sorted windows i-1 and i; slen[i] = code points of sorted window i;
// inv = perm^-1; srec[i] = the window record in sorted order.  Every lane fetches its own window
// once (record + key words) and leaves it in LDS for the lane after it; only the first lane of a
// block fetches its predecessor itself.  Dynamic LDS: (words + 1) * blockDim.x * 8 bytes.
__global__ __launch_bounds__(kThreads) void lcp_kernel(const uint8_t *bytes, const uint64_t *win, const uint32_t *perm, uint32_t N,
                                                        uint32_t words, uint8_t *lcp, uint8_t *slen, uint32_t *inv, uint32_t *fresh,
                                                        uint64_t *srec) {
	extern __shared__ uint64_t extract_rows[]; // [0 .. words) rows of key words, row `words` = records
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t t = threadIdx.x;
	uint64_t rb = 0;
	if (i < N) {
		const uint32_t b = perm[i];
		rb = win[b];
		inv[b] = (uint32_t)i;
		srec[i] = rb;
		for (uint32_t w = 0; w < words; w++) extract_rows[w * kThreads + t] = window_word(bytes, rb, w);
	}
	extract_rows[words * kThreads + t] = rb;
	__syncthreads();
	if (i >= N) return;
	const uint32_t lb = win_cps(rb), nb = win_bytes(rb);
	slen[i] = (uint8_t)lb;
	uint32_t common = 0;
	if (i > 0) {
		const uint64_t ra = t ? extract_rows[words * kThreads + t - 1] : win[perm[i - 1]];
		const uint32_t na = win_bytes(ra);
		const uint32_t n = na < nb ? na : nb;
		// common bytes, a word at a time (a window that ended reads as zeros, a document holds no zero byte)
		uint32_t same = n;
		for (uint32_t w = 0; 8 * w < n; w++) {
			const uint64_t A = t ? extract_rows[w * kThreads + t - 1] : window_word(bytes, ra, w);
			const uint64_t x = A ^ extract_rows[w * kThreads + t];
			if (x) {
				const uint32_t at = 8 * w + (uint32_t)(__builtin_clzll(x) >> 3);
				same = at < n ? at : n;
				break;
			}
		}
		// whole code points of window b inside them: offsets 1 .. same where a code point starts
		// (the zero byte after the window's end counts as one: the last code point ended there)
		for (uint32_t w = 0; 8 * w <= same && same > 0; w++) {
			const uint32_t lo = w == 0 ? 1 : 8 * w, hi = same < 8 * w + 7 ? same : 8 * w + 7;
			if (lo > hi) continue;
			const uint64_t B = w < words ? extract_rows[w * kThreads + t] : 0;
			const uint64_t y = (B ^ 0x8080808080808080ull) & 0xC0C0C0C0C0C0C0C0ull; // zero byte = continuation byte
			const uint64_t starts = (y | (y >> 1)) & 0x4040404040404040ull;
			const uint32_t qlo = lo - 8 * w, qhi = hi - 8 * w; // byte 0 is the most significant
			const uint64_t sel = (~0ull >> (8 * qlo)) & (~0ull << (8 * (7 - qhi)));
			common += (uint32_t)__builtin_popcountll(starts & sel);
		}
		if (common > lb) common = lb;
	}
	lcp[i] = (uint8_t)common;
	fresh[i] = lb - common; // trie nodes this window adds
}

// ---- 4. runs --------------------------------------------------------------------------------

struct RunStart { // input of the per-depth max-scan: i where a depth-d node starts, else 0
	const uint8_t *lcp;
	uint32_t d;
	__device__ uint32_t operator()(uint32_t i) const { return lcp[i] < d ? i : 0u; }
};

struct MaxOp {
	__device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};

__device__ __forceinline__ uint32_t token_of(const uint64_t *base, const uint8_t *lcp, uint32_t s, uint32_t d) {
	return (uint32_t)(base[s] + d - lcp[s] - 1);
}

// a run at depth d that ends before window j (j == N: the end of everything) gives its node's count
__global__ void run_uses_kernel(const uint32_t *run, const uint8_t *lcp, const uint8_t *slen, const uint64_t *base, uint32_t N,
                                uint32_t max_cp, uint64_t *uses) {
	const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j == 0 || j > N) return;
	const uint32_t l = j < N ? lcp[j] : 0;
	for (uint32_t d = l + 1; d <= max_cp; d++) { // every depth above lcp[j] starts a new run at j
		const uint32_t s = run[(uint64_t)(d - 1) * N + (j - 1)];
		if (slen[s] >= d) uses[token_of(base, lcp, s, d)] = j - s;
	}
}

// ---- 5. tokens ------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t first_cp(const uint8_t *p) { // utf8::unchecked::next (utf8/unchecked.h:104-131)
	uint32_t cp = p[0];
	if (cp < 0x80) return cp;
	if ((cp >> 5) == 0x6) return ((cp << 6) & 0x7ff) + (p[1] & 0x3f);
	if ((cp >> 4) == 0xe) return ((cp << 12) & 0xffff) + (((uint32_t)p[1] << 6) & 0xfff) + (p[2] & 0x3f);
	if ((cp >> 3) == 0x1e)
		return ((cp << 18) & 0x1fffff) + (((uint32_t)p[1] << 12) & 0x3ffff) + (((uint32_t)p[2] << 6) & 0xfff) + (p[3] & 0x3f);
	return cp;
}

__global__ void fill_tokens_kernel(const uint8_t *bytes, const uint64_t *srec, const uint32_t *perm, const uint32_t *inv,
                                   const uint32_t *run, const uint8_t *lcp, const uint8_t *slen, const uint64_t *base, uint32_t N,
                                   uint32_t *chr, uint32_t *lpar, uint32_t *rpar) {
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= N) return;
	const uint32_t l = lcp[i], n = slen[i];
	if (l >= n) return;
	const uint32_t r = perm[i];
	const uint32_t cp = first_cp(bytes + win_pos(srec[i])); // BuildToken hands the FIRST code point down (Trie.cpp:81-89)
	// the window one code point on (same document whenever this one has two code points or more)
	const uint32_t j = n >= 2 ? inv[r + 1] : 0;
	for (uint32_t d = l + 1; d <= n; d++) {
		const uint32_t t = token_of(base, lcp, (uint32_t)i, d);
		chr[t] = cp;
		if (d == 1) {
			lpar[t] = rpar[t] = kNoParent;
		} else {
			rpar[t] = token_of(base, lcp, run[(uint64_t)(d - 2) * N + i], d - 1); // Trie.cpp:98
			lpar[t] = token_of(base, lcp, run[(uint64_t)(d - 2) * N + j], d - 1); // Trie.cpp:99
		}
	}
}

struct DeviceBuf {
	void *p = nullptr;
	~DeviceBuf() {
		if (p) (void)hipFree(p);
	}
	hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
	template <typename T> T *as() const { return (T *)p; }
};

inline uint32_t blocks_for(uint64_t n) { return (uint32_t)((n + kThreads - 1) / kThreads); }

} // namespace

#define EX_TRY(expr)                                                                      \
	do {                                                                                  \
		hipError_t e_ = (expr);                                                           \
		if (e_ != hipSuccess) {                                                           \
			if (err) *err = std::string("extract_candidates: ") + hipGetErrorString(e_); \
			return 1;                                                                     \
		}                                                                                 \
	} while (0)

int extract_candidates(const CorpusDev &C, uint32_t max_cp, CandidateArraysDev *out, std::string *err, hipStream_t stream) {
	*out = CandidateArraysDev{};
	if (max_cp == 0 || max_cp > kExtractMaxCp) {
		if (err) *err = "extract_candidates: max_cp must be 1 .. " + std::to_string(kExtractMaxCp);
		return 2;
	}
	if (C.total >= 0xFFFFFFF0ull) {
		if (err) *err = "extract_candidates: corpus of 4 GiB or more (32-bit positions)";
		return 2;
	}
	if (C.total == 0) return 0;
	const uint32_t tiles = (uint32_t)((C.total + kTile - 1) / kTile);

	// 1. starts
	DeviceBuf tile_count, flags;
	EX_TRY(tile_count.alloc(((size_t)tiles + 1) * sizeof(uint32_t)));
	EX_TRY(flags.alloc(sizeof(uint32_t)));
	EX_TRY(hipMemsetAsync(flags.p, 0, sizeof(uint32_t), stream));
	EX_TRY(hipMemsetAsync(tile_count.as<uint32_t>() + tiles, 0, sizeof(uint32_t), stream));
	hipLaunchKernelGGL(count_starts_kernel, dim3(tiles), dim3(kThreads), 0, stream, C.bytes, C.total, tile_count.as<uint32_t>(),
	                   flags.as<uint32_t>());
	DeviceBuf temp;
	size_t temp_bytes = 0, need = 0;
	auto ensure_temp = [&](size_t n) -> hipError_t {
		if (n <= temp_bytes) return hipSuccess;
		if (temp.p) (void)hipFree(temp.p);
		temp.p = nullptr;
		temp_bytes = n;
		return temp.alloc(n);
	};
	// exclusive scan over tiles + 1 entries: the last one is the number of code points
	EX_TRY(rocprim::exclusive_scan(nullptr, need, tile_count.as<uint32_t>(), tile_count.as<uint32_t>(), 0u, tiles + 1,
	                               rocprim::plus<uint32_t>(), stream));
	EX_TRY(ensure_temp(need));
	EX_TRY(rocprim::exclusive_scan(temp.p, need, tile_count.as<uint32_t>(), tile_count.as<uint32_t>(), 0u, tiles + 1,
	                               rocprim::plus<uint32_t>(), stream));
	uint32_t N = 0, has_nul = 0;
	EX_TRY(hipMemcpyAsync(&N, tile_count.as<uint32_t>() + tiles, sizeof N, hipMemcpyDeviceToHost, stream));
	EX_TRY(hipMemcpyAsync(&has_nul, flags.p, sizeof has_nul, hipMemcpyDeviceToHost, stream));
	EX_TRY(hipStreamSynchronize(stream));
	if (has_nul) {
		if (err) *err = "extract_candidates: the corpus holds a NUL byte (documents are C strings in the reference, DataFile.cpp:27-29)";
		return 2;
	}
	if (N == 0) return 0;

	DeviceBuf S, doc_rank, win;
	EX_TRY(S.alloc(((size_t)N + 1) * sizeof(uint32_t)));
	EX_TRY(doc_rank.alloc((C.D + 1) * sizeof(uint32_t)));
	EX_TRY(win.alloc((size_t)N * sizeof(uint64_t)));
	hipLaunchKernelGGL(fill_starts_kernel, dim3(tiles), dim3(kThreads), 0, stream, C.bytes, C.total, tile_count.as<uint32_t>(),
	                   S.as<uint32_t>());
	const uint32_t total32 = (uint32_t)C.total;
	EX_TRY(hipMemcpyAsync(S.as<uint32_t>() + N, &total32, sizeof total32, hipMemcpyHostToDevice, stream));
	hipLaunchKernelGGL(doc_rank_kernel, dim3(blocks_for(C.D + 1)), dim3(kThreads), 0, stream, S.as<uint32_t>(), N, C.off, C.D,
	                   doc_rank.as<uint32_t>());
	EX_TRY(hipMemsetAsync(flags.p, 0, sizeof(uint32_t), stream)); // reused: longest window in bytes
	hipLaunchKernelGGL(window_len_kernel, dim3(blocks_for(N)), dim3(kThreads), 0, stream, S.as<uint32_t>(), N, C.off, C.D,
	                   doc_rank.as<uint32_t>(), max_cp, win.as<uint64_t>(), flags.as<uint32_t>());
	uint32_t max_bytes = 0;
	EX_TRY(hipMemcpyAsync(&max_bytes, flags.p, sizeof max_bytes, hipMemcpyDeviceToHost, stream));
	EX_TRY(hipStreamSynchronize(stream)); // (total32 is a stack variable, too)
	(void)hipFree(S.p), S.p = nullptr; // positions live on in the window records

	// 2. sort: least significant word first, every pass stable
	DeviceBuf keys_a, keys_b, perm_a, perm_b;
	EX_TRY(keys_a.alloc((size_t)N * sizeof(uint64_t)));
	EX_TRY(keys_b.alloc((size_t)N * sizeof(uint64_t)));
	EX_TRY(perm_a.alloc((size_t)N * sizeof(uint32_t)));
	EX_TRY(perm_b.alloc((size_t)N * sizeof(uint32_t)));
	uint32_t *perm = perm_a.as<uint32_t>(), *perm_alt = perm_b.as<uint32_t>();
	hipLaunchKernelGGL(iota_kernel, dim3(blocks_for(N)), dim3(kThreads), 0, stream, perm, N);
	const uint32_t words = (max_bytes + 7) / 8; // key words that hold anything: 2 for mostly-ASCII text at max_cp = 10
	EX_TRY(rocprim::radix_sort_pairs(nullptr, need, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), perm, perm_alt, N, 0, 64, stream));
	EX_TRY(ensure_temp(need));
	for (uint32_t w = words; w-- > 0;) {
		hipLaunchKernelGGL(gather_keys_kernel, dim3(blocks_for(N)), dim3(kThreads), 0, stream, C.bytes, win.as<uint64_t>(), perm, N,
		                   w, keys_a.as<uint64_t>());
		EX_TRY(rocprim::radix_sort_pairs(temp.p, need, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), perm, perm_alt, N, 0, 64,
		                                 stream));
		std::swap(perm, perm_alt);
	}
	(void)hipFree(keys_a.p), keys_a.p = nullptr;
	(void)hipFree(keys_b.p), keys_b.p = nullptr;

	// 3. lcp, token numbering
	DeviceBuf lcp, slen, inv, fresh, base, srec;
	EX_TRY(srec.alloc((size_t)N * sizeof(uint64_t)));
	EX_TRY(lcp.alloc(N));
	EX_TRY(slen.alloc(N));
	EX_TRY(inv.alloc(((size_t)N + 1) * sizeof(uint32_t)));
	EX_TRY(fresh.alloc((size_t)N * sizeof(uint32_t)));
	EX_TRY(base.alloc(((size_t)N + 1) * sizeof(uint64_t)));
	EX_TRY(hipMemsetAsync(inv.as<uint32_t>() + N, 0, sizeof(uint32_t), stream));
	hipLaunchKernelGGL(lcp_kernel, dim3(blocks_for(N)), dim3(kThreads), (words + 1) * kThreads * sizeof(uint64_t), stream, C.bytes,
	                   win.as<uint64_t>(), perm, N, words, lcp.as<uint8_t>(), slen.as<uint8_t>(), inv.as<uint32_t>(),
	                   fresh.as<uint32_t>(), srec.as<uint64_t>());
	EX_TRY(rocprim::exclusive_scan(nullptr, need, fresh.as<uint32_t>(), base.as<uint64_t>(), (uint64_t)0, N, rocprim::plus<uint64_t>(),
	                               stream));
	EX_TRY(ensure_temp(need));
	EX_TRY(rocprim::exclusive_scan(temp.p, need, fresh.as<uint32_t>(), base.as<uint64_t>(), (uint64_t)0, N, rocprim::plus<uint64_t>(),
	                               stream));
	uint64_t last_base = 0;
	uint32_t last_fresh = 0;
	EX_TRY(hipMemcpyAsync(&last_base, base.as<uint64_t>() + (N - 1), sizeof last_base, hipMemcpyDeviceToHost, stream));
	EX_TRY(hipMemcpyAsync(&last_fresh, fresh.as<uint32_t>() + (N - 1), sizeof last_fresh, hipMemcpyDeviceToHost, stream));
	EX_TRY(hipStreamSynchronize(stream));
	const uint64_t T = last_base + last_fresh;
	if (T >= kNoParent) {
		if (err) *err = "extract_candidates: 2^32 candidates or more (parent indices are 32-bit, Token.cpp:187-190)";
		return 2;
	}
	(void)hipFree(fresh.p), fresh.p = nullptr;
	(void)hipFree(win.p), win.p = nullptr;

	// 4. runs per depth
	DeviceBuf run;
	EX_TRY(run.alloc((size_t)max_cp * N * sizeof(uint32_t)));
	for (uint32_t d = 1; d <= max_cp; d++) {
		auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<uint32_t>(0), RunStart{lcp.as<uint8_t>(), d});
		uint32_t *dst = run.as<uint32_t>() + (size_t)(d - 1) * N;
		EX_TRY(rocprim::inclusive_scan(nullptr, need, in, dst, N, MaxOp(), stream));
		EX_TRY(ensure_temp(need));
		EX_TRY(rocprim::inclusive_scan(temp.p, need, in, dst, N, MaxOp(), stream));
	}

	// 5. tokens
	uint32_t *chr = nullptr, *lpar = nullptr, *rpar = nullptr;
	uint64_t *uses = nullptr;
	hipError_t e = hipMalloc((void **)&chr, T * sizeof(uint32_t));
	if (e == hipSuccess) e = hipMalloc((void **)&lpar, T * sizeof(uint32_t));
	if (e == hipSuccess) e = hipMalloc((void **)&rpar, T * sizeof(uint32_t));
	if (e == hipSuccess) e = hipMalloc((void **)&uses, T * sizeof(uint64_t));
	if (e == hipSuccess) {
		hipLaunchKernelGGL(run_uses_kernel, dim3(blocks_for((uint64_t)N + 1)), dim3(kThreads), 0, stream, run.as<uint32_t>(),
		                   lcp.as<uint8_t>(), slen.as<uint8_t>(), base.as<uint64_t>(), N, max_cp, uses);
		hipLaunchKernelGGL(fill_tokens_kernel, dim3(blocks_for(N)), dim3(kThreads), 0, stream, C.bytes, srec.as<uint64_t>(), perm,
		                   inv.as<uint32_t>(), run.as<uint32_t>(), lcp.as<uint8_t>(), slen.as<uint8_t>(), base.as<uint64_t>(), N, chr, lpar,
		                   rpar);
		e = hipGetLastError();
		if (e == hipSuccess) e = hipStreamSynchronize(stream);
	}
	if (e != hipSuccess) {
		if (chr) (void)hipFree(chr);
		if (lpar) (void)hipFree(lpar);
		if (rpar) (void)hipFree(rpar);
		if (uses) (void)hipFree(uses);
		if (err) *err = std::string("extract_candidates: ") + hipGetErrorString(e);
		return 1;
	}
	out->chr = chr, out->lpar = lpar, out->rpar = rpar, out->uses = uses, out->n = T, out->positions = N;
	return 0;
}

void free_candidates(CandidateArraysDev *c) {
	if (c->chr) (void)hipFree(c->chr);
	if (c->lpar) (void)hipFree(c->lpar);
	if (c->rpar) (void)hipFree(c->rpar);
	if (c->uses) (void)hipFree(c->uses);
	*c = CandidateArraysDev{};
}

} // namespace tokscore
