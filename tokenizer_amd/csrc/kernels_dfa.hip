// kernels_dfa.hip -- round 3's full-evaluation scan: the vocabulary as a one-transition-per-byte
// automaton (dfa.h), every lane of the grid stepping C independent ranges of the corpus in lockstep.
//
// What changed against scan_ranges_kernel (kernels_stream.hip): that kernel walks the trie as
// SolutionFile::Tokenize does (/root/reference/src/files/SolutionFile.cpp:89-97) -- down as far as
// the bytes go, back to the longest token end, again from there -- so a lane has a position of its
// own, a ring of bytes in LDS, the longest end so far and its slot to keep, and 1.15 (bench
// vocabulary) to 3.4 (the reference annealer's vocabulary) lookups per byte.  Here a lane never
// goes back: the automaton's state says what the bytes since the last decided token end spell, a
// miss in the state's row emits the tokens that are decided by it (dfa.h) and the next state follows
// from the byte alone (the depth-1 rows sit at code * stride).  Exactly one slot read per byte, for
// every lane, whatever the text: the 64 lanes of a wave stay in step, so the bytes come from
// registers (a dword of the transposed corpus per four steps, no ring, no position), ranges all end
// together (no per-lane range test) and a wave can keep C ranges per lane in flight to cover the
// LDS round trip.  LDS holds nothing but the hot-first prefix of the image; a lookup beyond it is
// one masked global load (L2).
//
// Counting: a step logs an EVENT -- the 16-bit slot whose state was left by a miss (miss stream),
// or the virtual slot arrived on (arrival stream) -- exactly like the slot stream of the ranges
// scan; hist_slots_kernel bins them and the reduction turns slots into ids through the event
// tables.  Events on slots beyond the LDS prefix are rare by construction (hot-first order) and go
// to per-slot counters in global memory.
//
// Range boundaries: lane g starts in the state the previous evaluation found range g - 1 to end in
// (the separator's state the first time) and writes the state it ends in; where exit[g - 1] !=
// entry[g] -- everywhere in a plan's first evaluation, nowhere afterwards while the vocabulary
// stays as it is -- dfa_fixup_kernel steps both automata (the true one +1, the counted one -1)
// over range g until their states agree.
#include "kernels.h"
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>

namespace tokscore {

namespace {

constexpr uint32_t kDfaBatchEntries = kStreamBatch * 512; // u16 entries of one stream batch (32 KiB)
constexpr uint32_t kArr = 0x80000000u, kPartial = 0x40000000u, kBaseMask = 0xFFFFu, kLabelMask = 0xFFu, kLabelShift = 16; // (dfa.h: DfaImage)
constexpr uint32_t kRootFail = 0x20000000u, kStateMask = kBaseMask | kPartial | kRootFail;
constexpr uint32_t kEvNone = 0xFFFFFFFFu, kEvList = 0x80000000u;

// One transition, as every path here takes it (the scan's loop in assembly does the same: tools/gen_dfa_asm.py).
// S: state word, G: the state's global base while S is under cold_base.  Returns true for a hit; *id = the
// entry arrived on (either array), *key = the key of the state left (what a miss counts under).
struct DfaStep {
	uint32_t x, id, g_next;
	bool hit, second_look;
};
template <typename HotArray>
__device__ __forceinline__ DfaStep dfa_step(const DfaDev &A, HotArray hot, uint32_t S, uint32_t G, uint32_t c) {
	DfaStep r;
	const uint32_t b = S & kBaseMask;
	r.id = b + c; // (always inside the LDS array)
	r.x = hot[r.id];
	r.g_next = G;
	r.hit = ((r.x >> kLabelShift) & kLabelMask) == c;
	r.second_look = !r.hit && (S & kPartial) != 0;
	return r;
}
// LDS does not hold the state's row completely: its copy in the global array decides
__device__ __forceinline__ void dfa_second_look(const DfaDev &A, DfaStep &r, uint32_t S, uint32_t G, uint32_t c) {
	const uint32_t b = S & kBaseMask;
	r.id = (b == A.cold_base ? G : A.gof[b]) + c;
	const uint2 w = A.slot[r.id];
	r.x = w.x;
	r.hit = ((r.x >> kLabelShift) & kLabelMask) == c;
	if (r.hit) r.g_next = w.y;
}
__device__ __forceinline__ uint32_t dfa_key(const DfaDev &A, uint32_t S, uint32_t G) {
	const uint32_t b = S & kBaseMask;
	return b == A.cold_base ? G : b;
}

// The LDS array from its copy in global memory (every workgroup reads the same 160 KB: L2 after the first).  All of a
// thread's loads are under way before its first store: one round trip instead of ten (a loop of load, wait, store
// was 3 - 4 us of a 56 us scan; blockDim.x >= 1024 covers 40,832 slots with ten 16-byte loads per thread).
__device__ __forceinline__ void dfa_fill_lds(const uint32_t *hot, uint32_t hot_slots, LDS_AS uint32_t *img) {
	const u32x4 *src = reinterpret_cast<const u32x4 *>(hot); // (the array is padded to 16 bytes)
	LDS_AS u32x4 *dst = (LDS_AS u32x4 *)img;
	const uint32_t n = (hot_slots + 3) / 4;
	constexpr int kInFlight = 10;
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

// The ranges transposed: 8-byte piece p of chain j of lane l of wave block wb at
// ((wb * pieces + p) * C + j) * 64 + l (in pieces); range g = (wb * C + j) * 64 + l.
__global__ void __launch_bounds__(256) dfa_transpose_kernel(const uint8_t *prep, uint64_t prep_total, uint32_t range_bytes,
                                                            uint32_t chains, uint64_t nwb, uint2 *tcodes) {
	const uint64_t pieces = range_bytes / 8;
	const uint64_t n = nwb * pieces * chains * 64;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t l = i % 64;
		const uint64_t j = (i / 64) % chains;
		const uint64_t p = (i / (64 * chains)) % pieces;
		const uint64_t wb = i / (64 * chains * pieces);
		const uint64_t src = ((wb * chains + j) * 64 + l) * range_bytes + 8 * p;
		uint32_t w[2] = {0, 0};
		if (src + 8 <= prep_total) {
			const uint2 v = *reinterpret_cast<const uint2 *>(prep + src); // (range_bytes is a multiple of 8, prep 16-byte aligned)
			w[0] = v.x, w[1] = v.y;
		} else {
			for (uint32_t k = 0; k < 8; k++)
				if (src + k < prep_total) w[k >> 2] |= (uint32_t)prep[src + k] << (8 * (k & 3));
		}
		tcodes[i] = uint2{w[0], w[1]};
	}
}

// entry states of a new plan: the separator's state everywhere; cursors and fix-up words
__global__ void __launch_bounds__(256) dfa_reset_states_kernel(uint2 *entry, uint32_t nranges, uint32_t start_word) {
	for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < nranges; g += gridDim.x * blockDim.x)
		entry[g] = uint2{start_word & kStateMask, 0u}; // (failure form: entered through slot 0, the root row's entry of the separator)
}

__global__ void __launch_bounds__(256) dfa_prepare_kernel(unsigned long long *out, uint32_t V, unsigned long long D,
                                                          unsigned long long total_bytes, uint32_t *cursor_m, uint32_t *cursor_a,
                                                          uint32_t first_batches, uint32_t *changed) {
	const uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x;
	if (out)
		for (uint32_t i = i0; i < V + 5; i += gridDim.x * blockDim.x) out[i] = i < 2 ? D : (i == V + 4 ? total_bytes : 0ull);
	if (i0 == 0) {
		cursor_m[0] = first_batches, cursor_m[1] = 0;
		cursor_a[0] = first_batches, cursor_a[1] = 0;
		changed[0] = changed[1] = 0;
	}
}

// A wave's place in a stream: one batch at a time, the lanes that log in a step store their 16-bit
// values at consecutive positions (kernels_stream.hip: DirectLog)
struct EventLog {
	uint16_t *base;
	uint32_t cur; // entries
	uint32_t batch;
	__device__ __forceinline__ void open_own(const StreamDev &S, uint32_t wave_id) {
		batch = wave_id;
		base = reinterpret_cast<uint16_t *>(S.blocks) + (size_t)batch * kDfaBatchEntries;
		cur = 0;
	}
	__device__ __forceinline__ void roll(const StreamDev &S, uint32_t lane) {
		if (lane == 0) S.fill[batch] = cur;
		uint32_t b = 0;
		if (lane == 0) b = atomicAdd(&S.cursor[0], 1u);
		batch = __builtin_amdgcn_readfirstlane(b);
		if (batch >= S.capacity_batches) { // cannot happen (one entry per byte at most); reported all the same
			if (lane == 0) S.cursor[1] = 1u;
			batch = 0;
		}
		base = reinterpret_cast<uint16_t *>(S.blocks) + (size_t)batch * kDfaBatchEntries;
		cur = 0;
	}
	__device__ __forceinline__ void close(const StreamDev &S, uint32_t lane) {
		if (lane == 0) S.fill[batch] = cur;
	}
	__device__ __forceinline__ void put(bool pred, uint32_t val) { // wave-uniform call
		const unsigned long long m = __builtin_amdgcn_ballot_w64(pred);
		if (pred) {
			const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
			base[cur + rank] = (uint16_t)val;
		}
		cur += (uint32_t)__builtin_popcountll(m);
	}
};

template <int C>
__global__ void __launch_bounds__(1024) dfa_scan_kernel(DfaDev A, DfaWork W, StreamDev Qm, StreamDev Qa, OutInit I) {
	extern __shared__ __attribute__((aligned(16))) uint32_t dfa_smem[];
	LDS_AS uint32_t *img = (LDS_AS uint32_t *)dfa_smem;
	if (I.out) // (nothing else touches the output vector before this kernel has ended)
		for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < I.V + 5; i += gridDim.x * blockDim.x)
			I.out[i] = i < 2 ? I.D : (i == I.V + 4 ? I.total_bytes : 0ull);
	dfa_fill_lds(A.hot, A.hot_slots, img);
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = uni32(threadIdx.x >> 6);
	const uint32_t waves = blockDim.x >> 6;
	const uint32_t wave_id = blockIdx.x * waves + wave;
	const uint32_t total_waves = gridDim.x * waves;
	EventLog lm, la;
	lm.open_own(Qm, wave_id);
	la.open_own(Qa, wave_id);
	const uint32_t pieces = W.range_bytes / 8;
	const uint32_t hot_slots = A.hot_slots, stride = A.stride;
	constexpr uint32_t kHigh = kDfaBatchEntries - 8 * C * 64;

	for (uint32_t wb = wave_id; wb < W.nwb; wb += total_waves) {
		uint32_t S[C], G[C];
#pragma unroll
		for (int j = 0; j < C; j++) {
			const uint2 e = W.entry[(wb * C + j) * 64 + lane];
			S[j] = e.x, G[j] = e.y;
		}
		const uint2 *src = W.tcodes + (size_t)wb * pieces * C * 64 + lane;
		uint2 cur[C], nxt[C];
#pragma unroll
		for (int j = 0; j < C; j++) cur[j] = src[j * 64], nxt[j] = cur[j];
		for (uint32_t p = 0; p < pieces; p++) {
			if (p + 1 < pieces) {
#pragma unroll
				for (int j = 0; j < C; j++) nxt[j] = src[((size_t)(p + 1) * C + j) * 64];
			}
#pragma unroll
			for (int k = 0; k < 8; k++) {
#pragma unroll
				for (int j = 0; j < C; j++) {
					const uint32_t w = k < 4 ? cur[j].x : cur[j].y;
					const uint32_t c = (w >> (8 * (k & 3))) & 0xFFu;
					DfaStep r = dfa_step(A, img, S[j], G[j], c);
					if (__builtin_amdgcn_ballot_w64(r.second_look) != 0) {
						if (r.second_look) {
							dfa_second_look(A, r, S[j], G[j], c);
							if (r.hit) atomicAdd(&W.cold_hit[r.id - hot_slots], 1u);
						}
					}
					const bool arr = r.hit && (r.x & kArr) != 0;
					// the state left by a miss: its base to the miss stream (cold_base counts for nothing: such a state has a counter)
					lm.put(!r.hit, S[j] & kBaseMask);
					if (!r.hit && (S[j] & kBaseMask) == A.cold_base) atomicAdd(&W.cold_miss[G[j] - hot_slots], 1u);
					if (__builtin_amdgcn_ballot_w64(arr) != 0) {
						la.put(arr && r.id < hot_slots, r.id);
						if (arr && r.id >= hot_slots) atomicAdd(&W.cold_arr[r.id - hot_slots], 1u);
					}
					S[j] = r.hit ? r.x : c * stride;
					G[j] = r.g_next;
				}
			}
#pragma unroll
			for (int j = 0; j < C; j++) cur[j] = nxt[j];
			if (lm.cur > kHigh) lm.roll(Qm, lane);
			if (la.cur > kHigh) la.roll(Qa, lane);
		}
#pragma unroll
		for (int j = 0; j < C; j++) W.exit_a[(wb * C + j) * 64 + lane] = uint2{S[j] & (kBaseMask | kPartial), G[j]};
	}
	lm.close(Qm, lane);
	la.close(Qa, lane);
}

// ---- the failure form (dfa.h: DfaImage::fail_code) ---------------------------------------------
// For vocabularies whose prefixes are mostly not tokens (the reference annealer's own output,
// /root/reference/src/tokenizer/TokenGenerator.cpp:190-217): their flattened rows run into millions of virtual
// entries, so the image holds the trie's own edges and every state's failure entry instead, and f is
// followed state by state -- a miss emits F(s) and leads to f(s) WITHOUT consuming the byte, unless f(s) is
// the root.  That is 1.5 lookups per byte instead of one, and the lanes of a wave no longer take their bytes
// at the same pace, so a lane keeps a window of its own: up to 16 bytes of its range in four registers, the
// byte in front taken off by a shift.  The wave still loads in step -- piece p of every lane is appended once
// every lane has room for it, a lane that is ahead runs on into the piece it already holds -- so the loads
// stay whole cache lines of the transposed corpus, asked for a round early.
//   A state is (word, T): T = the slot of the entry it was entered through, in either array.  Events are
// keyed by T (stream for T in LDS, a counter beyond), the global copy of a state's row starts at gof[T].
// LDS answers what it holds; a miss in a row that is not there completely, and a failure entry that is not
// there, are looked up in the global array (L2) by the lanes concerned while the others wait.
template <int C, bool HITS>
__global__ void __launch_bounds__(1024) dfa_fscan_kernel(DfaDev A, DfaWork W, StreamDev Qm, StreamDev Qa, OutInit I, uint32_t dbg) {
	extern __shared__ __attribute__((aligned(16))) uint32_t dfa_f_smem[];
	LDS_AS uint32_t *img = (LDS_AS uint32_t *)dfa_f_smem;
	if (I.out) // (nothing else touches the output vector before this kernel has ended)
		for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < I.V + 5; i += gridDim.x * blockDim.x)
			I.out[i] = i < 2 ? I.D : (i == I.V + 4 ? I.total_bytes : 0ull);
	dfa_fill_lds(A.hot, A.hot_slots, img);
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = uni32(threadIdx.x >> 6);
	const uint32_t waves = blockDim.x >> 6;
	const uint32_t wave_id = blockIdx.x * waves + wave;
	const uint32_t total_waves = gridDim.x * waves;
	EventLog lm;
	lm.open_own(Qm, wave_id);
	if (lane == 0) Qa.fill[wave_id] = 0; // (no arrivals in this form: the wave's own batch of that stream stays empty)
	const uint32_t pieces = uni32(W.range_bytes / 8);
	const uint32_t hot_slots = A.hot_slots, stride = A.stride, failc = A.fail_code;
	constexpr uint32_t kHigh = kDfaBatchEntries - C * 64 - 8;

	const uint2 *gslot = A.slot;
	const uint32_t *gof = A.gof;
	auto label = [](uint32_t w) { return (w >> kLabelShift) & kLabelMask; };
	for (uint32_t wb = wave_id; wb < W.nwb; wb += total_waves) {
		// a range's state: word, T = the entry it was entered through, G = its row's global base while T is an entry of the
		// global array; its window: w0 .. w3, `avail` bytes of the range from the low end of w0 on
		uint32_t S[C], T[C], G[C], avail[C], w0[C], w1[C], w2[C], w3[C];
		uint2 pre[C];
		const uint2 *src = W.tcodes + (size_t)wb * pieces * C * 64 + lane;
#pragma unroll
		for (int j = 0; j < C; j++) {
			const uint2 e = W.entry[(wb * C + j) * 64 + lane];
			S[j] = e.x, T[j] = e.y;
			G[j] = T[j] >= hot_slots ? gof[T[j]] : 0u;
			const uint2 p0 = src[(size_t)j * 64];
			w0[j] = p0.x, w1[j] = p0.y, w2[j] = w3[j] = 0, avail[j] = 8;
			pre[j] = pieces > 1 ? src[((size_t)C + j) * 64] : uint2{0u, 0u};
		}
		uint32_t next = 1; // the piece in `pre`
		for (;;) {
			const bool more = next < pieces; // (uniform)
			const uint32_t th = more ? 8u : 0u;
			// until every range of every lane has room for the next piece (has taken all its bytes, at the end)
			for (;;) {
				bool any = false;
#pragma unroll
				for (int j = 0; j < C; j++) any = any || avail[j] > th;
				if (__builtin_amdgcn_ballot_w64(any) == 0) break;
				// One step of every range, in stages: what LDS says for all of them, then the global array's answers for
				// all of them in ONE round trip (the loads of the C ranges are under way together), then the outcome.
				// No branch per range: a lane that needs nothing from the global array reads its slot 0 (one line for
				// all such lanes) and drops it.
				uint32_t c[C], id[C], idf[C], x[C], xf[C];
				bool act[C], hit[C], cold[C], cold_f[C];
				bool any_cold = false;
#pragma unroll
				for (int j = 0; j < C; j++) {
					c[j] = w0[j] & 0xFFu;
					const uint32_t b = S[j] & kBaseMask;
					id[j] = b + c[j], idf[j] = b + failc; // (both inside the LDS array)
					x[j] = img[id[j]], xf[j] = img[idf[j]];
				}
#pragma unroll
				for (int j = 0; j < C; j++) {
					act[j] = avail[j] != 0;
					hit[j] = label(x[j]) == c[j];
					// LDS does not hold the row completely: its global copy decides -- the byte's slot and, for a state
					// whose failure entry LDS does not hold either, that entry's, asked for together
					cold[j] = act[j] && !hit[j] && (S[j] & kPartial) != 0;
					cold_f[j] = cold[j] && (S[j] & kRootFail) == 0 && label(xf[j]) != failc;
					any_cold = any_cold || cold[j];
				}
				if (!(dbg & 4) && __builtin_amdgcn_ballot_w64(any_cold) != 0) {
					uint32_t g[C];
					uint2 w[C], wf[C];
#pragma unroll
					for (int j = 0; j < C; j++) { // the row's global base: came with the entry when that was a global one, else gof[T]
						const bool need = cold[j] && T[j] < hot_slots;
						const uint32_t gl = gof[need ? T[j] : 0u];
						g[j] = need ? gl : G[j];
					}
#pragma unroll
					for (int j = 0; j < C; j++) {
						w[j] = gslot[cold[j] ? g[j] + c[j] : 0u];
						wf[j] = gslot[cold_f[j] ? g[j] + failc : 0u];
					}
#pragma unroll
					for (int j = 0; j < C; j++) {
						const bool ghit = cold[j] && label(w[j].x) == c[j];
						const bool usef = cold_f[j] && !ghit;
						x[j] = ghit ? w[j].x : x[j], id[j] = ghit ? g[j] + c[j] : id[j];
						hit[j] = hit[j] || ghit;
						xf[j] = usef ? wf[j].x : xf[j], idf[j] = usef ? g[j] + failc : idf[j];
						G[j] = ghit ? w[j].y : usef ? wf[j].y : G[j];
						if (HITS) {
							if (ghit || usef) atomicAdd(&W.cold_hit[(ghit ? id[j] : idf[j]) - hot_slots], 1u);
						}
					}
				}
#pragma unroll
				for (int j = 0; j < C; j++) {
					const bool miss = act[j] && !hit[j];
					const bool root = (S[j] & kRootFail) != 0;
					if (!(dbg & 2)) lm.put(miss && T[j] < hot_slots, T[j]);
					if (!(dbg & 1) && miss && T[j] >= hot_slots) atomicAdd(&W.cold_miss[T[j] - hot_slots], 1u);
					const uint32_t s_new = hit[j] ? x[j] : root ? (c[j] * stride | kRootFail) : xf[j];
					const uint32_t t_new = hit[j] ? id[j] : root ? c[j] : idf[j];
					S[j] = act[j] ? s_new : S[j];
					T[j] = act[j] ? t_new : T[j];
					const uint32_t n = act[j] && (hit[j] || root) ? 1u : 0u; // the byte is taken, or stays for f's state
					w0[j] = __builtin_amdgcn_alignbyte(w1[j], w0[j], n);
					w1[j] = __builtin_amdgcn_alignbyte(w2[j], w1[j], n);
					w2[j] = __builtin_amdgcn_alignbyte(w3[j], w2[j], n);
					w3[j] = __builtin_amdgcn_alignbyte(0u, w3[j], n);
					avail[j] -= n;
				}
				if (lm.cur > kHigh) lm.roll(Qm, lane);
			}
			if (!more) break;
#pragma unroll
			for (int j = 0; j < C; j++) { // avail <= 8: the piece goes behind what is left
				const unsigned long long p = (unsigned long long)pre[j].y << 32 | pre[j].x;
				unsigned long long lo = (unsigned long long)w1[j] << 32 | w0[j], hi;
				const uint32_t sh = 8 * avail[j];
				if (avail[j] == 8) hi = p;
				else {
					lo |= p << sh;
					hi = sh ? p >> (64 - sh) : 0ull;
				}
				w0[j] = (uint32_t)lo, w1[j] = (uint32_t)(lo >> 32), w2[j] = (uint32_t)hi, w3[j] = (uint32_t)(hi >> 32);
				avail[j] += 8;
			}
			next++;
			if (next < pieces) {
#pragma unroll
				for (int j = 0; j < C; j++) pre[j] = src[((size_t)next * C + j) * 64];
			}
		}
#pragma unroll
		for (int j = 0; j < C; j++) W.exit_a[(wb * C + j) * 64 + lane] = uint2{S[j] & kStateMask, T[j]};
	}
	lm.close(Qm, lane);
}

// ---- the scan's loop, hand-written for gfx950 -------------------------------------------------
// dfa_scan_loop.inc is written by tools/gen_dfa_asm.py: the loop over the pieces of one wave block
// for 1, 2 and 4 chains per lane, the chains interleaved instruction by instruction (exec all ones
// throughout; lanes without an event store to a trash entry at the end of the batch; scalar masks per
// chain; rare paths -- the global array, event indices beyond LDS, arrivals -- out of line).  The
// schedule, the register plan and the reasons are in the generator.
#include "dfa_scan_loop.inc"

template <int C>
__global__ void __launch_bounds__(1024) dfa_scan_asm_kernel(DfaDev A, DfaWork W, StreamDev Qm, StreamDev Qa, OutInit I) {
	extern __shared__ __attribute__((aligned(16))) uint32_t dfa_asm_smem[];
	LDS_AS uint32_t *img = (LDS_AS uint32_t *)dfa_asm_smem;
	if (I.out) // (nothing else touches the output vector before this kernel has ended)
		for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < I.V + 5; i += gridDim.x * blockDim.x)
			I.out[i] = i < 2 ? I.D : (i == I.V + 4 ? I.total_bytes : 0ull);
	dfa_fill_lds(A.hot, A.hot_slots, img);
	__syncthreads();
	if ((uint32_t)(uintptr_t)img != 0) {
		// the loop takes 4 * slot index for an LDS address (the array starts at LDS address 0: this kernel
		// has no static LDS); reported like a stream overflow, never ignored
		if (threadIdx.x == 0) Qm.cursor[1] = 1u;
		return;
	}
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = uni32(threadIdx.x >> 6);
	const uint32_t waves = blockDim.x >> 6;
	const uint32_t wave_id = blockIdx.x * waves + wave;
	const uint32_t total_waves = gridDim.x * waves;
	EventLog lm, la;
	lm.open_own(Qm, wave_id);
	la.open_own(Qa, wave_id);
	const uint32_t pieces = uni32(W.range_bytes / 8);
	const uint32_t hot_slots = uni32(A.hot_slots), stride = A.stride;
	// (a step's store also writes the entry behind its events: one more than a piece's events must fit)
	constexpr uint32_t kHighBytes = (kDfaBatchEntries - 8 - 8 * C * 64) * 2;
	const unsigned long long slot_p = uni64((unsigned long long)(uintptr_t)A.slot), gof_p = uni64((unsigned long long)(uintptr_t)A.gof);
	const unsigned long long cmiss_p = uni64((unsigned long long)(uintptr_t)W.cold_miss), carr_p = uni64((unsigned long long)(uintptr_t)W.cold_arr);
	const unsigned long long chit_p = uni64((unsigned long long)(uintptr_t)W.cold_hit);
	const uint32_t cold_base = A.cold_base, cold_base_s = uni32(A.cold_base);

	for (uint32_t wb = wave_id; wb < W.nwb; wb += total_waves) {
		uint32_t S[4] = {0, 0, 0, 0}, G[4] = {0, 0, 0, 0};
#pragma unroll
		for (int j = 0; j < C; j++) {
			const uint2 e = W.entry[(wb * C + j) * 64 + lane];
			S[j] = e.x, G[j] = e.y;
		}
		const unsigned long long src = uni64((unsigned long long)(uintptr_t)(W.tcodes + (size_t)wb * pieces * C * 64));
		uint32_t p = 0;
		while (p < pieces) {
			uint32_t voff = lane * 8 + p * (C * 512);
			uint32_t curm = uni32(lm.cur * 2), cura = uni32(la.cur * 2);
			const unsigned long long basem = uni64((unsigned long long)(uintptr_t)lm.base), basea = uni64((unsigned long long)(uintptr_t)la.base);
			// the miss batch as a buffer of 16-bit records (stride 2): a store's index is the lane's rank among the step's
			// events, its scalar offset the fill in bytes; what lies beyond the batch is dropped by the range check
			const u32x4 srd = {uni32((uint32_t)basem), uni32((uint32_t)(basem >> 32) & 0xFFFFu) | (2u << 16), kDfaBatchEntries, 0x00020000u};
			uint32_t pp = uni32(p);
#define DFA_OPERANDS                                                                                                         \
	[p] "+s"(pp), [curm] "+s"(curm), [cura] "+s"(cura), [voff] "+v"(voff), [S0] "+v"(S[0]), [G0] "+v"(G[0]), [S1] "+v"(S[1]),         \
	    [G1] "+v"(G[1]), [S2] "+v"(S[2]), [G2] "+v"(G[2]), [S3] "+v"(S[3]), [G3] "+v"(G[3])                                           \
	    : [src] "s"(src), [pieces] "s"(pieces), [hot] "s"(hot_slots), [high] "s"(kHighBytes), [srd] "s"(srd),                        \
	      [basea] "s"(basea), [slot] "s"(slot_p), [gof] "s"(gof_p), [cmiss] "s"(cmiss_p), [carr] "s"(carr_p), [chit] "s"(chit_p),    \
	      [stride] "v"(stride), [coldb] "v"(cold_base), [coldbs] "s"(cold_base_s)
			if constexpr (C == 1) {
				asm volatile(DFA_GEN_LOOP1 : DFA_OPERANDS : DFA_GEN_CLOBBERS1);
			} else if constexpr (C == 2) {
				asm volatile(DFA_GEN_LOOP2 : DFA_OPERANDS : DFA_GEN_CLOBBERS2);
			} else {
				asm volatile(DFA_GEN_LOOP4 : DFA_OPERANDS : DFA_GEN_CLOBBERS4);
			}
#undef DFA_OPERANDS
			p = pp;
			lm.cur = curm / 2, la.cur = cura / 2;
			if (lm.cur * 2 > kHighBytes) lm.roll(Qm, lane);
			if (la.cur * 2 > kHighBytes) la.roll(Qa, lane);
		}
#pragma unroll
		for (int j = 0; j < C; j++) W.exit_a[(wb * C + j) * 64 + lane] = uint2{S[j] & (kBaseMask | kPartial), G[j]};
	}
	lm.close(Qm, lane);
	la.close(Qa, lane);
}

// ---- one-pass search of a batch of strings (tokscore_score_toggles) -----------------------------------
// The scan's loop on the search automaton of dfa.h (BuildSearchImage): every range starts in the start
// state, nothing is logged but arrivals -- (entry, prepared offset of the byte) in 8 bytes -- and there is
// no global array.  A string that crosses a range boundary is not seen here: search_items_kernel's last
// workgroups read those few bytes again.
template <int C>
__global__ void __launch_bounds__(1024) dfa_search_kernel(SearchDev A, DfaWork W, StreamDev Qa, const uint8_t *codes) {
	extern __shared__ __attribute__((aligned(16))) uint32_t dfa_search_smem[];
	LDS_AS uint32_t *img = (LDS_AS uint32_t *)dfa_search_smem;
	dfa_fill_lds(A.hot, A.hot_slots, img);
	__syncthreads();
	if ((uint32_t)(uintptr_t)img != 0) { // (as dfa_scan_asm_kernel: the loop takes 4 * slot index for an LDS address)
		if (threadIdx.x == 0) Qa.cursor[1] = 1u;
		return;
	}
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = uni32(threadIdx.x >> 6);
	const uint32_t waves = blockDim.x >> 6;
	const uint32_t wave_id = blockIdx.x * waves + wave;
	const uint32_t total_waves = gridDim.x * waves;
	EventLog la;
	la.open_own(Qa, wave_id);
	const uint32_t pieces = uni32(W.range_bytes / 8);
	const uint32_t stride = A.stride;
	// (a record is 8 bytes: a piece's steps add at most 8 * C * 64 of them)
	constexpr uint32_t kHighBytes = kDfaBatchEntries * 2 - 8 * C * 64 * 8;
	static_assert(kDfaBatchEntries * 2 >= 2 * 8 * C * 64 * 8, "a batch holds two pieces of records at least");
	for (uint32_t wb = wave_id; wb < W.nwb; wb += total_waves) {
		uint32_t S[4] = {0, 0, 0, 0}, G[4] = {0, 0, 0, 0}, P[4] = {0, 0, 0, 0};
#pragma unroll
		for (int j = 0; j < C; j++) P[j] = ((wb * C + j) * 64 + lane) * W.range_bytes; // (the prepared corpus is below 4 GiB)
		const unsigned long long src = uni64((unsigned long long)(uintptr_t)(W.tcodes + (size_t)wb * pieces * C * 64));
		uint32_t p = 0;
		while (p < pieces) {
			uint32_t voff = lane * 8 + p * (C * 512);
			uint32_t curm = 0, cura = uni32(la.cur * 2);
			const unsigned long long basea = uni64((unsigned long long)(uintptr_t)la.base);
			uint32_t pp = uni32(p);
#define DFA_SEARCH_OPERANDS                                                                                                  \
	[p] "+s"(pp), [curm] "+s"(curm), [cura] "+s"(cura), [voff] "+v"(voff), [S0] "+v"(S[0]), [G0] "+v"(G[0]), [S1] "+v"(S[1]),         \
	    [G1] "+v"(G[1]), [S2] "+v"(S[2]), [G2] "+v"(G[2]), [S3] "+v"(S[3]), [G3] "+v"(G[3])                                           \
	    : [src] "s"(src), [pieces] "s"(pieces), [high] "s"(kHighBytes), [basea] "s"(basea), [stride] "v"(stride), [coldb] "v"(0xFFFFu), \
	      [P0] "v"(P[0]), [P1] "v"(P[1]), [P2] "v"(P[2]), [P3] "v"(P[3])
			if constexpr (C == 1) {
				asm volatile(DFA_GEN_SEARCH1 : DFA_SEARCH_OPERANDS : DFA_GEN_CLOBBERS1);
			} else if constexpr (C == 2) {
				asm volatile(DFA_GEN_SEARCH2 : DFA_SEARCH_OPERANDS : DFA_GEN_CLOBBERS2);
			} else {
				asm volatile(DFA_GEN_SEARCH4 : DFA_SEARCH_OPERANDS : DFA_GEN_CLOBBERS4);
			}
#undef DFA_SEARCH_OPERANDS
			p = pp;
			la.cur = cura / 2;
			if (la.cur * 2 > kHighBytes) la.roll(Qa, lane);
		}
		// the strings that begin in a range and end in the next: every lane goes on from the state it ended in, over
		// the next range's first max_len - 1 bytes (from the linear copy), and logs the arrivals with how far behind
		// the boundary they are -- whoever reads the record keeps the strings longer than that
		if (A.max_len >= 2) {
			const uint32_t reach = uni32(A.max_len - 1);
#pragma unroll
			for (int j = 0; j < C; j++) {
				const uint2 *next = reinterpret_cast<const uint2 *>(codes + (size_t)P[j] + W.range_bytes); // (8-byte aligned; reads as separators behind the corpus)
				uint32_t Sj = S[j];
				uint2 w = next[0];
				for (uint32_t i = 0; i < reach; i++) {
					if (i != 0 && (i & 7u) == 0) w = next[i >> 3];
					const uint32_t c = ((i & 4u ? w.y : w.x) >> (8 * (i & 3u))) & 0xFFu, t = (Sj & kBaseMask) + c;
					const uint32_t x = img[t];
					const bool hit = ((x >> kLabelShift) & kLabelMask) == c;
					Sj = hit ? x : c * stride;
					const bool arr = hit && (x & kArr) != 0;
					const unsigned long long m = __builtin_amdgcn_ballot_w64(arr);
					if (m != 0) {
						if (arr) {
							const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
							reinterpret_cast<uint2 *>(la.base)[la.cur / 4 + rank] = uint2{t | ((x >> 24) & 63u) << 16 | (i + 1) << 24, P[j] + W.range_bytes + i};
						}
						la.cur += 4 * (uint32_t)__builtin_popcountll(m);
						if (la.cur * 2 > kHighBytes) la.roll(Qa, lane);
					}
				}
			}
		}
	}
	la.close(Qa, lane);
}

// (string, stream) -> the table's slot for it, claimed on first use.  Returns the slot when this call claimed it.
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
// does a token of the vocabulary as it is start in (q, q + len]?  (positions from the corpus' end on count as starts)
__device__ __forceinline__ bool search_start_within(const SearchItems &H, uint32_t q, uint32_t len) {
	const uint32_t lo = q + 1, hi = q + len; // inclusive
	if (hi >= H.prep_total) return true;
	for (uint32_t w = lo >> 6; w <= hi >> 6; w++) {
		unsigned long long bits = H.starts[w];
		if (w == lo >> 6) bits &= ~0ull << (lo & 63);
		if (w == hi >> 6) bits &= ~0ull >> (63 - (hi & 63));
		if (bits) return true;
	}
	return false;
}
__device__ __forceinline__ uint32_t search_emit(const SearchItems &H, uint32_t string, uint32_t q, uint32_t len) {
	if (!((H.starts[q >> 6] >> (q & 63)) & 1ull)) return kNoSlot; // inside a token of the vocabulary as it is: nothing can part there
	// the token that starts at q is longer than the string: it is taken with the string's end flipped as well (nine of
	// ten occurrences on a token start, on the bench shard)
	if (!search_start_within(H, q, len)) return kNoSlot;
	uint32_t l = H.block_stream[q >> 8];
	while (l + 1 < H.nstreams && H.pbeg[l + 1] <= q) l++;
	const uint32_t rel = q - H.pbeg[l];
	const unsigned long long key = ((unsigned long long)string << 32 | l) + 1ull;
	uint32_t h = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 32) & H.mask;
	uint32_t claimed = kNoSlot;
	for (uint32_t tries = 0;; tries++) {
		// (a look first: a slot somebody else's key holds costs no atomic -- device-wide atomics are what this kernel
		// waits for.  A key is written once, over 0: a stale 0 only sends the exchange, which then returns what is there)
		unsigned long long seen = *(volatile unsigned long long *)&H.keys[h];
		if (seen == 0) {
			seen = atomicCAS(&H.keys[h], 0ull, key);
			if (seen == 0) {
				H.items[h].stream = l, H.items[h].move = string;
				claimed = h;
				seen = key;
			}
		}
		if (seen == key) break;
		h = (h + 1) & H.mask;
		if (tries > 256) { // (the caller sizes the table at four slots per item and more)
			H.flags[0] = 1u;
			return kNoSlot;
		}
	}
	// (device-wide atomics are what this kernel waits for: whoever claimed the slot takes offset 0 without one --
	// `count` is the occurrences after the first)
	uint32_t k = 0;
	if (claimed == kNoSlot) k = 1u + atomicAdd(&H.count[h], 1u);
	if (rel >= 0xFFFFu) k = kToggleOffsets + atomicAdd(&H.count[h], 1000u); // (a stream of 64 KB and more: walked whole)
	if (k < kToggleOffsets) H.items[h].off[k] = (uint16_t)rel;
	return claimed;
}
__device__ __forceinline__ void search_list(const SearchItems &H, uint32_t q, uint32_t slot) {
	const uint32_t at = atomicAdd(&H.list_count[q], 1u);
	if (at < H.list_cap) H.list_index[(size_t)q * H.list_cap + at] = slot;
}

// every string that ends on `last_byte` and starts below `below`; returns one of the slots claimed (kNoSlot: none), any others are listed here
__device__ __forceinline__ uint32_t search_emit_all(const SearchDev &A, const SearchItems &H, uint32_t entry, uint32_t last_byte, uint32_t below,
                                                    uint32_t list) {
	uint32_t claimed = kNoSlot;
	const uint32_t f = A.out_first[entry];
	if (f & 0x40000000u) { // (SearchImage::kInline: one string)
		if (f == 0xFFFFFFFFu) return kNoSlot;
		const uint32_t len = (f >> 16) & 0x3FFFu, first = last_byte + 1 - len;
		return first < below ? search_emit(H, f & 0xFFFFu, first, len) : kNoSlot;
	}
	for (uint32_t o = f;; o++) {
		const uint32_t r = A.out_list[o];
		const uint32_t len = (r >> 16) & 0x3FFFu, first = last_byte + 1 - len;
		if (first < below) {
			const uint32_t c = search_emit(H, r & 0xFFFFu, first, len);
			if (c != kNoSlot) {
				if (claimed != kNoSlot) search_list(H, list, claimed);
				claimed = c;
			}
		}
		if (r & 0x80000000u) break;
	}
	return claimed;
}

// The search's records, a batch per wave at a time, into (string, stream) items.  New items are listed wave by wave:
// a wave collects the slots it claimed in LDS and appends them kItemStage at a time (the kToggleLists counters share
// two cache lines, which take about 90 atomics per microsecond each: an atomic per wave and round -- 940 k of them
// for 256 moves on 2.5 GB -- was 3.4 of the call's 12.5 ms).
constexpr uint32_t kItemStage = 512;
__global__ void __launch_bounds__(256) search_items_kernel(SearchDev A, StreamDev Q, SearchItems H) {
	__shared__ uint32_t s_stage[4][kItemStage];
	__shared__ uint32_t s_staged[4], s_at;
	const uint32_t record_blocks = gridDim.x;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t my_list = (blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)) % kToggleLists;
	uint32_t *stage = s_stage[threadIdx.x >> 6];
	uint32_t staged = 0; // (the same in every lane)
	auto flush = [&]() {
		if (staged == 0) return;
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); // (the lanes read what other lanes of the wave wrote)
		uint32_t at = 0;
		if (lane == 0) at = atomicAdd(&H.list_count[my_list], staged);
		at = __builtin_amdgcn_readfirstlane((int)at);
		for (uint32_t i = lane; i < staged; i += 64)
			if (at + i < H.list_cap) H.list_index[(size_t)my_list * H.list_cap + at + i] = stage[i];
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		staged = 0;
	};
	{
		// a batch per workgroup at a time, its waves side by side: a record is a dozen dependent reads and atomics, so
		// what counts is how many are under way
		uint32_t batches = Q.cursor[0];
		if (batches > Q.capacity_batches) batches = Q.capacity_batches;
		const uint2 *rec = reinterpret_cast<const uint2 *>(Q.blocks);
		for (uint32_t bt = blockIdx.x; bt < batches; bt += record_blocks) {
			uint32_t n = Q.fill[bt]; // 16-bit entries written
			if (n > kDfaBatchEntries) n = kDfaBatchEntries;
			n = uni32(n / 4);
			for (uint32_t i0 = (threadIdx.x >> 6) * 64; i0 < n; i0 += blockDim.x) {
				uint32_t claimed = kNoSlot;
				if (i0 + lane < n) {
					const uint2 r = rec[(size_t)bt * (kDfaBatchEntries / 4) + i0 + lane];
					// (r.x >> 16: found that many bytes behind a range boundary, by the lane of the range before it: the
					// strings that begin before the boundary are its own)
					const uint32_t behind = r.x >> 24, len = (r.x >> 16) & 63u;
					// (len: the bytes of the one string that ends there, where the entry's word could say so: three of four
					// occurrences lie inside a token of the vocabulary as it is, and are dropped here before any table is read)
					const uint32_t q = r.y + 1 - len;
					if (len == 0 || (((H.starts[q >> 6] >> (q & 63)) & 1ull) && search_start_within(H, q, len)))
						claimed = search_emit_all(A, H, r.x & 0xFFFFu, r.y, behind ? r.y + 1 - behind : 0xFFFFFFFFu, my_list);
				}
#ifdef DFA_EXP_STATS // (csrc/Makefile exp_dfa: records read, for the trace line of score_toggles_by_search)
				{
					const unsigned long long mr = __builtin_amdgcn_ballot_w64(i0 + lane < n);
					if (lane == 0) atomicAdd(&H.flags[2], (uint32_t)__builtin_popcountll(mr));
				}
#endif
				const unsigned long long m = __builtin_amdgcn_ballot_w64(claimed != kNoSlot);
				if (m != 0) {
					if (claimed != kNoSlot)
						stage[staged + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = claimed;
					staged += (uint32_t)__builtin_popcountll(m);
					if (staged > kItemStage - 64) flush();
				}
			}
		}
	}
	// what is left goes out workgroup by workgroup: one atomic for its four waves (at 100 MB every wave ends with a
	// few items, and 16,384 atomics on the counters' two lines were the kernel's 54 us)
	const uint32_t wv = threadIdx.x >> 6;
	if (lane == 0) s_staged[wv] = staged;
	__syncthreads();
	const uint32_t n0 = s_staged[0], n1 = s_staged[1], n2 = s_staged[2], n3 = s_staged[3];
	const uint32_t wg_list = blockIdx.x % kToggleLists;
	if (threadIdx.x == 0 && n0 + n1 + n2 + n3 != 0) s_at = atomicAdd(&H.list_count[wg_list], n0 + n1 + n2 + n3);
	__syncthreads();
	if (n0 + n1 + n2 + n3 != 0) {
		const uint32_t at = s_at + (wv > 0 ? n0 : 0u) + (wv > 1 ? n1 : 0u) + (wv > 2 ? n2 : 0u);
		for (uint32_t i = lane; i < staged; i += 64)
			if (at + i < H.list_cap) H.list_index[(size_t)wg_list * H.list_cap + at + i] = stage[i];
	}
}
__global__ void __launch_bounds__(256) stream_blocks_kernel(const uint32_t *pbeg, uint32_t nstreams, uint32_t nblocks, uint32_t *block_stream) {
	for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < nblocks; b += gridDim.x * blockDim.x) {
		const uint32_t q = b << 8;
		uint32_t l = 0, h = nstreams; // last stream with pbeg <= q (pbeg[0] = 0)
		while (h - l > 1) {
			const uint32_t mid = (l + h) >> 1;
			if (pbeg[mid] <= q) l = mid;
			else h = mid;
		}
		block_stream[b] = l;
	}
}

// ---- events -> counts ------------------------------------------------------------------------
__device__ __forceinline__ void book_event(const DfaDev &A, uint32_t ev, long long n, unsigned long long *out,
                                           uint32_t total_index) {
	if (ev == kEvNone || n == 0) return;
	if (!(ev & kEvList)) {
		atomicAdd(&out[ev], (unsigned long long)n);
		atomicAdd(&out[total_index], (unsigned long long)n);
		return;
	}
	const uint32_t o = ev & ~kEvList, len = A.pool[o];
	for (uint32_t i = 0; i < len; i++) atomicAdd(&out[A.pool[o + 1 + i]], (unsigned long long)n);
	atomicAdd(&out[total_index], (unsigned long long)(n * (long long)len));
}

// as book_event, the total left to the caller: returns the tokens the event stands for, times n
__device__ __forceinline__ long long book_event_ids(const DfaDev &A, uint32_t ev, long long n, unsigned long long *out) {
	if (ev == kEvNone || n == 0) return 0;
	if (!(ev & kEvList)) {
		atomicAdd(&out[ev], (unsigned long long)n);
		return n;
	}
	const uint32_t o = ev & ~kEvList, len = A.pool[o];
	for (uint32_t i = 0; i < len; i++) atomicAdd(&out[A.pool[o + 1 + i]], (unsigned long long)n);
	return n * (long long)len;
}
// one atomic per workgroup for the tokens its threads booked (every thread of the workgroup calls this)
__device__ __forceinline__ void add_block_total(long long mine, unsigned long long *out, uint32_t total_index) {
	__shared__ long long s_block_total;
	if (threadIdx.x == 0) s_block_total = 0;
	__syncthreads();
	for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
	if ((threadIdx.x & 63) == 0 && mine != 0) atomicAdd((unsigned long long *)&s_block_total, (unsigned long long)mine);
	__syncthreads();
	if (threadIdx.x == 0 && s_block_total != 0) atomicAdd(&out[total_index], (unsigned long long)s_block_total);
}

// ---- range entries: fix-up -----------------------------------------------------------------------
struct DfaFix {
	DfaDev A;
	const uint8_t *codes; // the prepared corpus, readable (zeros) up to nranges * range_bytes
	uint32_t range_bytes, nranges;
	uint2 *entry, *exit_a, *exit_b;
	uint32_t *exit_tag; // exit_b[g] counts when exit_tag[g] == epoch
	uint32_t *changed;  // [0] some repair moved an exit, [1] workgroups done
	uint32_t epoch;
	unsigned long long *out;
	uint32_t total_index;
	uint32_t blocks;
};

// Two range boundaries' states are the same when their words are and, under cold_base, their global bases (the
// global base a lane carries outside cold_base is whatever it last loaded: it says nothing)
__device__ __forceinline__ bool dfa_same_state(const DfaDev &A, uint2 a, uint2 b) {
	if (A.fail_code) return a.y == b.y; // (failure form: entered through the same entry -- the word follows from it)
	return a.x == b.x && ((a.x & kBaseMask) != A.cold_base || a.y == b.y);
}

// the failure form's step, until the byte is taken (G is T here: the entry the state was entered through)
__device__ void dfa_step_global_fail(const DfaFix &F, uint32_t &S, uint32_t &T, uint32_t c, long long sign) {
	const DfaDev &A = F.A;
	for (;;) {
		const uint32_t b = S & kBaseMask;
		uint32_t id = b + c, x = A.hot[id];
		bool hit = ((x >> kLabelShift) & kLabelMask) == c;
		if (!hit && (S & kPartial)) {
			id = A.gof[T] + c;
			x = A.slot[id].x;
			hit = ((x >> kLabelShift) & kLabelMask) == c;
		}
		if (hit) {
			S = x & kStateMask, T = id;
			return;
		}
		book_event(A, A.ev_miss[T], sign, F.out, F.total_index);
		if (S & kRootFail) {
			S = c * A.stride | kRootFail, T = c;
			return;
		}
		id = b + A.fail_code, x = A.hot[id];
		if (((x >> kLabelShift) & kLabelMask) != A.fail_code) {
			id = A.gof[T] + A.fail_code;
			x = A.slot[id].x;
		}
		S = x & kStateMask, T = id;
	}
}

__device__ __forceinline__ void dfa_step_global(const DfaFix &F, uint32_t &S, uint32_t &G, uint32_t c, long long sign) {
	if (F.A.fail_code) return dfa_step_global_fail(F, S, G, c, sign);
	DfaStep r = dfa_step(F.A, F.A.hot, S, G, c);
	if (r.second_look) dfa_second_look(F.A, r, S, G, c);
	if (r.hit) {
		if (r.x & kArr) book_event(F.A, F.A.ev_arr[r.id], sign, F.out, F.total_index);
		S = r.x & (kBaseMask | kPartial);
		G = r.g_next;
	} else {
		book_event(F.A, F.A.ev_miss[dfa_key(F.A, S, G)], sign, F.out, F.total_index);
		S = c * F.A.stride;
	}
}

// true when the two automata met inside the range (its exit stands); *exit_out = the true one's state at the end.
// (Two states are the same when their words are and, under cold_base, their global bases.)
__device__ bool dfa_dual_walk(const DfaFix &F, uint32_t g, uint2 want, uint2 have, uint2 *exit_out) {
	const uint32_t keep = kStateMask;
	uint32_t Sa = want.x & keep, Ga = want.y, Sb = have.x & keep, Gb = have.y;
	auto same = [&]() { return F.A.fail_code ? Ga == Gb : Sa == Sb && ((Sa & kBaseMask) != F.A.cold_base || Ga == Gb); };
	const uint8_t *p = F.codes + (size_t)g * F.range_bytes;
	for (uint32_t i = 0; i < F.range_bytes; i++) {
		if (same()) return true;
		const uint32_t c = p[i];
		dfa_step_global(F, Sa, Ga, c, 1);
		dfa_step_global(F, Sb, Gb, c, -1);
	}
	*exit_out = uint2{Sa, Ga};
	return same();
}

__device__ __forceinline__ uint2 dfa_exit_now(const DfaFix &F, uint32_t g) {
	return F.exit_tag[g] == F.epoch ? F.exit_b[g] : F.exit_a[g];
}

// ONE workgroup; every thread reaches every barrier.  Does nothing unless a repair moved an exit.
__device__ void dfa_fixup_ordered(const DfaFix &F) {
	__shared__ uint32_t s_first;
	if (__hip_atomic_load(&F.changed[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return; // uniform
	__threadfence();
	for (uint32_t base = 1; base < F.nranges; base += blockDim.x) {
		for (;;) {
			const uint32_t g = base + threadIdx.x;
			bool need = false;
			if (g < F.nranges) {
				need = !dfa_same_state(F.A, dfa_exit_now(F, g - 1), F.entry[g]);
			}
			if (threadIdx.x == 0) s_first = 0xFFFFFFFFu;
			__syncthreads();
			if (need) atomicMin(&s_first, g);
			__syncthreads();
			const uint32_t first = s_first;
			if (first == 0xFFFFFFFFu) break;
			if (g == first) {
				const uint2 want = dfa_exit_now(F, g - 1), have = F.entry[g];
				uint2 ex;
				if (!dfa_dual_walk(F, g, want, have, &ex)) {
					F.exit_b[g] = ex;
					F.exit_tag[g] = F.epoch;
				}
				F.entry[g] = want;
				__threadfence();
			}
			__syncthreads();
		}
		__syncthreads();
	}
}

__device__ void dfa_fixup_body(const DfaFix &F) {
	__shared__ uint32_t s_last, s_wrote;
	if (threadIdx.x == 0) s_wrote = 0;
	__syncthreads();
	for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < F.nranges; g += F.blocks * blockDim.x) {
		if (g == 0) continue;
		const uint2 want = F.exit_a[g - 1], have = F.entry[g];
		if (!dfa_same_state(F.A, want, have)) {
			uint2 ex;
			if (!dfa_dual_walk(F, g, want, have, &ex)) {
				F.exit_b[g] = ex;
				F.exit_tag[g] = F.epoch;
				atomicOr(&F.changed[0], 1u);
			}
			F.entry[g] = want;
			s_wrote = 1u;
		}
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		if (s_wrote) __threadfence();
		s_last = atomicAdd(&F.changed[1], 1u) == F.blocks - 1 ? 1u : 0u;
	}
	__syncthreads();
	if (!s_last) return;
	dfa_fixup_ordered(F);
}

// ---- streams -> counts: two launches per evaluation ------------------------------------------------
// dfa_hist_kernel: workgroups [0, split) bin the miss stream (keys = LDS bases of the states left),
// workgroups [split, gridDim.x) the arrival stream (keys = LDS entries arrived on), each into LDS and
// then into a slab row of 16 bits per key (as hist_slots_kernel, kernels_stream.hip, whose loop this
// is); the evaluation's fix-up of the range entries rides along in the first workgroups (with an
// unchanged vocabulary it only compares: a launch of its own costs more).
__global__ void __launch_bounds__(1024) dfa_hist_kernel(DfaDev A, StreamDev Qm, StreamDev Qa, int16_t *slab, uint32_t split, DfaFix F,
                                                        unsigned long long *out, uint32_t total_index) {
	extern __shared__ __attribute__((aligned(16))) unsigned char dfa_hist_smem[];
	uint32_t *bins = reinterpret_cast<uint32_t *>(dfa_hist_smem);
#ifdef DFA_EXP_HIST1 // (timing experiments of csrc/Makefile exp_dfa: parts left out, results wrong)
	return;
#endif
	if (F.blocks != 0 && blockIdx.x < F.blocks) dfa_fixup_body(F);
#ifdef DFA_EXP_HIST2
	return;
#endif
	const bool arrivals = blockIdx.x >= split;
	const StreamDev &Q = arrivals ? Qa : Qm;
	const uint32_t nblk = arrivals ? gridDim.x - split : split, bidx = arrivals ? blockIdx.x - split : blockIdx.x;
	const uint32_t width = A.hot_slots;
	for (uint32_t i = threadIdx.x; i < width + 64; i += blockDim.x) bins[i] = 0;
	__syncthreads();
#ifdef DFA_EXP_HIST3
	return;
#endif
	uint32_t batches = Q.cursor[0];
	if (batches > Q.capacity_batches) batches = Q.capacity_batches;
#ifdef DFA_EXP_HIST4
	batches = 0;
#endif
	constexpr uint32_t vecs_per_block = 512 * 2 / 16; // 64: one 1 KiB row per block of a batch
	constexpr int kRowsInFlight = 8;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t n_waves = nblk * (blockDim.x / 64);
	const uint32_t wave_id = uni32(bidx * (blockDim.x / 64) + (threadIdx.x >> 6));
	const uint4 *vec = reinterpret_cast<const uint4 *>(Q.blocks);
	const uint32_t spare = width + lane; // where entries past a batch's fill go: a word of the lane's own, never read
	auto blank = [&](uint4 v, uint32_t first, uint32_t live) { // a batch's last, partly written vector: what lies past `live` to 0xFFFF
		if (first + 8 <= live) return v;
		uint32_t w[4] = {v.x, v.y, v.z, v.w};
		for (uint32_t q = 0; q < 8; q++)
			if (first + q >= live) w[q >> 1] |= 0xFFFFu << (16 * (q & 1));
		return uint4{w[0], w[1], w[2], w[3]};
	};
	auto count8 = [&](const uint4 &v) {
		const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const uint32_t e0 = wds[k] & 0xFFFFu, e1 = wds[k] >> 16;
			atomicAdd(&bins[e0 < spare ? e0 : spare], 1u);
			atomicAdd(&bins[e1 < spare ? e1 : spare], 1u);
		}
	};
	if (!arrivals) {
		// rows first, first + step, ... of a batch, eight in flight
		auto some_rows = [&](uint32_t bt, uint32_t first_row, uint32_t step) {
			const uint4 *rows = vec + (uint64_t)bt * kStreamBatch * vecs_per_block + lane;
			// (the first rows are asked for together with the fill word that says how many of them count: one round
			// trip instead of two; a batch's 32 rows are all readable)
			uint4 v[kRowsInFlight];
#pragma unroll
			for (int u = 0; u < kRowsInFlight; u++) {
				const uint32_t r = first_row + u * step;
				v[u] = r < kStreamBatch ? rows[(size_t)r * 64] : uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
			}
			uint32_t live = Q.fill[bt]; // entries written to the batch (what follows them was never written, or is not counted)
			if (live > kDfaBatchEntries) live = kDfaBatchEntries;
			live = uni32(live);
			const uint32_t live_rows = (live + 511) / 512;
			for (uint32_t r0 = first_row; r0 < live_rows; r0 += kRowsInFlight * step) {
#pragma unroll
				for (int u = 0; u < kRowsInFlight; u++) {
					const uint32_t r = r0 + u * step;
					if (r0 != first_row) {
						v[u] = uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
						if (r < live_rows) v[u] = rows[(size_t)r * 64];
					}
					if (r < live_rows) v[u] = blank(v[u], r * 512 + lane * 8, live);
				}
#pragma unroll
				for (int u = 0; u < kRowsInFlight; u++)
					if (r0 + u * step < live_rows) count8(v[u]);
			}
		};
		// The scan's waves own a batch each and 248 workgroups' waves are fewer than that: a wave takes the batch
		// of its number, and the batches left over are split by rows among all waves (handing batches out by an
		// atomic counter was tried: 4,000 waves on one address took 45 us).
		if (wave_id < batches) some_rows(wave_id, 0, 1);
		const uint32_t tail = batches > n_waves ? batches - n_waves : 0u;
		if (tail != 0 && tail <= n_waves) {
			const uint32_t per = n_waves / tail, tb = wave_id / per;
			if (tb < tail) some_rows(n_waves + tb, wave_id % per, per);
		} else {
			for (uint32_t bt = n_waves + wave_id; tail != 0 && bt < batches; bt += n_waves) some_rows(bt, 0, 1);
		}
	} else {
		// the arrival stream: a batch per wave of the scan, a few dozen entries in most -- eight batches in flight, row by row
		for (uint32_t b0 = wave_id; b0 < batches; b0 += kRowsInFlight * n_waves) {
			uint32_t live[kRowsInFlight], most = 0;
#pragma unroll
			for (int u = 0; u < kRowsInFlight; u++) {
				const uint32_t bt = b0 + u * n_waves;
				live[u] = bt < batches ? Q.fill[bt] : 0u;
			}
#pragma unroll
			for (int u = 0; u < kRowsInFlight; u++) {
				if (live[u] > kDfaBatchEntries) live[u] = kDfaBatchEntries;
				live[u] = uni32(live[u]);
				most = live[u] > most ? live[u] : most;
			}
			for (uint32_t r = 0; r * 512 < most; r++) {
				uint4 v[kRowsInFlight];
#pragma unroll
				for (int u = 0; u < kRowsInFlight; u++) {
					v[u] = uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
					if (r * 512 < live[u])
						v[u] = blank(vec[((uint64_t)(b0 + u * n_waves) * kStreamBatch + r) * vecs_per_block + lane], r * 512 + lane * 8, live[u]);
				}
#pragma unroll
				for (int u = 0; u < kRowsInFlight; u++)
					if (r * 512 < live[u]) count8(v[u]);
			}
		}
	}
	__syncthreads();
	// the slab row: 16 bits per key; a bin that does not fit goes to the output directly, as the reduction would have booked it
	int16_t *row = slab + (size_t)blockIdx.x * ((width + 1) & ~1u);
	const uint32_t *ev = arrivals ? A.ev_arr : A.ev_miss;
	for (uint32_t i = threadIdx.x; i < width; i += blockDim.x) {
		const uint32_t n = bins[i];
		const bool fits = n <= 32767u;
		row[i] = fits ? (int16_t)n : (int16_t)0;
		if (!fits) book_event(A, ev[i], (long long)n, out, total_index);
	}
}

// dfa_reduce_kernel: the slab rows' column sums through the event tables into the output (workgroups
// [0, colgroups): 128 keys each, rows [0, split) miss events, [split, rows) arrivals); the counters of the
// keys and entries of the global array, cleared for the next evaluation (the workgroups after them);
// and the streams' cursors and the fix-up's words back to their start.
constexpr uint32_t kDfaReduceGroups = 16;
__global__ void __launch_bounds__(1024) dfa_reduce_kernel(DfaDev A, const int16_t *slab, uint32_t split, uint32_t rows, uint32_t colgroups,
                                                          uint32_t *cold_miss, uint32_t *cold_arr, StreamDev Qm, StreamDev Qa,
                                                          uint32_t first_batches, uint32_t *changed, unsigned long long *out,
                                                          uint32_t total_index, uint32_t defer) {
	__shared__ int part[kDfaReduceGroups][64][4];
#ifdef DFA_EXP_RED1
	if (blockIdx.x != 0 || threadIdx.x != 0) return;
#endif
	if (blockIdx.x == 0 && threadIdx.x == 0) { // (the histogram has read the cursors, the fix-up is done with its words)
		if (Qa.cursor[1]) Qm.cursor[1] = 1u; // one overflow word for the host
		Qm.cursor[0] = first_batches, Qa.cursor[0] = first_batches, Qa.cursor[1] = 0u;
		changed[0] = changed[1] = 0u;
	}
	if (blockIdx.x >= colgroups) {
		const uint32_t n = A.nslots - A.hot_slots;
		long long mine = 0;
		for (uint32_t i = (blockIdx.x - colgroups) * blockDim.x + threadIdx.x; i < n; i += (gridDim.x - colgroups) * blockDim.x) {
			const uint32_t m = cold_miss[i], a = cold_arr[i];
			if (m) {
				mine += book_event_ids(A, A.ev_miss[A.hot_slots + i], (long long)m, out);
				cold_miss[i] = 0;
			}
			if (a) {
				mine += book_event_ids(A, A.ev_arr[A.hot_slots + i], (long long)a, out);
				cold_arr[i] = 0;
			}
		}
		add_block_total(mine, out, total_index);
		return;
	}
	const uint32_t lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
	const uint32_t k = blockIdx.x * 128 + lane * 2;
	const uint32_t width = A.hot_slots;
	const size_t stride = ((width + 1) & ~1u) / 2; // in 32-bit words
	const uint32_t *slab2 = reinterpret_cast<const uint32_t *>(slab);
	int acc[4] = {0, 0, 0, 0}; // miss lo / hi, arrival lo / hi (rows x 32,767 fits easily)
	if (k < width) {
		for (uint32_t r = grp; r < rows; r += 8 * kDfaReduceGroups) { // (eight rows in flight: the loop waits on memory, not on adds)
			uint32_t a[8];
#pragma unroll
			for (int u = 0; u < 8; u++) {
				const uint32_t ru = r + u * kDfaReduceGroups;
				a[u] = ru < rows ? slab2[(size_t)ru * stride + k / 2] : 0u;
			}
#pragma unroll
			for (int u = 0; u < 8; u++) {
				const int o = r + u * kDfaReduceGroups < split ? 0 : 2;
				acc[o] += (int)(a[u] & 0xFFFFu);
				acc[o + 1] += (int)(a[u] >> 16);
			}
		}
	}
#pragma unroll
	for (int q = 0; q < 4; q++) part[grp][lane][q] = acc[q];
	__syncthreads();
	// one (key, kind) per thread of the first four waves: a booking is three dependent reads (event, list length,
	// ids) before its atomics, and four of them one after the other per lane took 10 of the kernel's 17 us
	long long mine = 0;
	if (grp < 4) {
		const uint32_t q = grp, key = k + (q & 1u);
		int sum = 0;
		for (uint32_t g = 0; g < kDfaReduceGroups; g++) sum += part[g][lane][q];
		if (sum != 0 && key < width) {
			// (defer: the failure form's keys name the same few thousand ids tens of thousands of times -- their sums go
			// behind the global array's counters, cold_miss / cold_arr[nslots - width + key], and dfa_reduce_cold_kernel
			// books all of them through LDS bins over the ids)
			if (defer) (q < 2 ? cold_miss : cold_arr)[A.nslots - width + key] = (uint32_t)sum;
			else mine = book_event_ids(A, (q < 2 ? A.ev_miss : A.ev_arr)[key], sum, out);
		}
	}
	add_block_total(mine, out, total_index);
}

__global__ void __launch_bounds__(256) dfa_fixup_kernel(DfaFix F) { dfa_fixup_body(F); }

// kernels.h: launch_dfa_warm_entries.  One thread per range; the automaton from global memory, as the fix-up walks it,
// with nothing booked (sign 0).
__global__ void __launch_bounds__(256) dfa_warm_entries_kernel(DfaFix F, uint32_t start_word, uint32_t lookback) {
	for (uint32_t g = 1 + blockIdx.x * blockDim.x + threadIdx.x; g < F.nranges; g += gridDim.x * blockDim.x) {
		const uint64_t at = (uint64_t)g * F.range_bytes;
		const uint64_t from = at > lookback ? at - lookback : 0;
		uint32_t S = start_word & kStateMask, G = 0;
		const uint8_t *p = F.codes + from;
		for (uint32_t i = 0; i < (uint32_t)(at - from); i++) dfa_step_global(F, S, G, p[i], 0);
		F.entry[g] = uint2{S & kStateMask, G};
	}
}

// The counters of the global array's keys through the event tables, for the failure form: there a few percent of
// an evaluation's events are counted under such keys (a state entered through an entry LDS does not hold), by a
// hundred thousand counters whose events name the same few thousand ids -- booked one by one (dfa_reduce_kernel's
// last workgroups) they queue on those ids' words, 0.34 ms per evaluation.  Here a workgroup adds its counters'
// events up in LDS bins over the ids first and books every id once.
// Vocabularies of more ids than LDS has bins for take one launch per range of ids [id_lo, id_lo + n_ids): every launch
// reads all the counters, the last one (`clear`) zeroes them.
__global__ void __launch_bounds__(1024) dfa_reduce_cold_kernel(DfaDev A, uint32_t *cold_miss, uint32_t *cold_arr, uint32_t id_lo, uint32_t n_ids,
                                                               uint32_t clear, unsigned long long *out, uint32_t total_index) {
	extern __shared__ __attribute__((aligned(16))) uint32_t cold_bins[];
	for (uint32_t i = threadIdx.x; i < n_ids; i += blockDim.x) cold_bins[i] = 0;
	__syncthreads();
	auto add1 = [&](uint32_t id, uint32_t n) {
		if (id - id_lo < n_ids) atomicAdd(&cold_bins[id - id_lo], n); // (unsigned: ids below id_lo wrap beyond n_ids)
	};
	auto add = [&](uint32_t ev, uint32_t n) {
		if (ev == kEvNone) return;
		if (!(ev & kEvList)) {
			add1(ev, n);
			return;
		}
		const uint32_t o = ev & ~kEvList, len = A.pool[o];
		for (uint32_t i = 0; i < len; i++) add1(A.pool[o + 1 + i], n);
	};
	// counters [0, nslots - hot_slots): the global array's keys hot_slots + i; behind them the sums of the LDS keys
	// that dfa_reduce_kernel left there
	const uint32_t n = A.nslots, n_cold = A.nslots - A.hot_slots;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint32_t m = cold_miss[i], a = cold_arr[i];
		const uint32_t key = i < n_cold ? A.hot_slots + i : i - n_cold;
		if (m) {
			add(A.ev_miss[key], m);
			if (clear) cold_miss[i] = 0;
		}
		if (a) {
			add(A.ev_arr[key], a);
			if (clear) cold_arr[i] = 0;
		}
	}
	__syncthreads();
	long long mine = 0;
	for (uint32_t i = threadIdx.x; i < n_ids; i += blockDim.x) {
		const uint32_t v = cold_bins[i];
		if (v) {
			atomicAdd(&out[id_lo + i], (unsigned long long)v);
			mine += v;
		}
	}
	add_block_total(mine, out, total_index);
}

} // namespace

// ---- host side -----------------------------------------------------------------------------------

void launch_dfa_transpose(const uint8_t *prep, uint64_t prep_total, uint32_t range_bytes, uint32_t chains, uint64_t nwb,
                          void *tcodes, hipStream_t stream) {
	const uint64_t n = nwb * (range_bytes / 8) * chains * 64;
	if (n == 0) return;
	const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, 1u << 16);
	hipLaunchKernelGGL(dfa_transpose_kernel, dim3(blocks), dim3(256), 0, stream, prep, prep_total, range_bytes, chains, nwb,
	                   (uint2 *)tcodes);
}

void launch_dfa_reset_states(const DfaWork &W, uint32_t start_word, hipStream_t stream) {
	if (W.nranges == 0) return;
	hipLaunchKernelGGL(dfa_reset_states_kernel, dim3(std::min<uint32_t>((W.nranges + 255) / 256, 1024)), dim3(256), 0, stream,
	                   W.entry, W.nranges, start_word);
}

void launch_dfa_warm_entries(const DfaDev &A, const DfaWork &W, const uint8_t *codes, uint32_t start_word, uint32_t lookback,
                             hipStream_t stream) {
	if (W.nranges < 2 || lookback == 0) return;
	DfaFix F{};
	F.A = A, F.codes = codes, F.range_bytes = W.range_bytes, F.nranges = W.nranges, F.entry = W.entry;
	hipLaunchKernelGGL(dfa_warm_entries_kernel, dim3(std::min<uint32_t>((W.nranges + 255) / 256, 4096)), dim3(256), 0, stream, F, start_word,
	                   lookback);
}

void launch_dfa_prepare(unsigned long long *out, uint32_t V, uint64_t D, uint64_t total_bytes, const StreamDev &Qm,
                        const StreamDev &Qa, uint32_t first_batches, uint32_t *changed, hipStream_t stream) {
	const uint32_t blocks = out ? std::min<uint32_t>((V + 5 + 255) / 256, 256) : 1;
	hipLaunchKernelGGL(dfa_prepare_kernel, dim3(blocks), dim3(256), 0, stream, out, V, (unsigned long long)D,
	                   (unsigned long long)total_bytes, Qm.cursor, Qa.cursor, first_batches, changed);
}

void launch_dfa_scan(const DfaDev &A, const DfaWork &W, const StreamDev &Qm, const StreamDev &Qa, const DfaScanPlan &plan,
                     hipStream_t stream, const OutInit &init, hipEvent_t ev_start, hipEvent_t ev_stop) {
	auto go = [&](auto kernel) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
		                          (int)plan.lds_bytes);
		if (ev_start && ev_stop)
			hipExtLaunchKernelGGL(kernel, dim3(plan.blocks), dim3(64 * plan.waves_per_block), plan.lds_bytes, stream, ev_start, ev_stop,
			                      0u, A, W, Qm, Qa, init);
		else hipLaunchKernelGGL(kernel, dim3(plan.blocks), dim3(64 * plan.waves_per_block), plan.lds_bytes, stream, A, W, Qm, Qa, init);
	};
	static const bool use_asm = [] {
		const char *e = getenv("TOKSCORE_ASM");
		return !(e && atoi(e) == 0);
	}();
	if (A.fail_code) { // the failure form: lanes with windows of their own (W.cold_hit: null once the layout has its counts)
		// (dbg: timing experiments, results wrong -- 1 no counters for the global array's keys, 2 no event log, 4 no global
		// lookups; tools/dbg/fail_dbg.sh.  Measured at 100 MB on the annealed vocabulary: 0.65 ms as it is, 0.53 / 0.62 / 0.31
		// without each, 0.21 without all three = the loop itself)
		const uint32_t dbg = getenv("TOKSCORE_FSCAN_DBG") ? (uint32_t)atoi(getenv("TOKSCORE_FSCAN_DBG")) : 0u;
		auto go = [&](auto kernel) {
			(void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes);
			if (ev_start && ev_stop)
				hipExtLaunchKernelGGL(kernel, dim3(plan.blocks), dim3(64 * plan.waves_per_block), plan.lds_bytes, stream, ev_start, ev_stop,
				                      0u, A, W, Qm, Qa, init, dbg);
			else hipLaunchKernelGGL(kernel, dim3(plan.blocks), dim3(64 * plan.waves_per_block), plan.lds_bytes, stream, A, W, Qm, Qa, init, dbg);
		};
		if (W.cold_hit) {
			switch (plan.chains) {
			case 1: go(&dfa_fscan_kernel<1, true>); break;
			case 2: go(&dfa_fscan_kernel<2, true>); break;
			default: go(&dfa_fscan_kernel<4, true>); break;
			}
		} else {
			switch (plan.chains) {
			case 1: go(&dfa_fscan_kernel<1, false>); break;
			case 2: go(&dfa_fscan_kernel<2, false>); break;
			default: go(&dfa_fscan_kernel<4, false>); break;
			}
		}
	} else if (use_asm) {
		switch (plan.chains) {
		case 1: go(&dfa_scan_asm_kernel<1>); break;
		case 2: go(&dfa_scan_asm_kernel<2>); break;
		default: go(&dfa_scan_asm_kernel<4>); break;
		}
	} else {
		switch (plan.chains) {
		case 1: go(&dfa_scan_kernel<1>); break;
		case 2: go(&dfa_scan_kernel<2>); break;
		default: go(&dfa_scan_kernel<4>); break;
		}
	}
}

void launch_dfa_count(const DfaDev &A, const DfaWork &W, const StreamDev &Qm, const StreamDev &Qa, const uint8_t *codes,
                      uint32_t *changed, uint32_t epoch, unsigned long long *out, uint32_t total_index, uint32_t *slab,
                      uint32_t miss_blocks, uint32_t arrival_blocks, uint32_t first_batches, hipStream_t stream) {
	DfaFix F{};
	if (W.nranges >= 2) {
		const uint32_t blocks = std::min<uint32_t>(std::min<uint32_t>((W.nranges + 1023) / 1024, 64), miss_blocks);
		F = DfaFix{A, codes, W.range_bytes, W.nranges, W.entry, W.exit_a, W.exit_b, W.exit_tag, changed, epoch, out, total_index, blocks};
	}
	const uint32_t lds = (A.hot_slots + 64) * 4; // + a spare word per lane
	(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&dfa_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	int16_t *rows = reinterpret_cast<int16_t *>(slab);
	hipLaunchKernelGGL(dfa_hist_kernel, dim3(miss_blocks + arrival_blocks), dim3(1024), lds, stream, A, Qm, Qa, rows, miss_blocks, F, out,
	                   total_index);
	const uint32_t colgroups = (A.hot_slots + 127) / 128;
	const uint32_t ncold = A.nslots - A.hot_slots;
	// (the failure form counts a share of its events under keys of the global array: a launch of its own adds them up in LDS first)
	const bool cold_apart = A.fail_code != 0 && ncold != 0;
	const uint32_t coldblocks = cold_apart ? 0u : std::min<uint32_t>((ncold + 1023) / 1024, 256);
	hipLaunchKernelGGL(dfa_reduce_kernel, dim3(colgroups + coldblocks), dim3(64 * kDfaReduceGroups), 0, stream, A, rows, miss_blocks,
	                   miss_blocks + arrival_blocks, colgroups, W.cold_miss, W.cold_arr, Qm, Qa, first_batches, changed, out, total_index,
	                   cold_apart ? 1u : 0u);
	if (cold_apart) {
		constexpr uint32_t kIdsPerLaunch = 38400; // bins of a launch: 150 KB of LDS
		const uint32_t per = std::min<uint32_t>(total_index, kIdsPerLaunch);
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&dfa_reduce_cold_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(per * 4));
		for (uint32_t lo = 0; lo < total_index; lo += per) {
			const uint32_t n = std::min<uint32_t>(per, total_index - lo);
			hipLaunchKernelGGL(dfa_reduce_cold_kernel, dim3(std::min<uint32_t>((A.nslots + 4095) / 4096, 64)), dim3(1024), per * 4, stream, A,
			                   W.cold_miss, W.cold_arr, lo, n, lo + n >= total_index ? 1u : 0u, out, total_index);
		}
	}
}

void launch_stream_blocks(const uint32_t *pbeg, uint32_t nstreams, uint32_t prep_total, uint32_t *block_stream, hipStream_t stream) {
	const uint32_t nblocks = prep_total / 256 + 1;
	hipLaunchKernelGGL(stream_blocks_kernel, dim3(std::min<uint32_t>((nblocks + 255) / 256, 4096)), dim3(256), 0, stream, pbeg, nstreams, nblocks,
	                   block_stream);
}

void launch_dfa_search(const SearchDev &A, const DfaWork &W, const StreamDev &Q, const uint8_t *codes, const DfaScanPlan &plan, hipStream_t stream) {
	const uint32_t lds = ((A.hot_slots + 3) & ~3u) * 4;
	auto go = [&](auto kernel) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		hipLaunchKernelGGL(kernel, dim3(plan.blocks), dim3(64 * plan.waves_per_block), lds, stream, A, W, Q, codes);
	};
	switch (plan.chains) {
	case 1: go(&dfa_search_kernel<1>); break;
	case 2: go(&dfa_search_kernel<2>); break;
	default: go(&dfa_search_kernel<4>); break;
	}
}

void launch_search_items(const SearchDev &A, const StreamDev &Q, const SearchItems &H, uint32_t num_cus, hipStream_t stream) {
	// (a record is a chain of a dozen dependent reads: many waves -- but every workgroup ends with an atomic on the lists'
	// counters, which is what a small call waits for: 5.2 against 4.7 ms at 2.5 GB with 8 per CU, 0.228 against 0.249 at 100 MB)
	const uint32_t record_blocks = std::max<uint32_t>(1, num_cus * (H.prep_total >= (512u << 20) ? 16 : 8));
	hipLaunchKernelGGL(search_items_kernel, dim3(record_blocks), dim3(256), 0, stream, A, Q, H);

}

bool plan_dfa_scan(uint32_t hot_slots, uint64_t prep_total, uint32_t num_cus, uint32_t chains_hint, DfaScanPlan *plan) {
	plan->lds_slots = hot_slots;
	plan->lds_bytes = ((plan->lds_slots + 3) & ~3u) * 4;
	plan->blocks = num_cus;
	plan->waves_per_block = 16;
	plan->chains = chains_hint == 1 || chains_hint == 4 ? chains_hint : 2; // (measured at 100 MB: 0.0613 / 0.0611 / 0.0666 ms for 1 / 2 / 4)
	const uint64_t total_waves = (uint64_t)plan->blocks * plan->waves_per_block;
	const uint64_t per_block = 64ull * plan->chains;
	uint64_t rb = (prep_total + total_waves * per_block - 1) / (total_waves * per_block);
	rb = std::max<uint64_t>(64, (rb + 7) & ~7ull);
	if (rb > 0xFFFFFFF8ull) return false;
	plan->range_bytes = (uint32_t)rb;
	plan->nwb = (uint32_t)((prep_total + rb * per_block - 1) / (rb * per_block));
	if (plan->nwb == 0) plan->nwb = 1;
	return (uint64_t)plan->nwb * per_block < 0xFFFFFFFFull;
}

uint32_t dfa_lds_slots_max(uint32_t cap) {
	// slots: 4 bytes each in the scan's LDS and a 4-byte bin each, plus 64 spare words and a few static ones, in
	// dfa_hist_kernel's -- all of a CU's 160 KiB but a few hundred bytes
	constexpr uint32_t kMaxLds = 40832;
	return cap ? std::min<uint32_t>(kMaxLds, std::max<uint32_t>(cap, 512)) : kMaxLds;
}

uint64_t dfa_stream_capacity_batches(const DfaScanPlan &plan) {
	const uint64_t bytes = (uint64_t)plan.nwb * 64 * plan.chains * plan.range_bytes;
	const uint64_t waves = (uint64_t)plan.blocks * plan.waves_per_block;
	return bytes / (kDfaBatchEntries - 8 * plan.chains * 64) + 2 * waves + 16;
}

} // namespace tokscore
