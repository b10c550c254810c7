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

namespace tokscore {

namespace {

constexpr uint32_t kDfaBatchEntries = kStreamBatch * 512; // u16 entries of one stream batch (32 KiB)
constexpr uint32_t kLabelShift = 24, kArr = 0x800000u, kBaseMask = 0x7FFFFFu;
constexpr uint32_t kEvNone = 0xFFFFFFFFu, kEvList = 0x80000000u;

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
		entry[g] = uint2{start_word & 0xFFFFFFu, 0u};
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
	{
		const u32x4 *src = reinterpret_cast<const u32x4 *>(A.slot);
		LDS_AS u32x4 *dst = (LDS_AS u32x4 *)img;
		for (uint32_t i = threadIdx.x; i < (A.lds_slots + 3) / 4; i += blockDim.x) dst[i] = src[i]; // (the image is padded to 16 bytes)
	}
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
	const uint32_t lds_slots = A.lds_slots, stride = A.stride;
	constexpr uint32_t kHigh = kDfaBatchEntries - 8 * C * 64;

	for (uint32_t wb = wave_id; wb < W.nwb; wb += total_waves) {
		uint32_t S[C], U[C];
#pragma unroll
		for (int j = 0; j < C; j++) {
			const uint2 e = W.entry[(wb * C + j) * 64 + lane];
			S[j] = e.x, U[j] = e.y;
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
					const uint32_t t = (S[j] & 0xFFFFFFu) + c; // (a virtual slot's state carries kArr: beyond LDS, masked below)
					uint32_t x;
					if (t < lds_slots) x = img[t];
					else x = A.slot[t & kBaseMask];
					const bool hit = (x >> kLabelShift) == c;
					const bool arr = hit && (x & kArr) != 0;
					const uint32_t ts = t & kBaseMask;
					// the state left by a miss: its slot to the miss stream (or, beyond the LDS prefix, its counter)
					lm.put(!hit && U[j] < lds_slots, U[j]);
					if (!hit && U[j] >= lds_slots) atomicAdd(&W.cold_miss[U[j] - lds_slots], 1u);
					if (__builtin_amdgcn_ballot_w64(arr) != 0) {
						la.put(arr && ts < lds_slots, ts);
						if (arr && ts >= lds_slots) atomicAdd(&W.cold_arr[ts - lds_slots], 1u);
					}
					S[j] = hit ? x : c * stride;
					U[j] = hit ? ts : c;
				}
			}
#pragma unroll
			for (int j = 0; j < C; j++) cur[j] = nxt[j];
			if (lm.cur > kHigh) lm.roll(Qm, lane);
			if (la.cur > kHigh) la.roll(Qa, lane);
		}
#pragma unroll
		for (int j = 0; j < C; j++) W.exit_a[(wb * C + j) * 64 + lane] = uint2{S[j] & 0xFFFFFFu, U[j]};
	}
	lm.close(Qm, lane);
	la.close(Qa, lane);
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

// the counters of the slots beyond the LDS prefix; cleared here for the next evaluation
__global__ void __launch_bounds__(256) dfa_cold_reduce_kernel(DfaDev A, uint32_t *cold_miss, uint32_t *cold_arr,
                                                              unsigned long long *out, uint32_t total_index) {
	const uint32_t n = A.nslots - A.lds_slots;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint32_t m = cold_miss[i], a = cold_arr[i];
		if (m) {
			book_event(A, A.ev_miss[A.lds_slots + i], (long long)m, out, total_index);
			cold_miss[i] = 0;
		}
		if (a) {
			book_event(A, A.ev_arr[A.lds_slots + i], (long long)a, out, total_index);
			cold_arr[i] = 0;
		}
	}
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

__device__ __forceinline__ void dfa_step_global(const DfaFix &F, uint32_t &S, uint32_t &U, uint32_t c, long long sign) {
	const uint32_t t = (S & kBaseMask) + c;
	const uint32_t x = F.A.slot[t];
	if ((x >> kLabelShift) == c) {
		if (x & kArr) book_event(F.A, F.A.ev_arr[t], sign, F.out, F.total_index);
		S = x & 0xFFFFFFu;
		U = t;
	} else {
		book_event(F.A, F.A.ev_miss[U], sign, F.out, F.total_index);
		S = c * F.A.stride;
		U = c;
	}
}

// true when the two automata met inside the range (its exit stands); *exit_out = the true one's state at the end
__device__ bool dfa_dual_walk(const DfaFix &F, uint32_t g, uint2 want, uint2 have, uint2 *exit_out) {
	uint32_t Sa = want.x & 0xFFFFFFu, Ua = want.y, Sb = have.x & 0xFFFFFFu, Ub = have.y;
	const uint8_t *p = F.codes + (size_t)g * F.range_bytes;
	for (uint32_t i = 0; i < F.range_bytes; i++) {
		if (Sa == Sb && Ua == Ub) return true;
		const uint32_t c = p[i];
		dfa_step_global(F, Sa, Ua, c, 1);
		dfa_step_global(F, Sb, Ub, c, -1);
	}
	*exit_out = uint2{Sa, Ua};
	return Sa == Sb && Ua == Ub;
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
				const uint2 w = dfa_exit_now(F, g - 1), h = F.entry[g];
				need = w.x != h.x || w.y != h.y;
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

__global__ void __launch_bounds__(256) dfa_fixup_kernel(DfaFix F) {
	__shared__ uint32_t s_last, s_wrote;
	if (threadIdx.x == 0) s_wrote = 0;
	__syncthreads();
	for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < F.nranges; g += F.blocks * blockDim.x) {
		if (g == 0) continue;
		const uint2 want = F.exit_a[g - 1], have = F.entry[g];
		if (want.x != have.x || want.y != have.y) {
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
	switch (plan.chains) {
	case 1: go(&dfa_scan_kernel<1>); break;
	case 2: go(&dfa_scan_kernel<2>); break;
	default: go(&dfa_scan_kernel<4>); break;
	}
}

void launch_dfa_fixup(const DfaDev &A, const DfaWork &W, const uint8_t *codes, uint32_t *changed, uint32_t epoch,
                      unsigned long long *out, uint32_t total_index, hipStream_t stream) {
	if (W.nranges < 2) return;
	const uint32_t blocks = std::min<uint32_t>((W.nranges + 255) / 256, 64);
	DfaFix F{A, codes, W.range_bytes, W.nranges, W.entry, W.exit_a, W.exit_b, W.exit_tag, changed, epoch, out, total_index, blocks};
	hipLaunchKernelGGL(dfa_fixup_kernel, dim3(blocks), dim3(256), 0, stream, F);
}

void launch_dfa_cold_reduce(const DfaDev &A, const DfaWork &W, unsigned long long *out, uint32_t total_index, hipStream_t stream) {
	const uint32_t n = A.nslots - A.lds_slots;
	if (n == 0) return;
	hipLaunchKernelGGL(dfa_cold_reduce_kernel, dim3(std::min<uint32_t>((n + 255) / 256, 2048)), dim3(256), 0, stream, A, W.cold_miss,
	                   W.cold_arr, out, total_index);
}

bool plan_dfa_scan(uint32_t nslots, uint64_t prep_total, uint32_t num_cus, uint32_t chains_hint, uint32_t lds_cap, DfaScanPlan *plan) {
	plan->lds_slots = std::min<uint32_t>(nslots, dfa_lds_slots_max(lds_cap));
	plan->lds_bytes = ((plan->lds_slots + 3) & ~3u) * 4;
	plan->blocks = num_cus;
	plan->waves_per_block = 16;
	plan->chains = chains_hint == 1 || chains_hint == 2 ? chains_hint : 4;
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
	constexpr uint32_t kMaxLds = 40000; // slots: what one histogram pass has bins for (kernels_stream.hip: histogram_slots)
	return cap ? std::min<uint32_t>(kMaxLds, std::max<uint32_t>(cap, 512)) : kMaxLds;
}

uint64_t dfa_stream_capacity_batches(const DfaScanPlan &plan) {
	const uint64_t bytes = (uint64_t)plan.nwb * 64 * plan.chains * plan.range_bytes;
	const uint64_t waves = (uint64_t)plan.blocks * plan.waves_per_block;
	return bytes / (kDfaBatchEntries - 8 * plan.chains * 64) + 2 * waves + 16;
}

} // namespace tokscore
