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
constexpr uint32_t kArr = 0x800000u, kBaseMask = 0x7FFFFFu;
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
		entry[g] = uint2{start_word & kBaseMask, 0u};
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
		const u32x4 *src = reinterpret_cast<const u32x4 *>(A.hot);
		LDS_AS u32x4 *dst = (LDS_AS u32x4 *)img;
		for (uint32_t i = threadIdx.x; i < (A.hot_slots + 3) / 4; i += blockDim.x) dst[i] = src[i]; // (the array is padded to 16 bytes)
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
	const uint32_t hot_slots = A.hot_slots, inc_bit = A.inc_bit, lmask = (1u << A.label_bits) - 1, stride = A.stride;
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
					const uint32_t b = S[j] & kBaseMask, t = b + c;
					uint32_t x, id = t;
					if (t < hot_slots) x = img[t];
					else x = A.slot[t];
					bool hit = ((x >> 24) & lmask) == c;
					// a partial row (dfa.h): a miss proves nothing, the row's complete copy in the global array decides
					const bool again = !hit && (S[j] & inc_bit) != 0;
					if (__builtin_amdgcn_ballot_w64(again) != 0) {
						if (again) {
							id = A.crow[b] + c;
							x = A.slot[id];
							hit = ((x >> 24) & lmask) == c;
						}
					}
					const bool arr = hit && (x & kArr) != 0;
					// the state left by a miss: its entry's index to the miss stream (or, beyond the LDS array, its counter)
					lm.put(!hit && U[j] < hot_slots, U[j]);
					if (!hit && U[j] >= hot_slots) atomicAdd(&W.cold_miss[U[j] - hot_slots], 1u);
					if (__builtin_amdgcn_ballot_w64(arr) != 0) {
						la.put(arr && id < hot_slots, id);
						if (arr && id >= hot_slots) atomicAdd(&W.cold_arr[id - hot_slots], 1u);
					}
					S[j] = hit ? x : c * stride;
					U[j] = hit ? id : c;
				}
			}
#pragma unroll
			for (int j = 0; j < C; j++) cur[j] = nxt[j];
			if (lm.cur > kHigh) lm.roll(Qm, lane);
			if (la.cur > kHigh) la.roll(Qa, lane);
		}
#pragma unroll
		for (int j = 0; j < C; j++) W.exit_a[(wb * C + j) * 64 + lane] = uint2{S[j] & (kBaseMask | inc_bit), U[j]};
	}
	lm.close(Qm, lane);
	la.close(Qa, lane);
}

// ---- the scan's loop, hand-written for gfx950 -------------------------------------------------
// One step of one chain = DFA_ISSUE (where the next slot is: 3 vector instructions and the LDS read)
// and, after the other chains' steps have covered the read's latency, DFA_RESOLVE (what it says: 11
// vector, 1 store and 14 scalar instructions on the common path).  The byte never becomes a
// register of its own: the four instructions that need it select it from the window dword (SDWA).
// exec is all ones throughout (every lane has a range; ranges are equal).  Per chain: window
// W0 W1, next piece N0 N1, T slot index, X slot word, L label / LDS address, M the state after a
// miss.  v20 v21 scratch, v32 stride, v33 0x7f, v34 1; s[38:39] all lanes, s[40:55] scratch:
//   s[40:41] lanes whose slot lies beyond the LDS array, s[42:43] lanes the global array decides,
//   s[44:45] miss lanes, s[52:53] lanes whose new event index lies beyond the LDS array.
// Rare paths, each behind one scalar branch: the global array (lane by lane with scalar loads -- a
// row beyond LDS: one; a partial row: crow[base], then the slot -- because a vector load's wait would
// also wait for every store before it), events with an index beyond the LDS array (a counter each
// instead of the 16-bit stream), arrivals on virtual entries.
#define DFA_RES_HEAD \
	".macro DFA_RESOLVE n, w, sel, s, u, t, x, l, m, uc\n"                                              \
	"  s_waitcnt lgkmcnt(\\n)\n"                                                                        \
	"  v_bfe_u32 \\l, \\x, 24, %[lbits]\n"
#define DFA_RES_HIT \
	"  v_cmp_eq_u32_sdwa vcc, \\w, \\l src0_sel:BYTE_\\sel src1_sel:DWORD\n" /* hit */
#ifdef DFA_EXP_NOSLOW /* (timing experiment: no lookups beyond LDS; results wrong unless LDS holds everything) */
#define DFA_RES_SLOWTEST ""
#define DFA_RES_SLOW "  s_mov_b64 s[52:53], 0\n"
#else
#define DFA_RES_SLOWTEST \
	"  v_cmp_le_u32_e64 s[40:41], %[hot], \\t\n"                                                        \
	"  v_cmp_gt_i32_e64 s[42:43], 0, \\s\n"             /* the state's LDS row is a partial one */
#define DFA_RES_SLOW \
	"  s_andn2_b64 s[42:43], s[42:43], vcc\n"                                                           \
	"  s_or_b64 s[42:43], s[42:43], s[40:41]\n"                                                         \
	"  s_mov_b64 s[52:53], 0\n"                                                                         \
	"  s_cmp_eq_u64 s[42:43], 0\n"                                                                      \
	"  s_cbranch_scc1 .Lfast%=_\\@\n"                                                                      \
	"  s_andn2_b64 s[50:51], vcc, s[42:43]\n"           /* ---- the global array decides ---- */        \
	"  s_mov_b64 s[56:57], s[42:43]\n"                  /* lane by lane, through the scalar cache: a    \
	                                                       vector load would wait for the stores too */  \
	".Lwf%=_\\@:\n"                                                                                     \
	"  s_ff1_i32_b64 s58, s[56:57]\n"                                                                   \
	"  v_readlane_b32 s59, \\t, s58\n"                  /* slot index (right for a row beyond LDS) */   \
	"  s_bitcmp1_b64 s[40:41], s58\n"                                                                   \
	"  s_cbranch_scc1 .Lwc%=_\\@\n"                                                                     \
	"  v_readlane_b32 s60, \\s, s58\n"                  /* a partial row: crow[base] + code */          \
	"  v_readlane_b32 s61, \\w, s58\n"                                                                  \
	"  s_and_b32 s60, s60, 0x7fffff\n"                                                                  \
	"  s_lshl_b32 s60, s60, 2\n"                                                                        \
	"  s_load_dword s59, %[crow], s60\n"                                                                \
	"  s_bfe_u32 s61, s61, (\\sel * 8) | 0x80000\n"                                                     \
	"  s_waitcnt lgkmcnt(0)\n"                                                                          \
	"  s_add_u32 s59, s59, s61\n"                                                                       \
	".Lwc%=_\\@:\n"                                                                                     \
	"  s_lshl_b32 s60, s59, 2\n"                                                                        \
	"  s_load_dword s60, %[slot], s60\n"                                                                \
	"  s_bitset0_b64 s[56:57], s58\n"                                                                   \
	"  s_mov_b32 m0, s58\n"                            /* (writelane takes one SGPR: the lane comes from m0) */ \
	"  s_waitcnt lgkmcnt(0)\n"                                                                          \
	"  v_writelane_b32 \\x, s60, m0\n"                                                                  \
	"  v_writelane_b32 \\t, s59, m0\n"                                                                 \
	"  s_cmp_lg_u64 s[56:57], 0\n"                                                                      \
	"  s_cbranch_scc1 .Lwf%=_\\@\n"                                                                     \
	"  s_mov_b64 exec, s[42:43]\n"                                                                      \
	"  v_bfe_u32 \\l, \\x, 24, %[lbits]\n"                                                              \
	"  v_cmp_eq_u32_sdwa vcc, \\w, \\l src0_sel:BYTE_\\sel src1_sel:DWORD\n"                            \
	"  s_mov_b64 s[52:53], vcc\n"                       /* their new event index is a global one */     \
	"  s_or_b64 vcc, vcc, s[50:51]\n"                                                                   \
	"  s_mov_b64 exec, s[38:39]\n"                                                                      \
	".Lfast%=_\\@:\n"
#endif
#ifdef DFA_EXP_NOLOG /* (timing experiment: no miss events; results wrong) */
#define DFA_RES_LOG ""
#else
#define DFA_RES_LOG \
	"  s_andn2_b64 s[44:45], s[38:39], vcc\n"           /* ---- the states left by a miss ---- */       \
	"  s_and_b64 s[46:47], s[44:45], \\uc\n"                                                            \
	"  s_cbranch_scc0 .Lnocu%=_\\@\n"                                                                      \
	"  s_mov_b64 exec, s[46:47]\n"                      /* event index beyond LDS: its counter */       \
	"  v_subrev_u32 v21, %[hot], \\u\n"                                                                 \
	"  v_lshlrev_b32 v21, 2, v21\n"                                                                     \
	"  global_atomic_add v21, v34, %[cmiss]\n"                                                          \
	".Lnocu%=_\\@:\n"                                                                                      \
	"  s_andn2_b64 exec, s[44:45], \\uc\n"                                                              \
	"  s_bcnt1_i32_b64 s48, exec\n"                                                                     \
	"  s_cmp_eq_u32 s48, 0\n"                          /* no lane: lane 0 stores an entry nobody counts, */ \
	"  s_cselect_b64 exec, 1, exec\n"                  /* so that every step issues one store (vmcnt)   */ \
	"  v_mbcnt_lo_u32_b32 v20, exec_lo, 0\n"                                                            \
	"  v_mbcnt_hi_u32_b32 v20, exec_hi, v20\n"          /* rank among them */                           \
	"  v_lshl_add_u32 v20, v20, 1, %[curm]\n"                                                           \
	"  global_store_short v20, \\u, %[basem]\n"                                                         \
	"  s_lshl1_add_u32 %[curm], s48, %[curm]\n"                                                         \
	"  s_mov_b64 exec, s[38:39]\n"
#endif
#ifdef DFA_EXP_NOARR /* (timing experiment: no arrival events; results wrong) */
#define DFA_RES_ARR ""
#else
#define DFA_RES_ARR \
	"  v_cmp_lt_u32_sdwa s[46:47], v33, \\x src0_sel:DWORD src1_sel:BYTE_2\n" /* kArr */                \
	"  s_and_b64 s[46:47], s[46:47], vcc\n"             /* ---- arrivals on virtual entries ---- */     \
	"  s_cbranch_scc0 .Lnoarr%=_\\@\n"                                                                     \
	"  s_andn2_b64 exec, s[46:47], s[52:53]\n"                                                          \
	"  v_mbcnt_lo_u32_b32 v20, exec_lo, 0\n"                                                            \
	"  v_mbcnt_hi_u32_b32 v20, exec_hi, v20\n"                                                          \
	"  v_lshl_add_u32 v20, v20, 1, %[cura]\n"                                                           \
	"  global_store_short v20, \\t, %[basea]\n"                                                         \
	"  s_bcnt1_i32_b64 s48, exec\n"                                                                     \
	"  s_lshl1_add_u32 %[cura], s48, %[cura]\n"                                                         \
	"  s_and_b64 exec, s[46:47], s[52:53]\n"                                                            \
	"  s_cbranch_execz .Larrdone%=_\\@\n"                                                                  \
	"  v_subrev_u32 v21, %[hot], \\t\n"                                                                 \
	"  v_lshlrev_b32 v21, 2, v21\n"                                                                     \
	"  global_atomic_add v21, v34, %[carr]\n"                                                           \
	".Larrdone%=_\\@:\n"                                                                                   \
	"  s_mov_b64 exec, s[38:39]\n"                                                                      \
	".Lnoarr%=_\\@:\n"
#endif
#define DFA_RES_TAIL \
	"  v_mul_u32_u24_sdwa \\m, \\w, v32 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_\\sel src1_sel:DWORD\n" \
	"  v_cndmask_b32_e32 \\s, \\m, \\x, vcc\n"          /* the next state */                            \
	"  v_cndmask_b32_sdwa \\u, \\w, \\t, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_\\sel src1_sel:DWORD\n" \
	"  s_mov_b64 \\uc, s[52:53]\n"                                                                      \
	".endm\n"
#define DFA_ASM_MACROS                                                                                  \
	".macro DFA_ISSUE w, sel, s, t, a, x\n"                                                             \
	"  v_and_b32 \\t, 0x7fffff, \\s\n"                                                                  \
	"  v_add_u32_sdwa \\t, \\w, \\t dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_\\sel src1_sel:DWORD\n" \
	"  v_lshlrev_b32 \\a, 2, \\t\n"                                                                     \
	"  ds_read_b32 \\x, \\a\n"                                                                          \
	".endm\n"                                                                                           \
	DFA_RES_HEAD DFA_RES_SLOWTEST DFA_RES_HIT DFA_RES_SLOW DFA_RES_LOG DFA_RES_ARR DFA_RES_TAIL
// chain j: W0 W1 N0 N1 T X L M = v[40 + 8 j ...]
#define DFA_I0(w, sel) "DFA_ISSUE " w ", " #sel ", %[S0], v44, v46, v45\n"
#define DFA_I1(w, sel) "DFA_ISSUE " w ", " #sel ", %[S1], v52, v54, v53\n"
#define DFA_I2(w, sel) "DFA_ISSUE " w ", " #sel ", %[S2], v60, v62, v61\n"
#define DFA_I3(w, sel) "DFA_ISSUE " w ", " #sel ", %[S3], v68, v70, v69\n"
#define DFA_R0(n, w, sel) "DFA_RESOLVE " #n ", " w ", " #sel ", %[S0], %[U0], v44, v45, v46, v47, %[UC0]\n"
#define DFA_R1(n, w, sel) "DFA_RESOLVE " #n ", " w ", " #sel ", %[S1], %[U1], v52, v53, v54, v55, %[UC1]\n"
#define DFA_R2(n, w, sel) "DFA_RESOLVE " #n ", " w ", " #sel ", %[S2], %[U2], v60, v61, v62, v63, %[UC2]\n"
#define DFA_R3(n, w, sel) "DFA_RESOLVE " #n ", " w ", " #sel ", %[S3], %[U3], v68, v69, v70, v71, %[UC3]\n"
// one byte of every chain resolved and the next one issued (a*: the window dwords of this byte, b*: of the next)
#define DFA_K1(a0, sel, b0, seln) DFA_R0(0, a0, sel) DFA_I0(b0, seln)
#define DFA_K2(a0, a1, sel, b0, b1, seln) DFA_R0(1, a0, sel) DFA_I0(b0, seln) DFA_R1(1, a1, sel) DFA_I1(b1, seln)
#define DFA_K4(a0, a1, a2, a3, sel, b0, b1, b2, b3, seln)                                               \
	DFA_R0(3, a0, sel) DFA_I0(b0, seln) DFA_R1(3, a1, sel) DFA_I1(b1, seln) DFA_R2(3, a2, sel) DFA_I2(b2, seln)   \
	DFA_R3(3, a3, sel) DFA_I3(b3, seln)
#define DFA_BODY1                                                                                       \
	DFA_I0("v40", 0) DFA_K1("v40", 0, "v40", 1) DFA_K1("v40", 1, "v40", 2) DFA_K1("v40", 2, "v40", 3)           \
	DFA_K1("v40", 3, "v41", 0) DFA_K1("v41", 0, "v41", 1) DFA_K1("v41", 1, "v41", 2) DFA_K1("v41", 2, "v41", 3) \
	DFA_R0(0, "v41", 3)
#define DFA_BODY2                                                                                       \
	DFA_I0("v40", 0) DFA_I1("v48", 0)                                                                   \
	DFA_K2("v40", "v48", 0, "v40", "v48", 1) DFA_K2("v40", "v48", 1, "v40", "v48", 2)                       \
	DFA_K2("v40", "v48", 2, "v40", "v48", 3) DFA_K2("v40", "v48", 3, "v41", "v49", 0)                       \
	DFA_K2("v41", "v49", 0, "v41", "v49", 1) DFA_K2("v41", "v49", 1, "v41", "v49", 2)                       \
	DFA_K2("v41", "v49", 2, "v41", "v49", 3) DFA_R0(1, "v41", 3) DFA_R1(0, "v49", 3)
#define DFA_BODY4                                                                                       \
	DFA_I0("v40", 0) DFA_I1("v48", 0) DFA_I2("v56", 0) DFA_I3("v64", 0)                                 \
	DFA_K4("v40", "v48", "v56", "v64", 0, "v40", "v48", "v56", "v64", 1)                                \
	DFA_K4("v40", "v48", "v56", "v64", 1, "v40", "v48", "v56", "v64", 2)                                \
	DFA_K4("v40", "v48", "v56", "v64", 2, "v40", "v48", "v56", "v64", 3)                                \
	DFA_K4("v40", "v48", "v56", "v64", 3, "v41", "v49", "v57", "v65", 0)                                \
	DFA_K4("v41", "v49", "v57", "v65", 0, "v41", "v49", "v57", "v65", 1)                                \
	DFA_K4("v41", "v49", "v57", "v65", 1, "v41", "v49", "v57", "v65", 2)                                \
	DFA_K4("v41", "v49", "v57", "v65", 2, "v41", "v49", "v57", "v65", 3)                                \
	DFA_R0(3, "v41", 3) DFA_R1(2, "v49", 3) DFA_R2(1, "v57", 3) DFA_R3(0, "v65", 3)
// loads of a piece (8 bytes per chain) into the registers W (current) or N (next): chain j at offset 512 j
#define DFA_LOAD1(r0) "global_load_dwordx2 v[" r0 "], %[voff], %[src]\n"
#define DFA_LOAD2(r0, r1) DFA_LOAD1(r0) "global_load_dwordx2 v[" r1 "], %[voff], %[src] offset:512\n"
#define DFA_LOAD4(r0, r1, r2, r3)                                                                       \
	DFA_LOAD2(r0, r1) "global_load_dwordx2 v[" r2 "], %[voff], %[src] offset:1024\n"                    \
	                  "global_load_dwordx2 v[" r3 "], %[voff], %[src] offset:1536\n"
#define DFA_SWAP1 "v_mov_b32 v40, v42\n v_mov_b32 v41, v43\n"
#define DFA_SWAP2 DFA_SWAP1 "v_mov_b32 v48, v50\n v_mov_b32 v49, v51\n"
#define DFA_SWAP4 DFA_SWAP2 "v_mov_b32 v56, v58\n v_mov_b32 v57, v59\n v_mov_b32 v64, v66\n v_mov_b32 v65, v67\n"
// The loop over the pieces of one wave block, from piece p on; left when the pieces are done or a
// stream batch may not take another piece's events (the C++ around it opens the next batch and
// comes back).  LOADW / LOADN / BODY / SWAP: the forms above for the number of chains; STEP = 512 * chains.
#ifdef DFA_EXP_NOLOG
#define DFA_PIECE_WAIT(STORES) "s_waitcnt vmcnt(0)\n"
#else
#define DFA_PIECE_WAIT(STORES) "s_waitcnt vmcnt(" #STORES ")\n" /* (in issue order: the piece came before the body's stores, one per step) */
#endif
#define DFA_LOOP(LOADW, LOADN, BODY, SWAP, STEP, STORES)                                                        \
	DFA_ASM_MACROS                                                                                      \
	"s_mov_b64 s[38:39], exec\n"                                                                        \
	"s_mov_b32 s55, m0\n"                               /* (the rare path borrows m0) */                 \
	"v_mov_b32 v32, %[stride]\n"                                                                        \
	"v_mov_b32 v33, 0x7f\n"                                                                             \
	"v_mov_b32 v34, 1\n"                                                                                \
	LOADW                                                                                               \
	"s_waitcnt vmcnt(0)\n"                                                                              \
	".Lpiece%=:\n"                                                                                      \
	"s_add_u32 s54, %[p], 1\n"                                                                          \
	"s_cmp_lt_u32 s54, %[pieces]\n"                                                                     \
	"s_cbranch_scc0 .Lnopf%=\n"                                                                         \
	"v_add_u32 %[voff], " #STEP ", %[voff]\n"           /* the next piece, under way during this one */ \
	LOADN                                                                                               \
	".Lnopf%=:\n"                                                                                       \
	BODY                                                                                                \
	DFA_PIECE_WAIT(STORES)                                                                              \
	SWAP                                                                                                \
	"s_mov_b32 %[p], s54\n"                                                                             \
	"s_cmp_ge_u32 %[p], %[pieces]\n"                                                                    \
	"s_cbranch_scc1 .Lout%=\n"                                                                          \
	"s_cmp_gt_u32 %[curm], %[high]\n"                                                                   \
	"s_cbranch_scc1 .Lout%=\n"                                                                          \
	"s_cmp_gt_u32 %[cura], %[high]\n"                                                                   \
	"s_cbranch_scc0 .Lpiece%=\n"                                                                        \
	".Lout%=:\n"                                                                                        \
	"s_mov_b32 m0, s55\n"                                                                               \
	"s_mov_b64 exec, s[38:39]\n"                                                                        \
	".purgem DFA_ISSUE\n"                                                                               \
	".purgem DFA_RESOLVE\n"
#define DFA_CLOBBERS                                                                                    \
	"v20", "v21", "v32", "v33", "v34", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51",  \
	    "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67",    \
	    "v68", "v69", "v70", "v71", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49",    \
	    "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "vcc", "scc", "memory"

template <int C>
__global__ void __launch_bounds__(1024) dfa_scan_asm_kernel(DfaDev A, DfaWork W, StreamDev Qm, StreamDev Qa, OutInit I) {
	extern __shared__ __attribute__((aligned(16))) uint32_t dfa_asm_smem[];
	LDS_AS uint32_t *img = (LDS_AS uint32_t *)dfa_asm_smem;
	if (I.out) // (nothing else touches the output vector before this kernel has ended)
		for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < I.V + 5; i += gridDim.x * blockDim.x)
			I.out[i] = i < 2 ? I.D : (i == I.V + 4 ? I.total_bytes : 0ull);
	{
		const u32x4 *src = reinterpret_cast<const u32x4 *>(A.hot);
		LDS_AS u32x4 *dst = (LDS_AS u32x4 *)img;
		for (uint32_t i = threadIdx.x; i < (A.hot_slots + 3) / 4; i += blockDim.x) dst[i] = src[i];
	}
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
	const uint32_t hot_slots = uni32(A.hot_slots), lbits = uni32(A.label_bits), stride = A.stride;
	constexpr uint32_t kHighBytes = (kDfaBatchEntries - 8 * C * 64) * 2;
	const unsigned long long slot_p = uni64((unsigned long long)(uintptr_t)A.slot), crow_p = uni64((unsigned long long)(uintptr_t)A.crow);
	const unsigned long long cmiss_p = uni64((unsigned long long)(uintptr_t)W.cold_miss), carr_p = uni64((unsigned long long)(uintptr_t)W.cold_arr);

	for (uint32_t wb = wave_id; wb < W.nwb; wb += total_waves) {
		uint32_t S[4] = {0, 0, 0, 0}, U[4] = {0, 0, 0, 0};
		unsigned long long uc[4] = {0, 0, 0, 0}; // lanes whose event index lies beyond the LDS array
#pragma unroll
		for (int j = 0; j < C; j++) {
			const uint2 e = W.entry[(wb * C + j) * 64 + lane];
			S[j] = e.x, U[j] = e.y;
			uc[j] = uni64(__builtin_amdgcn_ballot_w64(e.y >= hot_slots));
		}
		const unsigned long long src = uni64((unsigned long long)(uintptr_t)(W.tcodes + (size_t)wb * pieces * C * 64));
		uint32_t p = 0;
		while (p < pieces) {
			uint32_t voff = lane * 8 + p * (C * 512);
			uint32_t curm = uni32(lm.cur * 2), cura = uni32(la.cur * 2);
			const unsigned long long basem = uni64((unsigned long long)(uintptr_t)lm.base), basea = uni64((unsigned long long)(uintptr_t)la.base);
			uint32_t pp = uni32(p);
#define DFA_OPERANDS                                                                                                         \
	[p] "+s"(pp), [curm] "+s"(curm), [cura] "+s"(cura), [voff] "+v"(voff), [S0] "+v"(S[0]), [U0] "+v"(U[0]), [UC0] "+s"(uc[0]),       \
	    [S1] "+v"(S[1]), [U1] "+v"(U[1]), [UC1] "+s"(uc[1]), [S2] "+v"(S[2]), [U2] "+v"(U[2]), [UC2] "+s"(uc[2]), [S3] "+v"(S[3]),    \
	    [U3] "+v"(U[3]), [UC3] "+s"(uc[3])                                                                                            \
	    : [src] "s"(src), [pieces] "s"(pieces), [hot] "s"(hot_slots), [lbits] "s"(lbits), [high] "s"(kHighBytes),                    \
	      [basem] "s"(basem), [basea] "s"(basea), [slot] "s"(slot_p), [crow] "s"(crow_p), [cmiss] "s"(cmiss_p),                      \
	      [carr] "s"(carr_p), [stride] "v"(stride)                                                                                    \
	    : DFA_CLOBBERS
			if constexpr (C == 1) {
				asm volatile(DFA_LOOP(DFA_LOAD1("40:41"), DFA_LOAD1("42:43"), DFA_BODY1, DFA_SWAP1, 512, 8) : DFA_OPERANDS);
			} else if constexpr (C == 2) {
				asm volatile(DFA_LOOP(DFA_LOAD2("40:41", "48:49"), DFA_LOAD2("42:43", "50:51"), DFA_BODY2, DFA_SWAP2, 1024, 16) : DFA_OPERANDS);
			} else {
				asm volatile(DFA_LOOP(DFA_LOAD4("40:41", "48:49", "56:57", "64:65"), DFA_LOAD4("42:43", "50:51", "58:59", "66:67"),
				                      DFA_BODY4, DFA_SWAP4, 2048, 32)
				             : DFA_OPERANDS);
			}
#undef DFA_OPERANDS
			p = pp;
			lm.cur = curm / 2, la.cur = cura / 2;
			if (lm.cur * 2 > kHighBytes) lm.roll(Qm, lane);
			if (la.cur * 2 > kHighBytes) la.roll(Qa, lane);
		}
#pragma unroll
		for (int j = 0; j < C; j++) W.exit_a[(wb * C + j) * 64 + lane] = uint2{S[j] & (kBaseMask | A.inc_bit), U[j]};
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

// the counters of the events beyond the LDS array; cleared here for the next evaluation
__global__ void __launch_bounds__(256) dfa_cold_reduce_kernel(DfaDev A, uint32_t *cold_miss, uint32_t *cold_arr,
                                                              unsigned long long *out, uint32_t total_index) {
	const uint32_t n = A.nslots - A.hot_slots;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint32_t m = cold_miss[i], a = cold_arr[i];
		if (m) {
			book_event(A, A.ev_miss[A.hot_slots + i], (long long)m, out, total_index);
			cold_miss[i] = 0;
		}
		if (a) {
			book_event(A, A.ev_arr[A.hot_slots + i], (long long)a, out, total_index);
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
	const uint32_t lmask = (1u << F.A.label_bits) - 1;
	const uint32_t b = S & kBaseMask, t = b + c;
	uint32_t x, id = t;
	bool hit;
	if (t < F.A.hot_slots) {
		x = F.A.hot[t];
		hit = ((x >> 24) & lmask) == c;
		if (!hit && (S & F.A.inc_bit) != 0) {
			id = F.A.crow[b] + c;
			x = F.A.slot[id];
			hit = ((x >> 24) & lmask) == c;
		}
	} else {
		x = F.A.slot[t];
		hit = ((x >> 24) & lmask) == c;
	}
	if (hit) {
		if (x & kArr) book_event(F.A, F.A.ev_arr[id], sign, F.out, F.total_index);
		S = x & (kBaseMask | F.A.inc_bit);
		U = id;
	} else {
		book_event(F.A, F.A.ev_miss[U], sign, F.out, F.total_index);
		S = c * F.A.stride;
		U = c;
	}
}

// true when the two automata met inside the range (its exit stands); *exit_out = the true one's state at the end
__device__ bool dfa_dual_walk(const DfaFix &F, uint32_t g, uint2 want, uint2 have, uint2 *exit_out) {
	const uint32_t keep = kBaseMask | F.A.inc_bit;
	uint32_t Sa = want.x & keep, Ua = want.y, Sb = have.x & keep, Ub = have.y;
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
	static const bool use_asm = [] {
		const char *e = getenv("TOKSCORE_ASM");
		return !(e && atoi(e) == 0);
	}();
	if (use_asm) {
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

void launch_dfa_fixup(const DfaDev &A, const DfaWork &W, const uint8_t *codes, uint32_t *changed, uint32_t epoch,
                      unsigned long long *out, uint32_t total_index, hipStream_t stream) {
	if (W.nranges < 2) return;
	const uint32_t blocks = std::min<uint32_t>((W.nranges + 255) / 256, 64);
	DfaFix F{A, codes, W.range_bytes, W.nranges, W.entry, W.exit_a, W.exit_b, W.exit_tag, changed, epoch, out, total_index, blocks};
	hipLaunchKernelGGL(dfa_fixup_kernel, dim3(blocks), dim3(256), 0, stream, F);
}

void launch_dfa_cold_reduce(const DfaDev &A, const DfaWork &W, unsigned long long *out, uint32_t total_index, hipStream_t stream) {
	const uint32_t n = A.nslots - A.hot_slots;
	if (n == 0) return;
	hipLaunchKernelGGL(dfa_cold_reduce_kernel, dim3(std::min<uint32_t>((n + 255) / 256, 2048)), dim3(256), 0, stream, A, W.cold_miss,
	                   W.cold_arr, out, total_index);
}

bool plan_dfa_scan(uint32_t hot_slots, uint64_t prep_total, uint32_t num_cus, uint32_t chains_hint, DfaScanPlan *plan) {
	plan->lds_slots = hot_slots;
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
