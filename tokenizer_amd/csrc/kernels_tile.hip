// kernels_tile.hip -- the hot path: LDS-tiled exact greedy segmentation for gfx950, plus
// the token-stream histogram.
//
// scan_tile_kernel.  One wave (64 lanes) is an independent worker; the workgroup only
// shares the LDS copy of the hot trie slots.  A wave pulls work chunks (whole-document
// aligned where possible) from an atomic counter and streams each chunk through its own
// LDS tile of 64 segments x S bytes (S = 4 * odd, so the 64 lanes' byte reads fall in
// distinct banks when they run in step):
//   load     coalesced 16 B/lane global loads, ASCII lower-casing on the dwords in flight
//            (SolutionFile.cpp:87-88), next tile prefetched into registers meanwhile;
//   bounds   document starts inside the tile become a bitmap; a match window is clipped
//            at the next start (`input.size() - pos`, SolutionFile.cpp:90);
//   phase 1  every lane walks the greedy chain of its own segment from the segment's
//            first byte -- one trie lookup per iteration in one flat loop, so lanes never
//            wait for each other's token ends -- and logs the ids it emits in its own
//            column of the wave's LDS log block (no cross-lane traffic);
//   phase 2  lane k's true entry is lane k-1's exit overshoot.  Lanes whose guess (0) was
//            wrong dual-walk the true chain (+ entries) against the guessed one (- entries)
//            until the chains meet; rounds repeat while some exit still moves.
// A log block goes to HBM whenever one lane's column is full (unused cells are null
// entries); hist_kernel turns the stream into counts with an LDS histogram per workgroup.
// Results equal the serial loop exactly.
#include "kernels.h"

#include <algorithm>
#include <cstdlib>

namespace tokscore {

namespace {

constexpr uint32_t kNoLimit = 0x7FFFFFFFu;

// LDS pointers are 32-bit byte addresses; the walk keeps its positions as such addresses so
// that a byte read needs no address arithmetic.
__device__ __forceinline__ uint32_t lds_addr(const LDS_AS void *p) { return (uint32_t)(uintptr_t)p; }
template <class T> __device__ __forceinline__ LDS_AS T *lds_ptr(uint32_t a) { return (LDS_AS T *)(uintptr_t)a; }
constexpr uint32_t kLogDepth = 8; // stream entries per lane per log block
constexpr uint32_t kPendDepth = 16; // first tokens of a lane's speculative chain remembered for the repair

template <bool WIDE> struct QTraits;
template <> struct QTraits<false> {
	using entry_t = uint16_t;
	static constexpr uint32_t neg = 0x8000u;
	static constexpr uint32_t null_entry = 0x7FFFu;
	static constexpr uint32_t null_word = 0x7FFF7FFFu;
};
template <> struct QTraits<true> {
	using entry_t = uint32_t;
	static constexpr uint32_t neg = 0x80000000u;
	static constexpr uint32_t null_word = 0x7FFFFFFFu;
};

// next document start strictly after pos, looked up to `stop` (exclusive)
__device__ __forceinline__ uint32_t find_limit(const LDS_AS uint32_t *bm, uint32_t pos, uint32_t stop) {
	const uint32_t i = pos + 1;
	uint32_t w = i >> 5;
	uint32_t word = bm[w] >> (i & 31);
	if (word) return i + __builtin_ctz(word);
	for (w = w + 1; (w << 5) < stop; w++) {
		word = bm[w];
		if (word) return (w << 5) + __builtin_ctz(word);
	}
	return kNoLimit;
}

// The wave's log block in LDS: 64 * kLogDepth entries filled front to back.  The lanes that
// complete a token in a step append together: each takes the slot cur + (its rank among them),
// cur advances by their number -- all scalar / mbcnt arithmetic, no LDS atomics.  The block
// goes to the stream when the next step might not fit (cur > capacity - 64); unused cells are
// null entries (order is irrelevant to a histogram).
template <bool WIDE> struct LaneLog {
	using Q = QTraits<WIDE>;
	static constexpr uint32_t kEntries = 64 * kLogDepth;
	static constexpr uint32_t kHigh = kEntries - 64; // flush above this fill
	static constexpr uint32_t kShift = WIDE ? 2 : 1;  // log2(entry bytes)
	LDS_AS typename Q::entry_t *cells;
	uint32_t cur; // wave uniform: entries used

	__device__ __forceinline__ void init(LDS_AS typename Q::entry_t *c, uint32_t lane) {
		cells = c;
		clear(lane);
	}
	__device__ __forceinline__ void clear(uint32_t lane) {
		constexpr uint32_t words = kEntries * sizeof(typename Q::entry_t) / 4;
		LDS_AS uint32_t *w = (LDS_AS uint32_t *)cells;
		for (uint32_t i = lane; i < words; i += 64) w[i] = Q::null_word;
		cur = 0;
	}
	// wave-uniform call: the lanes with `pred` append `val`
	__device__ __forceinline__ void put(bool pred, uint32_t val) {
		const unsigned long long m = __builtin_amdgcn_ballot_w64(pred);
		if (pred) {
			const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
			cells[cur + rank] = (typename Q::entry_t)val;
		}
		cur += (uint32_t)__builtin_popcountll(m);
	}
	// Stream space is handed out in batches of kStreamBatch blocks, so the single
	// allocation word sees one atomic per batch, not per block (one word sustains only
	// ~90 atomics/us chip-wide).  fill[batch] records how many blocks of a batch hold data.
	uint32_t batch = 0xFFFFFFFFu; // wave uniform
	uint32_t used = kStreamBatch; // wave uniform: blocks written in the current batch

	// wave-uniform call: send the block out and start an empty one
	__device__ __forceinline__ void flush(const StreamDev &S, uint32_t lane) {
		constexpr uint32_t vecs = kEntries * sizeof(typename Q::entry_t) / 16; // 16-byte vectors
		if (used == kStreamBatch) {
			if (lane == 0 && batch < S.capacity_batches) S.fill[batch] = kStreamBatch;
			uint32_t b = 0;
			if (lane == 0) b = atomicAdd(&S.cursor[0], 1u);
			batch = __builtin_amdgcn_readfirstlane(b);
			used = 0;
			if (batch >= S.capacity_batches && lane == 0) S.cursor[1] = 1u; // overflow: reported, never dropped silently
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
	__device__ __forceinline__ void flush_if_full(const StreamDev &S, uint32_t lane) {
		if (cur > kHigh) flush(S, lane);
	}
	__device__ __forceinline__ void finish(const StreamDev &S, uint32_t lane) {
		if (cur != 0) flush(S, lane);
		if (lane == 0 && batch < S.capacity_batches) S.fill[batch] = used;
	}
};

struct WaveCtx {
	const LDS_AS u32x2 *lds_trie;
	const u32x2 *g_trie;
	const LDS_AS uint8_t *tile;    // tile-relative byte offsets
	const LDS_AS uint32_t *bitmap; // bit i = a document starts at tile byte i
	uint32_t lds_slots;
	uint32_t root_base;
	uint32_t unknown_id;
};

// One greedy chain walker (per lane).  Positions are LDS byte addresses inside the tile.
struct Walker {
	uint32_t cp;   // address of the next byte to look up (token start + depth)
	uint32_t nb;   // base of the current trie node
	uint32_t lend; // address just past the best token so far (start + 1 while there is none)
	uint32_t lid;  // its id (UNKNOWN while there is none, SolutionFile.cpp:92-94)
	__device__ __forceinline__ void start(uint32_t at, uint32_t root_base, uint32_t unknown_id) {
		cp = at;
		nb = root_base;
		lend = at + 1;
		lid = unknown_id;
	}
};

// One trie lookup of the token in progress.  Returns true when the token is complete (miss,
// window clipped at `lim`, or a leaf was reached): then it is [start, w.cp) with id w.lid,
// w.cp is the next token's start and w.nb is the root again; the caller re-arms w.lend /
// w.lid.
__device__ __forceinline__ bool walk_step(const WaveCtx &W, Walker &w, uint32_t lim) {
	const uint32_t c = *lds_ptr<const uint8_t>(w.cp);
	const uint32_t t = w.nb + c;
	const bool cold = t >= W.lds_slots;
	u32x2 e = W.lds_trie[cold ? 0u : t];
	if (cold) e = W.g_trie[t]; // cold slot: L2 (t < nslots: base + 255 is always in range)
	const bool hit = (e.x & kSlotKeyMask) == (c | kSlotValid) && w.cp < lim;
	const uint32_t cp1 = w.cp + 1;
	const bool term = hit && e.y != 0;
	w.lend = term ? cp1 : w.lend;
	w.lid = term ? e.y : w.lid;
	const uint32_t nbase = e.x >> kSlotBaseShift;
	const bool cont = hit && nbase != 0; // a leaf ends the token without another lookup
	w.cp = cont ? cp1 : w.lend;
	w.nb = cont ? nbase : W.root_base;
	return !cont;
}

// ---- the two hot loops, written in gfx950 assembly --------------------------------------
// The scan is bound by instruction issue (a SIMD issues about one instruction per four
// cycles over all of its waves), so what counts is the number of instructions per trie
// lookup, scalar ones included.  The compiler's code for the loops below (kept in C++ for the
// rarely taken dual walk) needs about 100 per lookup; these need about 50.
// Conventions: exec = lanes still walking; v24-v27, s[38:43] are scratch; a lane's state is
// the Walker (cp, nb, lend, lid); the loop is left when no lane walks any more or when
// `stall` is non-zero (a log column is full, or a lane's next token starts at or after its
// document limit), which the C++ around it handles before it re-enters.
//
// One trie lookup for the lanes in exec (= %[act]); v24 holds each lane's next byte, read
// at the end of the previous lookup so that its LDS latency hides behind the token-end
// block.  Afterwards vcc = lanes whose token goes on, s[42:43] = lanes whose token is
// complete, scc = any such; cp is every lane's next position (after the byte when the token
// goes on, the next token's start otherwise) and its byte is on the way into v24.
#define TOK_ASM_LOOKUP(TAG)                                                            \
	"s_waitcnt lgkmcnt(0)\n"                                                           \
	"v_add_lshl_u32 v25, %[nb], v24, 3\n"      /* byte offset of slot nb + c        */ \
	"v_cmp_gt_u32 vcc, %[ldsb], v25\n"         /* slot staged in LDS?               */ \
	"s_andn2_b64 s[40:41], exec, vcc\n"                                                \
	"s_mov_b64 exec, vcc\n"                                                            \
	"ds_read_b64 v[26:27], v25\n"                                                      \
	"s_cbranch_scc0 .Lhot" TAG "%=\n"                                                  \
	"s_mov_b64 exec, s[40:41]\n"                                                       \
	"global_load_dwordx2 v[26:27], v25, %[trie]\n" /* cold slot: L2                 */ \
	".Lhot" TAG "%=:\n"                                                                \
	"s_mov_b64 exec, %[act]\n"                                                         \
	"v_or_b32 v24, 0x100, v24\n"               /* (these three while the slot loads) */ \
	"v_cmp_lt_u32 s[40:41], %[cp], %[lim]\n"   /* byte inside the document          */ \
	"v_add_u32 %[cp], 1, %[cp]\n"                                                      \
	"s_waitcnt vmcnt(0) lgkmcnt(0)\n"                                                  \
	"v_and_b32 v25, 0x1ff, v26\n"                                                      \
	"v_cmp_eq_u32 vcc, v25, v24\n"             /* label matches and slot valid      */ \
	"s_and_b64 s[40:41], s[40:41], vcc\n"      /* hit                               */ \
	"v_cmp_ne_u32 vcc, 0, v27\n"                                                       \
	"s_and_b64 vcc, vcc, s[40:41]\n"           /* hit on a token end                */ \
	"v_cndmask_b32 %[lend], %[lend], %[cp], vcc\n"                                     \
	"v_cndmask_b32 %[lid], %[lid], v27, vcc\n"                                         \
	"v_lshrrev_b32 %[nb], 9, v26\n"                                                    \
	"v_cmp_ne_u32 vcc, 0, %[nb]\n"                                                     \
	"s_and_b64 vcc, vcc, s[40:41]\n"           /* hit on a node with children       */ \
	"s_andn2_b64 s[42:43], exec, vcc\n"        /* token complete                    */ \
	"v_cndmask_b32 %[cp], %[lend], %[cp], vcc\n"                                       \
	"ds_read_u8 v24, %[cp]\n"

#define TOK_ASM_ENTER                                                                  \
	"s_mov_b64 s[38:39], exec\n"                                                       \
	"s_mov_b64 %[act], %[actin]\n"                                                     \
	"s_mov_b32 %[cur], %[curin]\n"                                                     \
	"s_mov_b64 exec, %[act]\n"                                                         \
	"ds_read_u8 v24, %[cp]\n"

// (the byte read in flight targets v24, which the compiler may reuse at once: wait for it)
#define TOK_ASM_LEAVE                                                                  \
	"s_waitcnt lgkmcnt(0)\n"                                                           \
	"s_mov_b64 exec, s[38:39]\n"

// The lanes in exec (= s[42:43], the lanes that completed a token) append their ids to the log.
#define TOK_ASM_APPEND(DSW, SHIFT)                                                     \
	"v_mbcnt_lo_u32_b32 v25, s42, 0\n"                                                 \
	"v_mbcnt_hi_u32_b32 v25, s43, v25\n"       /* rank among them                   */ \
	"v_add_u32 v25, %[cur], v25\n"                                                     \
	"v_lshl_add_u32 v25, v25, " SHIFT ", %[logb]\n"                                    \
	DSW " v25, %[lid]\n"                                                               \
	"s_bcnt1_i32_b64 s40, s[42:43]\n"                                                  \
	"s_add_u32 %[cur], %[cur], s40\n"

// Leave the loop when a lane's next token starts at or after its document limit, when the
// log may not take another step, or when no lane walks any more.
#define TOK_ASM_NEXT(HIGH)                                                             \
	"v_cmp_ge_u32 s[40:41], %[cp], %[lim]\n"                                           \
	"s_and_b64 %[stall], s[40:41], %[act]\n"                                           \
	"s_mov_b64 exec, %[act]\n"                                                         \
	"s_cbranch_scc1 .Lout%=\n"                                                         \
	"s_cmp_gt_u32 %[cur], " HIGH "\n"                                                  \
	"s_cbranch_scc1 .Lout%=\n"                                                         \
	"s_cmp_lg_u64 %[act], 0\n"                                                         \
	"s_cbranch_scc1 .Ltop%=\n"                                                         \
	".Lout%=:\n"

// Phase 1 (speculative chain): log the id, remember the token's end while it lies within 64
// bytes of the segment start, re-arm the walker, retire the lane at its segment end.
#define TOK_ASM_PHASE1(DSW, ROW, SHIFT, HIGH)                                                       \
	asm volatile(TOK_ASM_ENTER                                                         \
	             ".Ltop%=:\n" TOK_ASM_LOOKUP("a")                                      \
	             "s_cbranch_scc0 .Ltop%=\n"                                            \
	             "s_mov_b64 exec, s[42:43]\n"                                          \
	             "v_mov_b32 %[nb], %[root]\n"                                          \
	             TOK_ASM_APPEND(DSW, SHIFT)                                            \
	             "v_sub_u32 v25, %[cp], %[seg]\n"          /* end offset in segment */ \
	             "v_cmp_gt_u32 vcc, 64, v25\n"                                         \
	             "v_cmp_lt_u32 s[40:41], %[pw], %[pwend]\n"                            \
	             "s_and_b64 s[40:41], vcc, s[40:41]\n"      /* remembered            */ \
	             "v_cmp_gt_u32 vcc, 32, v25\n"                                         \
	             "s_and_b64 exec, s[40:41], vcc\n"                                     \
	             "v_lshl_or_b32 %[mlo], 1, v25, %[mlo]\n"                              \
	             "s_andn2_b64 exec, s[40:41], vcc\n"                                   \
	             "v_lshl_or_b32 %[mhi], 1, v25, %[mhi]\n"  /* shifts by offset - 32 */ \
	             "s_mov_b64 exec, s[40:41]\n"                                          \
	             DSW " %[pw], %[lid]\n"                                                \
	             "v_add_u32 %[pw], " ROW ", %[pw]\n"                                   \
	             "s_mov_b64 exec, s[42:43]\n"                                          \
	             "v_add_u32 %[lend], 1, %[cp]\n"                                       \
	             "v_mov_b32 %[lid], %[unk]\n"                                          \
	             "v_cmp_ge_u32 vcc, %[cp], %[hi]\n"                                    \
	             "s_andn2_b64 %[act], %[act], vcc\n"                                   \
	             TOK_ASM_NEXT(HIGH) TOK_ASM_LEAVE                                      \
	             : [cp] "+v"(wk.cp), [nb] "+v"(wk.nb), [lend] "+v"(wk.lend), [lid] "+v"(wk.lid),   \
	               [pw] "+v"(pa_w), [mlo] "+v"(mask_lo), [mhi] "+v"(mask_hi), [act] "=&s"(act_out),   \
	               [cur] "=&s"(cur_out), [stall] "=&s"(stall)                                         \
	             : [lim] "v"(lim_a), [hi] "v"(hi_a), [seg] "v"(seg_a), [curin] "s"(uni32(log.cur)), \
	               [logb] "s"(log_a),                                                              \
	               [pwend] "v"(pend_end), [actin] "s"(uni64(act)), [ldsb] "s"(lds_trie_bytes), \
	               [trie] "s"(W.g_trie), [unk] "s"(W.unknown_id), [root] "s"(W.root_base)          \
	             : "v24", "v25", "v26", "v27", "s38", "s39", "s40", "s41", "s42", "s43", "vcc",    \
	               "scc", "memory")

// Phase 2, first round (true chain of a lane whose entry guess was wrong): log the id, stop
// at the first token end that the speculative chain also had (`mask`) or at `fasthi`.
#define TOK_ASM_REPAIR(DSW, ROW, SHIFT, HIGH)                                                       \
	asm volatile(TOK_ASM_ENTER                                                         \
	             ".Ltop%=:\n" TOK_ASM_LOOKUP("b")                                      \
	             "s_cbranch_scc0 .Ltop%=\n"                                            \
	             "s_mov_b64 exec, s[42:43]\n"                                          \
	             "v_mov_b32 %[nb], %[root]\n"                                          \
	             TOK_ASM_APPEND(DSW, SHIFT)                                            \
	             "v_sub_u32 v25, %[cp], %[seg]\n"                                      \
	             "v_cmp_gt_u32 vcc, 32, v25\n"                                         \
	             "s_nop 1\n"                                /* vcc: VALU to VALU     */ \
	             "v_cndmask_b32 v26, %[mhi], %[mlo], vcc\n"                            \
	             "v_lshrrev_b32 v26, v25, v26\n"                                       \
	             "v_and_b32 v26, 1, v26\n"                                             \
	             "v_cmp_gt_u32 vcc, 64, v25\n"                                         \
	             "v_cmp_eq_u32 s[40:41], 1, v26\n"                                     \
	             "s_and_b64 s[40:41], s[40:41], vcc\n"      /* met the counted chain */ \
	             "s_or_b64 %[merged], %[merged], s[40:41]\n"                           \
	             "v_cmp_lt_u32 vcc, %[cp], %[fasthi]\n"                                \
	             "s_andn2_b64 vcc, vcc, s[40:41]\n"         /* goes on               */ \
	             "s_andn2_b64 %[act], %[act], s[42:43]\n"                              \
	             "s_or_b64 %[act], %[act], vcc\n"                                      \
	             "v_add_u32 %[lend], 1, %[cp]\n"                                       \
	             "v_mov_b32 %[lid], %[unk]\n"                                          \
	             TOK_ASM_NEXT(HIGH) TOK_ASM_LEAVE                                      \
	             : [cp] "+v"(wk.cp), [nb] "+v"(wk.nb), [lend] "+v"(wk.lend), [lid] "+v"(wk.lid),   \
	               [act] "=&s"(act_out), [merged] "+s"(merged_m), [cur] "=&s"(cur_out),             \
	               [stall] "=&s"(stall)                                                            \
	             : [lim] "v"(lim_a), [fasthi] "v"(fast_hi), [seg] "v"(seg_a),                      \
	               [curin] "s"(uni32(log.cur)), [logb] "s"(log_a),                             \
	               [mlo] "v"(mask_lo), [mhi] "v"(mask_hi), [actin] "s"(uni64(act)),                \
	               [ldsb] "s"(lds_trie_bytes),                                                     \
	               [trie] "s"(W.g_trie), [unk] "s"(W.unknown_id), [root] "s"(W.root_base)          \
	             : "v24", "v25", "v26", "v27", "s38", "s39", "s40", "s41", "s42", "s43", "vcc",    \
	               "scc", "memory")

// Per-wave time breakdown for tuning (make -C tokenizer_amd/csrc prof; tools/_prof/run_prof.py):
// cycles per phase, chunks taken and finish time of every wave, left in g_prof.  Compiled out
// of the product library.
#ifdef TOK_PROFILE
__device__ unsigned long long g_prof[4096 * 8];
#define PROF_NOW() ((unsigned long long)__builtin_readcyclecounter())
#define PROF_ADD(k, since)                          \
	do {                                            \
		const unsigned long long n_ = PROF_NOW();   \
		prof_acc[k] += n_ - (since);                \
		(since) = n_;                               \
	} while (0)
#else
#define PROF_NOW() 0ull
#define PROF_ADD(k, since) \
	do {                   \
	} while (0)
#endif

template <bool WIDE>
__global__ void __launch_bounds__(1024) scan_tile_kernel(TrieDev T, CorpusDev C, ChunkState S, StreamDev Qd,
                                                         const uint64_t *chunk_first_doc, uint32_t seg_bytes,
                                                         uint32_t halo_bytes, uint32_t max_len, uint32_t wave_lds_bytes,
                                                         uint64_t readable_end, const uint32_t *sel_list, const uint32_t *sel_count) {
	using Q = QTraits<WIDE>;
	extern __shared__ __attribute__((aligned(16))) unsigned char tile_smem[];
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	LDS_AS unsigned char *lds = (LDS_AS unsigned char *)tile_smem;

	// ---- stage the hot trie slots in LDS (shared by the workgroup's waves) ----
	{
		LDS_AS u32x2 *dst = (LDS_AS u32x2 *)lds;
		const u32x2 *src = reinterpret_cast<const u32x2 *>(T.slots);
		for (uint32_t i = threadIdx.x; i < T.lds_slots; i += blockDim.x) dst[i] = src[i];
	}
	__syncthreads();

	const uint32_t tile_span = 64 * seg_bytes;            // bytes owned by one tile
	const uint32_t tile_cap = tile_span + halo_bytes + 16; // bytes staged (halo + alignment slack)
	const uint32_t n_vec = (tile_cap + 15) / 16;           // 16-byte vectors per tile
	const uint32_t bm_words = (tile_cap + 31) / 32 + 1;
	LDS_AS unsigned char *wbase = lds + T.lds_slots * 8 + wave * wave_lds_bytes;
	LDS_AS unsigned char *tile = wbase;
	LDS_AS uint32_t *bitmap = (LDS_AS uint32_t *)(wbase + n_vec * 16);
	LaneLog<WIDE> log;
	log.init((LDS_AS typename Q::entry_t *)(wbase + n_vec * 16 + ((bm_words * 4 + 15) & ~15u)), lane);
	// ids of the first kPendDepth tokens of every lane's speculative chain, cell (j, lane)
	const uint32_t pend_a = lds_addr(log.cells + 64 * kLogDepth + lane);
	const uint32_t log_a = uni32(lds_addr(log.cells));
	constexpr uint32_t kRow = 64 * sizeof(typename Q::entry_t);
	const uint32_t tile_a = lds_addr(tile);
	const uint32_t lds_trie_bytes = T.lds_slots * 8;

	WaveCtx W;
	W.lds_trie = (const LDS_AS u32x2 *)lds;
	W.g_trie = reinterpret_cast<const u32x2 *>(T.slots);
	W.tile = tile;
	W.bitmap = bitmap;
	W.lds_slots = T.lds_slots;
	W.root_base = T.root_base;
	W.unknown_id = T.unknown_id;

	unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	unsigned long long prof_t = PROF_NOW();
	const unsigned long long prof_start = prof_t;
	(void)prof_acc, (void)prof_t, (void)prof_start;
	// (a grid smaller than kWorkQueues -- a partition that reports few CUs -- serves as many heads as it has workgroups)
	const uint32_t n_queues = gridDim.x < kWorkQueues ? gridDim.x : kWorkQueues;
	const uint32_t queue = blockIdx.x % n_queues;
	const uint32_t n_work = sel_list ? sel_count[0] : S.nchunks; // chunks in the queue
	const uint32_t min_seg = 4 * (((max_len + 4 + 3) / 4) | 1u); // shortest legal lane segment
	constexpr int kMaxVec = 7; // register prefetch depth: up to 7 x 16 B per lane
	const bool reg_prefetch = n_vec <= 64 * kMaxVec;

	for (;;) {
		// ---- next work chunk ----
		// work queue sharded over n_queues heads (one 128-byte line each): chunk j belongs
		// to head j % n_queues, workgroup b pulls from head b % n_queues
		uint32_t j = 0;
		if (lane == 0) j = atomicAdd(&Qd.cursor[kQueueHeadWord + queue * kQueueHeadStride], 1u);
		j = __builtin_amdgcn_readfirstlane(j) * n_queues + queue;
		if (j >= n_work) break;
		if (sel_list) j = sel_list[j]; // incremental re-score: the queue runs over the selected chunks only
		PROF_ADD(0, prof_t); // queue
		prof_acc[7]++;
		const uint64_t begin = S.begin[j];
		const uint64_t end = S.begin[j + 1];
		uint64_t di = chunk_first_doc[j]; // index of the first document offset >= begin
		uint32_t carry = 0;              // entry overshoot of lane 0 (0 = speculated at chunk start)
		uint32_t chunk_exit = 0;

		uint64_t a0 = begin & ~15ull; // 16-byte aligned tile base
		// The chunk is cut into equal tiles: as few as fit the LDS tile, each with the same
		// lane segment length (4 * odd bytes, longer than any token), so that no chunk ends
		// in a nearly empty tile.
		uint32_t seg_c = seg_bytes;
		{
			const uint64_t len = end - a0;
			const uint64_t nt = (len + tile_span - 1) / tile_span;
			const uint32_t per = (uint32_t)((len + nt - 1) / nt);
			seg_c = 4 * ((((per + 63) / 64 + 3) / 4) | 1u);
			if (seg_c < min_seg) seg_c = min_seg;
			if (seg_c > seg_bytes) seg_c = seg_bytes;
		}
		const uint32_t span_c = 64 * seg_c;                           // bytes owned by one tile of this chunk
		const uint32_t n_vec_c = (span_c + halo_bytes + 16 + 15) / 16; // 16-byte vectors staged per tile
		u32x4 pre[kMaxVec];
		auto load_tile_regs = [&](uint64_t base_abs) {
#pragma unroll
			for (int i = 0; i < kMaxVec; i++) {
				const uint32_t v = lane + 64 * i;
				const uint64_t a = base_abs + (uint64_t)v * 16;
				u32x4 x = {0, 0, 0, 0};
				if (v < n_vec_c && a + 16 <= readable_end) x = *reinterpret_cast<const u32x4 *>(C.bytes + a);
				pre[i] = x;
			}
		};
		if (reg_prefetch) load_tile_regs(a0);

		while (a0 < end) {
			// ---- tile bytes -> LDS (lower-cased) ----
			wave_sync();
			if (reg_prefetch) {
#pragma unroll
				for (int i = 0; i < kMaxVec; i++) {
					const uint32_t v = lane + 64 * i;
					if (v < n_vec_c) {
						u32x4 x = pre[i];
						x.x = lower4(x.x);
						x.y = lower4(x.y);
						x.z = lower4(x.z);
						x.w = lower4(x.w);
						((LDS_AS u32x4 *)tile)[v] = x;
					}
				}
				const uint64_t next_a0 = a0 + span_c;
				if (next_a0 < end) load_tile_regs(next_a0); // in flight while this tile is walked
			} else {
				for (uint32_t v = lane; v < n_vec_c; v += 64) {
					const uint64_t a = a0 + (uint64_t)v * 16;
					u32x4 x = {0, 0, 0, 0};
					if (a + 16 <= readable_end) x = *reinterpret_cast<const u32x4 *>(C.bytes + a);
					x.x = lower4(x.x);
					x.y = lower4(x.y);
					x.z = lower4(x.z);
					x.w = lower4(x.w);
					((LDS_AS u32x4 *)tile)[v] = x;
				}
			}
			// ---- document-start bitmap ----
			bool has_bounds = false; // wave uniform: any document start inside the staged bytes
			{
				// offsets are sorted, so the ones inside [a0, a0 + tile_cap) are a run starting at
				// di; di itself advances to the first offset at/after the NEXT tile's base
				const uint64_t tile_end_abs = a0 + (uint64_t)n_vec_c * 16;
				const uint64_t next_abs = a0 + span_c;
				uint64_t scan = di;
				bool di_done = false;
				bool zeroed = false;
				for (;;) {
					const uint64_t idx = scan + lane;
					uint64_t o = ~0ull;
					if (idx <= C.D) o = C.off[idx];
					const bool in_tile = o < tile_end_abs; // o >= begin >= a0 because idx >= di
					const unsigned long long in_mask = __ballot(in_tile);
					if (in_mask && !zeroed) {
						for (uint32_t w = lane; w < bm_words; w += 64) bitmap[w] = 0;
						wave_sync();
						zeroed = true;
						has_bounds = true;
					}
					if (in_tile) {
						const uint32_t b = (uint32_t)(o - a0);
						atomicOr((uint32_t *)&bitmap[b >> 5], 1u << (b & 31));
					}
					if (!di_done) {
						const uint32_t cnt = (uint32_t)__builtin_popcountll(__ballot(o < next_abs));
						di += cnt;
						di_done = cnt < 64;
					}
					if (in_mask != ~0ull) break;
					scan += 64;
				}
			}
			wave_sync();

			PROF_ADD(1, prof_t); // tile load + bitmap
			// ---- phase 1: speculative chain per lane segment ----
			const uint32_t off0 = (uint32_t)(begin > a0 ? begin - a0 : 0); // only the chunk's first tile
			const uint64_t rem = end - a0;
			const uint32_t tile_hi = rem < span_c ? (uint32_t)rem : span_c;
			// the chunk's last tile may still be a little short: shorter segments again, so
			// that as many lanes as possible share it
			uint32_t seg_t = seg_c;
			if (tile_hi < span_c) {
				const uint32_t dw = ((tile_hi + 63) / 64 + 3) / 4;
				seg_t = 4 * (dw | 1u);
				if (seg_t < min_seg) seg_t = min_seg;
				if (seg_t > seg_c) seg_t = seg_c;
			}
			const uint32_t seg_nom = lane * seg_t;
			uint32_t seg_lo = seg_nom > off0 ? seg_nom : off0;
			uint32_t seg_hi = seg_nom + seg_t < tile_hi ? seg_nom + seg_t : tile_hi;
			const bool nonempty = seg_lo < seg_hi;
			if (!nonempty) seg_hi = seg_lo;
			const uint32_t stop = seg_hi + max_len; // no walk of this segment reads at/after stop
			const uint32_t seg_a = tile_a + seg_nom;  // LDS addresses from here on
			const uint32_t hi_a = tile_a + seg_hi;
			uint32_t cur_entry = (lane == 0) ? carry : 0u;
			Walker wk;
			wk.start(tile_a + seg_lo + cur_entry, W.root_base, W.unknown_id);
			uint32_t lim_a = kNoLimit;
			bool active = nonempty && wk.cp < hi_a;
			if (has_bounds && active) lim_a = tile_a + find_limit(bitmap, wk.cp - tile_a, stop);
			// The ends of the chain's first tokens (at most kPendDepth, within 64 bytes of the
			// segment start) are remembered: end bit in `mask`, id in the lane's pend column;
			// `bound` = the last such end.  The repair then needs no second walk over them.
			uint32_t mask_lo = 0, mask_hi = 0;
			uint32_t pa_w = pend_a;
			const uint32_t pend_end = pend_a + kPendDepth * kRow;
			const uint32_t first_a = wk.cp;
			// The hot loop (assembly, see TOK_ASM_PHASE1) holds nothing but the walk; it is left
			// whenever a log column is full or a lane's next token starts past its document
			// limit, both handled out here (wk.lend > wk.cp tells a lane at a token start from
			// one in the middle of a token, whose cp may legitimately have reached its limit).
			for (unsigned long long act = __builtin_amdgcn_ballot_w64(active); act != 0;) {
				// (the compiler takes the loop-carried mask for divergent, hence uni64() on
				// the way in and separate in / out operands)
				unsigned long long stall, act_out;
				uint32_t cur_out;
				if constexpr (WIDE) TOK_ASM_PHASE1("ds_write_b32", "256", "2", "448");
				else TOK_ASM_PHASE1("ds_write_b16", "128", "1", "448");
				(void)stall;
				PROF_ADD(2, prof_t); // phase 1 loop
				act = act_out;
				log.cur = cur_out;
				log.flush_if_full(Qd, lane);
				if (has_bounds && ((act >> lane) & 1ull) && wk.lend > wk.cp && wk.cp >= lim_a)
					lim_a = tile_a + find_limit(bitmap, wk.cp - tile_a, stop);
				PROF_ADD(3, prof_t); // log flush, limit refresh
			}
			const unsigned long long mask = ((unsigned long long)mask_hi << 32) | mask_lo;
			const uint32_t bound = mask ? seg_a + (63u - (uint32_t)__builtin_clzll(mask)) : first_a;
			uint32_t cur_exit = nonempty ? wk.cp - hi_a : 0u;

			// ---- phase 2: settle entries ----
			// Every lane whose counted chain assumed another entry than its neighbour's current
			// exit repairs itself, all such lanes at once; a repair that never met moves that
			// lane's exit, so the round repeats until nothing changes.  Lane k is final after at
			// most k+1 rounds (lane 0's entry is known), usually after the first.
			for (uint32_t round = 0;; round++) {
				uint32_t want = __shfl_up(cur_exit, 1, 64);
				if (lane == 0) want = cur_entry;
				const bool need = nonempty && want != cur_entry;
				if (__builtin_amdgcn_ballot_w64(need) == 0) break;
				uint32_t pa = seg_a + want;      // true chain  (+)
				uint32_t pb = seg_a + cur_entry; // counted chain (-)
				if (round == 0) {
					// First round: the counted chain is the speculative one (entry 0) whose first
					// token ends are in `mask`.  Walk only the true chain until it lands on one of
					// them: from there on both chains are the same chain.
					const uint32_t fast_hi = bound < hi_a ? bound : hi_a;
					bool merged = need && want < 64 && ((mask >> (want & 63)) & 1ull);
					bool fw = need && !merged && pa < fast_hi;
					wk.start(pa, W.root_base, W.unknown_id);
					lim_a = kNoLimit;
					if (has_bounds && fw) lim_a = tile_a + find_limit(bitmap, pa - tile_a, stop);
					unsigned long long merged_m = __builtin_amdgcn_ballot_w64(merged);
					for (unsigned long long act = __builtin_amdgcn_ballot_w64(fw); act != 0;) {
						unsigned long long stall, act_out;
						uint32_t cur_out;
						if constexpr (WIDE) TOK_ASM_REPAIR("ds_write_b32", "256", "2", "448");
						else TOK_ASM_REPAIR("ds_write_b16", "128", "1", "448");
						(void)stall;
						PROF_ADD(4, prof_t); // repair loop
						act = act_out;
						log.cur = cur_out;
						log.flush_if_full(Qd, lane);
						if (has_bounds && ((act >> lane) & 1ull) && wk.lend > wk.cp && wk.cp >= lim_a)
							lim_a = tile_a + find_limit(bitmap, wk.cp - tile_a, stop);
						PROF_ADD(3, prof_t);
					}
					merged = (merged_m >> lane) & 1ull;
					pa = wk.cp;
					// take back the speculative tokens that end at or before the meeting point
					unsigned long long taken = mask;
					if (merged) taken &= (2ull << ((pa - seg_a) & 63)) - 1ull;
					const uint32_t rank = need ? (uint32_t)__builtin_popcountll(taken) : 0u;
					for (uint32_t jn = 0; __builtin_amdgcn_ballot_w64(jn < rank); jn++) {
						const uint32_t row = jn < rank ? jn : 0u;
						log.put(jn < rank, (uint32_t)*lds_ptr<typename Q::entry_t>(pend_a + row * kRow) | Q::neg);
						log.flush_if_full(Qd, lane);
					}
					// not met yet: the generic dual walk below continues from (pa, bound)
					pb = merged ? pa : bound;
				}
				bool walking = need && pa != pb && (pa < hi_a || pb < hi_a);
				bool on_a = pa < pb;
				wk.start(on_a ? pa : pb, W.root_base, W.unknown_id);
				lim_a = kNoLimit;
				if (has_bounds && walking) lim_a = tile_a + find_limit(bitmap, wk.cp - tile_a, stop);
				while (__builtin_amdgcn_ballot_w64(walking)) {
					bool fin = false;
					if (walking) fin = walk_step(W, wk, lim_a);
					log.put(fin, on_a ? wk.lid : (wk.lid | Q::neg));
					if (fin) {
						pa = on_a ? wk.cp : pa;
						pb = on_a ? pb : wk.cp;
						walking = pa != pb && (pa < hi_a || pb < hi_a);
						on_a = pa < pb;
						wk.cp = on_a ? pa : pb;
						wk.lend = wk.cp + 1;
						wk.lid = W.unknown_id;
						if (has_bounds && walking && wk.cp >= lim_a) lim_a = tile_a + find_limit(bitmap, wk.cp - tile_a, stop);
					}
					log.flush_if_full(Qd, lane);
				}
				if (need) {
					cur_entry = want;
					if (pa != pb) cur_exit = pa - hi_a; // never met: the exit moved
				}
			}

			PROF_ADD(5, prof_t); // rest of phase 2: take-back, dual walk, round control
			// ---- carry into the next tile / chunk exit ----
			const unsigned long long ne = __ballot(nonempty);
			const uint32_t last = ne ? 63u - (uint32_t)__builtin_clzll(ne) : 0u; // (no lane has bytes: a chunk whose begin was moved to the corpus end)
			chunk_exit = __shfl(cur_exit, last, 64);
			carry = chunk_exit;
			a0 += span_c;
		}
		if (lane == 0) {
			S.entry[j] = 0;
			S.exit_a[j] = chunk_exit;
		}
	}
	log.finish(Qd, lane);
#ifdef TOK_PROFILE
	PROF_ADD(0, prof_t);
	if (lane == 0) {
		unsigned long long *row = g_prof + (size_t)((blockIdx.x * 16 + wave) % 4096) * 8;
		for (int k = 0; k < 6; k++) row[k] = prof_acc[k];
		row[6] = prof_t - prof_start;
		row[7] = prof_acc[7];
	}
#endif
}

// ---- token stream -> counts ------------------------------------------------------------
// grid = (slab_blocks, id_ranges); each workgroup keeps bins for its id range in LDS, consumes
// its share of the stream (16-byte vectors, null entries skipped), then stores the bins to its
// slab row as int32.
//
// 16-bit entries (V + 3 <= 32,767 ids): one int32 bin per id, one id range.
// 32-bit entries: every id range is another pass over the whole stream (4 B per token: at
// V = 131 k four passes with int32 bins were 24 % of an evaluation), so the bins are BYTES, four
// to a word, 136 k ids per pass.  A bin is moved by returned 32-bit LDS atomics of +-(1 << 8j);
// the word is exact modulo 2^32 whatever the order, so what has to be kept beside it is how
// often each byte wrapped.  The operation whose returned value shows byte j at 0xFF before a +1
// (0x00 before a -1) is the one that wrapped it -- each wrap is seen by exactly one operation,
// atomics on a word being totally ordered -- and it notes the wrap (id, direction, how many
// bytes up the word the carry ran) in a list in LDS.  Booking a wrap: [id_j] += +-256 and, since
// the carry went into byte j+1, [id_j+1] -= +-1 -- in the workgroup's slab row when the list is
// worked off at the end, in the output vector itself at once if the list is full.  At the end byte j holds
// bias + count_j - 256 * wraps_j + wraps_(j-1), which is what the slab row plus the bookings add
// up to.  Bins start near the end they move away from (0x10 counting up, 0xF0 for the
// subtracting pass of a re-score; the few take-back entries move them the other way), so an id
// needs 240 entries in one workgroup before its first wrap.
constexpr uint32_t kWrapListBytes = 24 * 1024;
__device__ __forceinline__ uint32_t wrapped_bytes(uint32_t old, uint32_t j, bool down) {
	const uint32_t t = old >> (8 * j); // byte j and those above it; zeros shifted in
	const uint32_t edge_run = down ? (t ? (uint32_t)__builtin_ctz(t) >> 3 : 4u) : (~t ? (uint32_t)__builtin_ctz(~t) >> 3 : 4u);
	return edge_run < 4u - j ? edge_run : 4u - j;
}
template <typename T>
__device__ __forceinline__ void book_wraps(uint32_t note, uint32_t lo, uint32_t hi, T *counts) {
	// note = id << 3 | (bytes wrapped - 1) << 1 | down
	const long long step = (note & 1u) ? -256 : 256, carry = (note & 1u) ? 1 : -1;
	const uint32_t id = note >> 3, n = ((note >> 1) & 3u) + 1;
	const uint32_t j = (id - lo) & 3u;
	for (uint32_t t = 0; t < n && id + t < hi; t++) {
		atomicAdd(&counts[id + t], (T)step);
		if (j + t < 3 && id + t + 1 < hi) atomicAdd(&counts[id + t + 1], (T)carry);
	}
}
__device__ __noinline__ void book_wraps_now(uint32_t note, uint32_t lo, uint32_t hi, unsigned long long *counts) {
	book_wraps(note, lo, hi, counts);
}

template <bool WIDE, bool ONE_RANGE>
__global__ void __launch_bounds__(1024) hist_kernel(StreamDev Qd, uint32_t n_ids, uint32_t range_ids, uint32_t *slab,
                                                    unsigned long long *counts, uint32_t flip) {
	using Q = QTraits<WIDE>;
	extern __shared__ __attribute__((aligned(16))) unsigned char tile_hist_smem[];
	uint32_t *bins = reinterpret_cast<uint32_t *>(tile_hist_smem);
	__shared__ int s_total;
	const uint32_t lo = blockIdx.y * range_ids;
	const uint32_t hi = lo + range_ids < n_ids ? lo + range_ids : n_ids;
	const uint32_t width = hi - lo;
	const uint32_t n_words = WIDE ? (width + 3) / 4 : width;
	const uint32_t bias = flip ? 0xF0u : 0x10u;
	for (uint32_t i = threadIdx.x; i < n_words + (WIDE ? 64u : 0u); i += blockDim.x) bins[i] = WIDE ? bias * 0x01010101u : 0u;
	// (byte bins) the wrap list follows the bins and the lanes' spare words
	__shared__ uint32_t s_wraps;
	uint32_t *wrap_list = bins + n_words + 64;
	constexpr uint32_t wrap_cap = kWrapListBytes / 4;
	if (threadIdx.x == 0) s_total = 0, s_wraps = 0;
	__syncthreads();
	uint32_t batches = Qd.cursor[0];
	if (batches > Qd.capacity_batches) batches = Qd.capacity_batches;
	// The stream is read in rows of 64 vectors (1 KiB, one vector per lane); a block is one or
	// two rows.  A wave takes every n-th batch: one look at fill[batch] (blocks past it were
	// handed out but never written), then its rows kRowsInFlight at a time (16-bit entries: the loop
	// is bound by the latency of these loads, not by the LDS atomics -- without them: - 5 %).
	constexpr uint32_t vecs_per_block = 64 * kLogDepth * sizeof(typename Q::entry_t) / 16;
	constexpr uint32_t rows_per_block = vecs_per_block / 64;
	static_assert(vecs_per_block % 64 == 0, "a stream block is a whole number of 1 KiB rows");
	constexpr int kRowsInFlight = 8;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t n_waves = gridDim.x * (blockDim.x / 64);
	const uint32_t wave_id = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6));
	const uint4 *vec = reinterpret_cast<const uint4 *>(Qd.blocks);
	int total = 0; // +1 / -1 per entry seen, whatever its id range
	if constexpr (WIDE) {
		// Sixteen entries (four rows) at a time.  All their returned atomics go out before any
		// result is looked at, and looking is branch-free (one bit per entry that wrapped its byte;
		// one branch for the lot): a branch per entry, or per eight, cost more than the rest of the
		// pass, which is bound by the instructions it issues and by the loads it keeps in flight
		// (eight rows per wave; prefetching the next rows while counting lost: more registers
		// than a wave of sixteen has).  With s = +1 / -1 the add is s << 8j at word id >> 2, and the
		// byte wrapped iff old byte + s leaves 0..255, i.e. has bit 8 set.  A null entry or one
		// outside the id range adds 0 to a spare word of the lane's own, which therefore stays at
		// the bias and never looks wrapped.  (Variants measured on 157 M entries, this one 0.41 ms:
		// no test for null, the word index clamped by min() instead, 0.45; two or three chunks of
		// four rows refilled as they are counted, 0.46 - 0.53.  Plain adds without any wrap
		// handling, which is not exact: 0.22.)
		constexpr int kEntries = 16;
		const uint32_t flip_bit = flip ? 0x80000000u : 0u;
		for (uint32_t bt = wave_id; bt < batches; bt += n_waves) {
			uint32_t live = Qd.fill[bt];
			if (live > kStreamBatch) live = kStreamBatch;
			const uint32_t live_rows = __builtin_amdgcn_readfirstlane(live * rows_per_block);
			const uint4 *rows = vec + (uint64_t)bt * kStreamBatch * vecs_per_block + lane;
			for (uint32_t r0 = 0; r0 < live_rows; r0 += kRowsInFlight) {
				uint4 v[kRowsInFlight];
#pragma unroll
				for (int u = 0; u < kRowsInFlight; u++) {
					v[u] = uint4{Q::null_word, Q::null_word, Q::null_word, Q::null_word};
					if (r0 + u < live_rows) v[u] = rows[(size_t)(r0 + u) * 64];
				}
#pragma unroll
				for (int half = 0; half < 4 * kRowsInFlight / kEntries; half++) {
					uint32_t was[kEntries];
#pragma unroll
					for (int q = 0; q < kEntries; q++) {
						const uint4 &vv = v[half * (kEntries / 4) + (q >> 2)];
						const uint32_t e = (q & 3) == 0 ? vv.x : (q & 3) == 1 ? vv.y : (q & 3) == 2 ? vv.z : vv.w;
						const int s_any = ((int)(e ^ flip_bit) >> 31) | 1;
						const uint32_t b = (e & 0x7FFFFFFFu) - lo;
						const bool mine = e != Q::null_word && (ONE_RANGE || b < width);
						total += e != Q::null_word ? s_any : 0;
						const int s = mine ? s_any : 0;
						was[q] = atomicAdd(&bins[mine ? b >> 2 : n_words + lane], (uint32_t)s << (8 * (b & 3u)));
					}
					uint32_t wrapped = 0;
#pragma unroll
					for (int q = 0; q < kEntries; q++) {
						const uint4 &vv = v[half * (kEntries / 4) + (q >> 2)];
						const uint32_t e = (q & 3) == 0 ? vv.x : (q & 3) == 1 ? vv.y : (q & 3) == 2 ? vv.z : vv.w;
						const int s = ((int)(e ^ flip_bit) >> 31) | 1;
						const uint32_t j = (e - lo) & 3u;
						wrapped |= ((((was[q] >> (8 * j)) & 0xFFu) + (uint32_t)s) >> 8 & 1u) << q;
					}
					if (__builtin_expect(wrapped != 0, 0)) {
#pragma unroll
						for (int q = 0; q < kEntries; q++) {
							if (!((wrapped >> q) & 1u)) continue;
							const uint4 &vv = v[half * (kEntries / 4) + (q >> 2)];
							const uint32_t e = (q & 3) == 0 ? vv.x : (q & 3) == 1 ? vv.y : (q & 3) == 2 ? vv.z : vv.w;
							const uint32_t id = e & 0x7FFFFFFFu, j = (id - lo) & 3u;
							const bool down = ((e ^ flip_bit) >> 31) != 0;
							const uint32_t note = id << 3 | (wrapped_bytes(was[q], j, down) - 1) << 1 | (down ? 1u : 0u);
							const uint32_t at = atomicAdd(&s_wraps, 1u);
							if (at < wrap_cap) wrap_list[at] = note;
							else book_wraps_now(note, lo, hi, counts);
						}
					}
				}
			}
		}
	} else {
		for (uint32_t bt = wave_id; bt < batches; bt += n_waves) {
			uint32_t live = Qd.fill[bt];
			if (live > kStreamBatch) live = kStreamBatch;
			const uint32_t live_rows = __builtin_amdgcn_readfirstlane(live * rows_per_block);
			const uint4 *rows = vec + (uint64_t)bt * kStreamBatch * vecs_per_block + lane;
			for (uint32_t r0 = 0; r0 < live_rows; r0 += kRowsInFlight) {
				uint4 v[kRowsInFlight];
#pragma unroll
				for (int u = 0; u < kRowsInFlight; u++) {
					v[u] = uint4{Q::null_word, Q::null_word, Q::null_word, Q::null_word};
					if (r0 + u < live_rows) v[u] = rows[(size_t)(r0 + u) * 64];
				}
#pragma unroll
				for (int u = 0; u < kRowsInFlight; u++) {
					const uint32_t wds[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
					for (int k = 0; k < 4; k++) {
						if (wds[k] == Q::null_word) continue;
#pragma unroll
						for (int h = 0; h < 2; h++) {
							const uint32_t e = (wds[k] >> (16 * h)) & 0xFFFFu;
							const uint32_t id = e & 0x7FFFu;
							if (id == Q::null_entry) continue;
							const int sgn = (((e & Q::neg) != 0) != (flip != 0)) ? -1 : 1;
							total += sgn;
							if (id >= lo && id < hi) atomicAdd(&bins[id - lo], (uint32_t)sgn);
						}
					}
				}
			}
		}
	}
	// the row's extra column n_ids: this workgroup's token total (main.cpp:41), from the
	// first id range only so that it is counted once
	if (blockIdx.y == 0) {
		for (int o = 32; o > 0; o >>= 1) total += __shfl_down(total, o, 64);
		if ((threadIdx.x & 63) == 0 && total != 0) atomicAdd(&s_total, total);
	}
	__syncthreads();
	uint32_t *row = slab + (size_t)blockIdx.x * (n_ids + 1);
	for (uint32_t i = threadIdx.x; i < width; i += blockDim.x)
		row[lo + i] = WIDE ? (uint32_t)((int)((bins[i >> 2] >> (8 * (i & 3))) & 0xFFu) - (int)bias) : bins[i];
	if (WIDE) {
		// the noted wraps go into the workgroup's own slab row (once its stores have landed): the
		// hot ids are neighbours, and 256 workgroups booking them in the output vector queued up on
		// two or three cache lines (+ 0.4 ms per 157 M entries)
		__threadfence();
		__syncthreads();
		const uint32_t noted = s_wraps < wrap_cap ? s_wraps : wrap_cap;
		for (uint32_t i = threadIdx.x; i < noted; i += blockDim.x) book_wraps(wrap_list[i], lo, hi, row);
	}
	if (blockIdx.y == 0 && threadIdx.x == 0) row[n_ids] = (uint32_t)s_total;
}

// counts[id] += sum over slab rows (int32 partials, sign-extended); column n_ids of a row is
// its token total and lands in counts[n_ids] = total_tokens.  grid = (columns / 256, row
// groups): each thread sums every kReduceGroups-th row and adds its partial with one u64
// atomic, so the boundary fix-up (which also adds to counts) may run beside this kernel.
constexpr uint32_t kReduceGroups = 16; // 4 / 8 / 16 / 32: 20 / 12 / 10 / 11 us on 256 rows of 32 k columns
__global__ void __launch_bounds__(256) slab_reduce_kernel(const uint32_t *slab, uint32_t rows, uint32_t n_cols,
                                                          unsigned long long *counts) {
	const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
	if (id >= n_cols) return;
	long long sum = 0;
	for (uint32_t r = blockIdx.y; r < rows; r += kReduceGroups) sum += (long long)(int32_t)slab[(size_t)r * n_cols + id];
	if (sum != 0) atomicAdd(&counts[id], (unsigned long long)sum);
}

__global__ void __launch_bounds__(256) prepare_kernel(unsigned long long *out, uint32_t V, unsigned long long D,
                                                      unsigned long long total_bytes, uint32_t *cursor, uint32_t *fill,
                                                      uint32_t n_fill, uint32_t first_batches) {
	const uint32_t n = gridDim.x * blockDim.x;
	const uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x;
	if (out)
		for (uint32_t i = i0; i < V + 5; i += n) out[i] = i < 2 ? D : (i == V + 4 ? total_bytes : 0ull);
	for (uint32_t i = i0; i < kCursorWords; i += n) cursor[i] = i == 0 ? first_batches : 0u; // [0]: batches handed out
	for (uint32_t i = i0; i < n_fill; i += n) fill[i] = 0;
}

} // namespace

// ---- host side -----------------------------------------------------------------------------

uint32_t stream_block_bytes(uint32_t wide_ids) { return 64 * kLogDepth * (wide_ids ? 4 : 2); }

bool plan_tile_scan(const TrieDev &T, uint64_t max_len, uint32_t V, uint32_t num_cus, TileScanPlan *plan) {
	static const uint32_t kSeg[] = {36, 52, 68, 84, 92, 100, 116, 132, 196, 260, 516, 1028};
	uint32_t seg = 0;
	uint32_t want_seg = 100;
	if (const char *s = getenv("TOKSCORE_SEG")) want_seg = (uint32_t)atoi(s);
	for (uint32_t s : kSeg) {
		if (s >= max_len + 4 && s >= want_seg) {
			seg = s;
			break;
		}
	}
	if (!seg) return false; // tokens longer than a lane segment: generic scan
	uint32_t waves = 16;
	if (const char *s = getenv("TOKSCORE_WAVES")) waves = (uint32_t)atoi(s);
	if (waves < 1) waves = 1;
	if (waves > 16) waves = 16;
	const uint32_t wide = (V + 2 >= 0x7FFFu) ? 1u : 0u;
	const uint32_t halo = (uint32_t)((max_len + 15) & ~15ull);
	const uint32_t lds_total = 160 * 1024;
	uint32_t wave_bytes = 0;
	for (;;) {
		const uint32_t tile_cap = 64 * seg + halo + 16;
		const uint32_t n_vec = (tile_cap + 15) / 16;
		const uint32_t bm_words = (tile_cap + 31) / 32 + 1;
		wave_bytes = n_vec * 16 + ((bm_words * 4 + 15) & ~15u) + stream_block_bytes(wide) + 64 * kPendDepth * (wide ? 4 : 2);
		// (how much of the trie sits in LDS matters little: some lane of 64 misses anyway)
		if ((uint64_t)wave_bytes * waves + 2 * 1024 <= lds_total) break;
		if (waves == 1) return false;
		waves -= waves > 8 ? 2 : 1; // long segments / wide ids: fewer waves so that the tiles still fit
	}
	uint32_t lds_budget = lds_total;
	if (const char *s = getenv("TOKSCORE_BLOCKS_PER_CU")) lds_budget = lds_total / std::max(1, atoi(s)) - 512;
	if (wave_bytes * waves + 64 > lds_budget) return false;
	uint32_t lds_slots = (lds_budget - wave_bytes * waves) / 8;
	if (const char *s = getenv("TOKSCORE_LDS_SLOTS")) lds_slots = std::min<uint32_t>(lds_slots, (uint32_t)atoi(s));
	if (lds_slots > T.nslots) lds_slots = T.nslots;
	if (lds_slots < 1) lds_slots = 1;
	plan->waves_per_block = waves;
	plan->seg_bytes = seg;
	plan->lds_slots = lds_slots;
	plan->lds_bytes = lds_slots * 8 + wave_bytes * waves;
	uint32_t per_cu = 1;
	if (const char *s = getenv("TOKSCORE_BLOCKS_PER_CU")) per_cu = std::max(1, atoi(s));
	plan->blocks = num_cus * per_cu;
	if (const char *e = getenv("TOKSCORE_TILE_BLOCKS")) plan->blocks = (uint32_t)std::max(1, atoi(e)); // tests: a small grid
	plan->wide_ids = wide;
	plan->hist_range = 0;
	if (const char *e = getenv("TOKSCORE_HIST_RANGE")) plan->hist_range = (uint32_t)std::max(64, atoi(e)); // tests: several id ranges
	return true;
}

void launch_scan_tile(const TrieDev &T, const CorpusDev &C, const ChunkState &S, const StreamDev &Q,
                      const TileScanPlan &plan, const uint64_t *chunk_first_doc, uint64_t max_len,
                      hipStream_t stream, const uint32_t *sel_list, const uint32_t *sel_count) {
	const uint32_t halo = (uint32_t)((max_len + 15) & ~15ull);
	const uint32_t wave_bytes = (plan.lds_bytes - plan.lds_slots * 8) / plan.waves_per_block;
	const uint64_t readable_end = (C.total + 256) & ~15ull;
	TrieDev Tl = T;
	Tl.lds_slots = plan.lds_slots;
	if (plan.wide_ids) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_tile_kernel<true>),
		                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes);
		hipLaunchKernelGGL(scan_tile_kernel<true>, dim3(plan.blocks), dim3(64 * plan.waves_per_block), plan.lds_bytes,
		                   stream, Tl, C, S, Q, chunk_first_doc, plan.seg_bytes, halo, (uint32_t)max_len, wave_bytes,
		                   readable_end, sel_list, sel_count);
	} else {
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_tile_kernel<false>),
		                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes);
		hipLaunchKernelGGL(scan_tile_kernel<false>, dim3(plan.blocks), dim3(64 * plan.waves_per_block), plan.lds_bytes,
		                   stream, Tl, C, S, Q, chunk_first_doc, plan.seg_bytes, halo, (uint32_t)max_len, wave_bytes,
		                   readable_end, sel_list, sel_count);
	}
}

void launch_prepare(unsigned long long *out, uint32_t V, uint64_t D, uint64_t total_bytes, const StreamDev &Q,
                    hipStream_t stream, uint32_t first_batches) {
	hipLaunchKernelGGL(prepare_kernel, dim3(128), dim3(256), 0, stream, out, V, (unsigned long long)D,
	                   (unsigned long long)total_bytes, Q.cursor, Q.fill, Q.capacity_batches, first_batches);
}

void launch_slab_reduce(const uint32_t *slab, uint32_t rows, uint32_t n_cols, unsigned long long *counts,
                        hipStream_t stream) {
	hipLaunchKernelGGL(slab_reduce_kernel, dim3((n_cols + 255) / 256, kReduceGroups), dim3(256), 0, stream, slab, rows,
	                   n_cols, counts);
}

void launch_histogram(const StreamDev &Q, const TileScanPlan &plan, uint32_t V, unsigned long long *counts,
                      uint32_t *slab, uint32_t slab_blocks, hipStream_t stream, bool flip) {
	const uint32_t n_ids = V + 3;
	// every id range is a pass over the whole stream: as few as the 160 KB of LDS allow, equal in
	// size (int32 bins for 16-bit entries: always one range; byte bins for 32-bit entries)
	uint32_t max_range = plan.wide_ids ? 136000u : 40000u;
	if (plan.hist_range && plan.hist_range < max_range) max_range = plan.hist_range;
	const uint32_t ranges = (n_ids + max_range - 1) / max_range;
	const uint32_t range = (((n_ids + ranges - 1) / ranges) + 3) & ~3u; // whole words of byte bins
	auto launch = [&](auto kernel, uint32_t lds) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		hipLaunchKernelGGL(kernel, dim3(slab_blocks, ranges), dim3(1024), lds, stream, Q, n_ids, range, slab, counts,
		                   flip ? 1u : 0u);
	};
	if (plan.wide_ids) {
		const uint32_t lds = range + 256 + kWrapListBytes; // + a spare word per lane + the wrap list
		if (ranges == 1) launch(&hist_kernel<true, true>, lds);
		else launch(&hist_kernel<true, false>, lds);
	} else {
		launch(&hist_kernel<false, false>, range * 4);
	}
	launch_slab_reduce(slab, slab_blocks, n_ids + 1, counts, stream);
}

} // namespace tokscore

#ifdef TOK_PROFILE
extern "C" int tokscore_debug_prof(unsigned long long *out, size_t n) {
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tokscore::g_prof), n * sizeof(unsigned long long));
}
#endif
