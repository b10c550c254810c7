// tokscore.hip -- C-ABI of libtokscore.so (include/tokscore.h): context, HBM residency,
// chunk planning and kernel sequencing.  There is no CPU scoring path in this library:
// without a HIP device every scoring entry point fails with TOKSCORE_E_NO_DEVICE.
#include "../../include/tokscore.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <map>
#include <string>
#include <vector>

#include "dfa.h"
#include "kernels.h"
#include "trie.h"

using namespace tokscore;

namespace {

constexpr uint64_t kCorpusPad = 4096;         // readable zero bytes past the corpus end (tile halo)
constexpr uint64_t kSamplePieces = 16;
constexpr size_t kMaxPatch = 8;               // toggles an incremental re-score takes in one go

uint64_t env_u64_early(const char *name, uint64_t dflt) {
	const char *s = getenv(name);
	if (!s || !*s) return dflt;
	return strtoull(s, nullptr, 10);
}

uint64_t env_u64(const char *name, uint64_t dflt) {
	const char *s = getenv(name);
	if (!s || !*s) return dflt;
	return strtoull(s, nullptr, 10);
}

const uint64_t kSampleBytes = std::max<uint64_t>(1, env_u64_early("TOKSCORE_SAMPLE_MB", 2)) << 20; // host sample for the hot-first layouts

// interval between two recorded events; 0 when both are the same event
hipError_t elapsed_ms(float *ms, hipEvent_t a, hipEvent_t b) {
	if (a == b) {
		*ms = 0.0f;
		return hipSuccess;
	}
	return hipEventElapsedTime(ms, a, b);
}

// ---- scratch of tokscore_count_occurrences_static, the one entry point that allocates nothing.
// Each GPU the library runs on holds its own copy of the array until the process ends.  In host
// code the array's name is only a stand-in: the device address of the current GPU's copy comes
// from hipGetSymbolAddress (once per device, in tokscore_create) and is what every kernel
// argument and memset below uses.
constexpr size_t kStaticScratchBytes = (size_t)16 << 20;
constexpr int kStaticScratchDevices = 64;
static __device__ __align__(256) unsigned char g_static_scratch[kStaticScratchBytes];
std::mutex g_static_scratch_mu;
unsigned char *g_static_scratch_ptr[kStaticScratchDevices]; // by HIP device ordinal

// `device` is the current device
hipError_t static_scratch_address(int device, unsigned char **ptr) {
	*ptr = nullptr;
	if (device < 0 || device >= kStaticScratchDevices) return hipErrorInvalidDevice;
	std::lock_guard<std::mutex> lock(g_static_scratch_mu);
	if (!g_static_scratch_ptr[device]) {
		void *p = nullptr;
		const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_static_scratch));
		if (e != hipSuccess) return e;
		if (!p || ((uintptr_t)p & 255)) return hipErrorInvalidSymbol;
		g_static_scratch_ptr[device] = (unsigned char *)p;
	}
	*ptr = g_static_scratch_ptr[device];
	return hipSuccess;
}

} // namespace

struct tokscore_ctx {
	bool hostonly = false;
	int device = -1;
	unsigned char *static_scratch = nullptr; // this device's copy of g_static_scratch
	int num_cus = 256;
	hipStream_t stream = nullptr;
	hipStream_t side = nullptr;      // boundary fix-up runs here, beside the histogram
	hipEvent_t ev_fork = nullptr, ev_join = nullptr;
	// ring of event sets, one per score call, so a caller can time many calls back to back and
	// read the kernel durations afterwards without a host sync inside its timed region
	static constexpr int kRing = 256;
	hipEvent_t ring[kRing][5] = {};
	// the events the timing code reads for a call: a full evaluation records three of its five
	// (every record costs about 5 us of stream time) and reads 0 = 1, 3 = 2
	hipEvent_t view[kRing][5] = {};
	bool timing_on = true;      // TOKSCORE_TIMING=0: no events at all
	// tokscore_set_timing: events around the scan kernel only, on every timing_every-th full
	// evaluation (an event record holds the stream for 3 - 4 us: three per evaluation are 8 % of
	// one on the bench shard)
	bool timing_scan_only = false;
	uint32_t timing_every = 1;
	uint64_t timing_skipped = 0;
	bool timing_detail = false; // TOKSCORE_TIMING=2: the incremental re-score records its passes too
	hipEvent_t *ev = ring[0];
	uint64_t calls = 0;   // score calls enqueued
	uint64_t drained = 0; // calls whose timings were handed out by tokscore_drain_timings
	std::string err;

	// vocabulary
	VocabTrie trie;
	PackedTrie packed;
	bool vocab_set = false;
	bool packed_valid = false;  // host image matches trie + hotness
	bool device_trie_valid = false;
	uint64_t *d_slots = nullptr;
	uint64_t d_slots_cap = 0;
	uint64_t V = 0;
	// the token strings (incremental re-score looks for the toggled tokens' occurrences)
	std::vector<uint8_t> h_vocab_bytes;
	std::vector<uint64_t> h_vocab_off;
	uint8_t *d_vocab_bytes = nullptr;
	bool d_vocab_valid = false;
	// toggles since the device trie was last brought up to date: (token, slot whose id changed)
	std::vector<std::pair<uint64_t, int64_t>> pending;
	// d_out holds the result of the last synchronous evaluation of the resident corpus under
	// the device trie's state (what an incremental re-score starts from)
	bool state_valid = false;
	// the last use of the token stream's cursor words was a full evaluation over ranges, whose
	// reduction left them as the next one needs them (stream_ready = its number of waves): that
	// evaluation then has no prepare launch.  Every other user of the words clears this.
	uint32_t stream_ready = 0;
	uint32_t rescored_chunks = 0xFFFFFFFFu; // by the last incremental re-score (UINT32_MAX: full evaluation)

	// corpus
	bool corpus_set = false;
	bool own_corpus = false;
	uint8_t *d_bytes = nullptr;
	uint64_t *d_off = nullptr;
	uint64_t D = 0;
	uint64_t total = 0;
	std::vector<uint64_t> h_off;
	std::vector<uint8_t> sample;
	bool sample_used = false;

	// plan
	bool plan_valid = false;
	// what made the plan invalid (another corpus, another vocabulary, a repacked trie) also ends the automaton
	// scan's image and buffers, which live apart from the plan: the plan of the other scan paths is only built
	// when something needs it (0.55 s per GB), the automaton serves full evaluations without it
	bool dfa_invalid = false;
	int scan_path = 0; // 0 generic, 1 tile
	bool force_generic = false; // set after a token-stream overflow
	TileScanPlan tile = {};
	uint64_t chunk_bytes = 0;
	std::vector<uint64_t> h_chunk_begin;
	uint64_t *d_chunk_begin = nullptr;
	uint64_t *d_chunk_first_doc = nullptr;
	uint32_t *d_chunk_state = nullptr; // entry | exit_a | exit_b
	uint32_t nchunks = 0;
	uint32_t unsynced = 0; // chunks whose entry is speculated (not at a guaranteed token start)
	bool uniform_chunks = false; // boundaries at multiples of chunk_bytes, each moved forward by < chunk_bytes
	uint32_t *d_sel = nullptr;   // chunk bitmap of an incremental re-score
	uint32_t *d_sel_list = nullptr; // the selected chunks as a list [nchunks] + 1 word: their number
	// tokscore_score_toggles scratch, grown on demand: the batch description and results | the
	// (move, chunk) pairs | the (move, stream) items
	unsigned char *d_tg_meta = nullptr;
	size_t tg_meta_cap = 0;
	uint4 *d_tg_pairs = nullptr;
	size_t tg_pairs_cap = 0;
	ToggleItem *d_tg_items = nullptr;
	size_t tg_items_cap = 0;
	unsigned char *h_tg_meta = nullptr; // pinned: the batch description on its way up, the results on their way down
	size_t h_tg_meta_cap = 0;
	uint32_t *d_tg_seen = nullptr; // one bit per stream: listed already (incremental re-score)
	size_t tg_seen_cap = 0;
	// row f1 on the stream image: the prepared corpus without lower-casing, transposed like d_tprep
	// (built by the first tokscore_count_occurrences of a plan), per-slot totals
	uint8_t *d_tprep_raw = nullptr;
	unsigned long long *d_slot_ends = nullptr; // [nslots + 2]: ends per slot | entries | positions
	size_t slot_ends_cap = 0;
	uint32_t *d_stream_order = nullptr;  // [nstreams] streams by length, longest first (plan_stream)
	unsigned long long *d_tok_starts = nullptr; // one bit per prepared byte: a token starts there; same validity as d_stream_tokens
	size_t tok_starts_cap = 0;
	uint32_t *d_stream_tokens = nullptr; // [nstreams] tokens per stream, valid while stream_tokens_epoch == trie_epoch
	size_t stream_tokens_cap = 0;
	uint64_t trie_epoch = 1, stream_tokens_epoch = 0; // trie_epoch: another one whenever the device trie or the plan changes
	uint64_t entries_epoch = 0; // the trie_epoch whose full evaluation over ranges (on the context's own stream) left d_range_state's entries exact
	// knobs of the move paths, read once at tokscore_create (not on every call)
	bool knob_noincr = false, knob_nobatch = false;
	uint64_t knob_incr_div = 0;      // 0: the cost estimate decides
	bool knob_no_mark_ranges = false; // TOKSCORE_NOMARKRANGES=1: always count_streams_kernel before batched moves (tests)
	uint64_t knob_toggle_items = 32ull << 20; // (move, chunk) pairs one batch of moves may have
	uint64_t tg_list_hint = 8192;             // entries per item list provided first (doubled when a batch needed more)
	uint64_t toggle_items = 0; // (move, stream) items walked by score_toggles so far (diagnostics)
	// ... by one pass of a search automaton over the corpus (kernels_dfa.hip: dfa_search_kernel) instead of the occurrence
	// index and a search per (move, chunk): the (move, stream) table, the stream of every 256-byte block of the prepared corpus
	bool knob_search = true;        // TOKSCORE_SEARCH=0: the index and per-chunk searches
	uint64_t knob_search_moves = 256; // TOKSCORE_SEARCH_MOVES: moves per pass
	uint32_t knob_search_whole = kToggleOffsets; // TOKSCORE_SEARCH_WHOLE: a stream with more occurrences of a move's string than this is walked whole
	unsigned char *d_sr_table = nullptr; // keys u64[slots] | count u32[slots] | items ToggleItem[slots]
	size_t sr_table_bytes = 0;
	uint64_t sr_slots_hint = 1u << 16;
	bool sr_slots_fixed = false;   // TOKSCORE_SEARCH_SLOTS given: the first table exactly that (tests)
	uint32_t *d_block_stream = nullptr; // per stream plan
	uint64_t search_batches = 0;    // batches judged this way (diagnostics)
	// occurrence index (one bit per string and chunk), built by the first incremental re-score of a plan
	uint32_t *d_index = nullptr;
	int index_state = 0; // 0 not tried, 1 built, -1 not available (too large / allocation failed)
	std::vector<int32_t> tok_row;        // token -> index row
	std::vector<uint32_t> row_chunks;    // row -> number of chunks the string occurs in
	void *d_stream = nullptr;
	uint32_t stream_cap_batches = 0;
	uint32_t *d_fill = nullptr;
	uint32_t *d_cursor = nullptr;
	uint32_t *d_slab = nullptr;
	uint32_t slab_blocks = 0;

	// stream scan (kernels_stream.hip): trie image, prepared corpus, stream boundaries
	StreamImage simg;
	bool simg_valid = false;         // host image matches the trie + hotness
	uint8_t *d_slt = nullptr;
	uint16_t *d_sbase = nullptr;
	uint32_t *d_slot_id = nullptr;
	std::vector<int64_t> pending_s;  // stream-image slots of the pending toggles
	uint64_t d_simg_cap = 0;
	uint8_t *d_prep = nullptr;       // codes of the lower-cased documents, code 0 after each (per corpus + vocabulary)
	uint64_t prep_total = 0;         // total + D
	bool prep_ok = false;            // built, and no document holds a NUL byte
	StreamScanPlan splan = {};
	uint32_t *d_pbeg = nullptr;      // [nstreams + 1] prepared offsets
	uint32_t *d_chunk_first = nullptr; // [nchunks + 1] first stream of every chunk
	bool chunks_synced = false;      // every chunk begins at a kept cut (what an incremental re-score needs)
	uint32_t nstreams = 0;
	uint32_t streams_per_chunk = 1;
	uint32_t widest_chunk = 1; // most streams any chunk has
	uint32_t range_bytes = 0;        // full evaluation: equal ranges of the prepared corpus, one per lane
	uint32_t nranges = 0;
	uint32_t *d_range_state = nullptr; // entry | exit_a | exit_b, nranges words each
	uint32_t fix_epoch = 0;          // tag of this evaluation's exit_b entries
	bool fix_epoch_wrapped = false;  // (set after the first evaluation: tag 1 after that means the tags have gone round)
	uint8_t *d_tprep = nullptr;      // the ranges transposed, 64 to a block (kernels_stream.hip: scan_ranges_kernel)
	uint32_t halo = 0;
	StreamScanPlan rplan = {};       // launch of scan_ranges_kernel
	uint8_t *h_patch_lt = nullptr;    // pinned, beside h_patch
	uint32_t *h_patch_id = nullptr;

	// automaton scan (kernels_dfa.hip, dfa.h): serves full evaluations while the image matches the
	// vocabulary; built with the plan, rebuilt after toggles once two full evaluations in a row found it stale
	DfaImage dimg;
	int dfa_state = 0;             // 0 not tried for this plan, 1 ready, -1 not available
	uint64_t toggle_version = 0;   // counts tokscore_toggle calls that changed something
	uint64_t dfa_version = 0;      // toggle_version the device image was built from
	uint32_t dfa_stale_evals = 0;
	bool dfa_enabled = true;       // TOKSCORE_DFA=0 switches it off
	uint64_t dfa_max_slots = 6u << 20;
	uint32_t dfa_chains = 0;
	uint32_t dfa_lds_cap = 0;
	bool dfa_relayout_on = true;   // TOKSCORE_DFA_RELAYOUT=0: what LDS holds stays as the host's sample chose it
	uint32_t dfa_relayouts_max = 1; // TOKSCORE_DFA_RELAYOUTS
	// the automaton serves a vocabulary while the sample's lookups beyond LDS and its arrivals on virtual entries stay
	// below these shares (parts per million; TOKSCORE_DFA_MAX_COLD_PPM, TOKSCORE_DFA_MAX_VIRTUAL_PPM)
	uint64_t dfa_max_cold_ppm = 10000, dfa_max_virtual_ppm = 30000;
	// the failure form (dfa.h: DfaImage::fail_code; kernels_dfa.hip: dfa_fscan_kernel) for the vocabularies the policy
	// above turns away: taken while its own sample sends at most this share of the lookups beyond LDS
	// (TOKSCORE_DFA_FAIL_MAX_COLD_PPM).  TOKSCORE_DFA_FORM: 0 = as the samples say, 1 = flattened or nothing, 2 = the
	// failure form whatever the vocabulary.
	uint64_t dfa_fail_max_cold_ppm = 300000; // (measured: wiki_annealed 0.18 - 0.28 wins over the tile scan, random UTF-8 V = 128 k 0.38 loses)
	uint32_t dfa_form_knob = 0;
	uint32_t dfa_warm_bytes = 64;  // TOKSCORE_DFA_WARM: bytes in front of a range that its first entry state is worked out from (0: the separator's state)
	bool dfa_fail = false;         // the image is in the failure form
	DfaScanPlan dplan = {};
	uint32_t *d_dhot = nullptr, *d_dgof = nullptr, *d_dslot = nullptr, *d_dev_miss = nullptr, *d_dev_arr = nullptr, *d_dpool = nullptr;
	uint8_t *d_dcodes = nullptr;   // the prepared corpus in the automaton's codes, linear
	void *d_dtcodes = nullptr;     // ... transposed for the scan
	uint2 *d_dstate = nullptr;     // entry | exit_a | exit_b
	uint32_t *d_dtag = nullptr;
	uint32_t *d_dcold = nullptr;   // miss | arrival counters of the slots beyond the LDS prefix
	void *d_dstream_m = nullptr, *d_dstream_a = nullptr;
	uint32_t *d_dfill_m = nullptr, *d_dfill_a = nullptr;
	uint32_t *d_dcursor = nullptr; // [0..1] miss stream, [4..5] arrival stream, [8..9] fix-up words
	uint32_t dstream_cap = 0;
	uint32_t *d_dslab = nullptr;
	uint32_t dslab_blocks = 0;
	uint32_t dfa_epoch = 0;
	bool dfa_used_last = false;    // the last full evaluation ran on the automaton scan
	// what LDS holds is chosen again after a plan's first evaluations, from the scan's own counters
	std::map<uint32_t, uint32_t> dfa_prior; // uses of entries that LDS did not hold, by DfaImage::ent_key (DfaOptions::prior_*)
	std::vector<uint32_t> dfa_prior_keys, dfa_prior_counts; // ... flattened for the builder
	uint32_t dfa_evals = 0;          // evaluations on the image as it is
	uint32_t dfa_relayouts = 0;      // images built from counters (at most dfa_relayouts_max per vocabulary)
	uint64_t dfa_second_hits = 0;    // lookups the global array answered with a hit, in the evaluations counted last
	double dfa_build_ms = 0, dfa_plan_ms = 0;

	// outputs
	unsigned long long *d_out = nullptr;
	uint64_t d_out_cap = 0;
	// pinned: the result as it comes back (V+5 words + the stream cursor pair) and the slot
	// values of the pending toggles, so that neither copy stalls the host
	uint64_t *h_out = nullptr;
	uint64_t *h_patch = nullptr;
	CandidateArraysDev cands; // result of the last tokscore_extract_candidates, until copied out
	bool patch_in_flight = false; // h_patch may still be read by a copy nobody waited for
	float last_ms[4] = {0, 0, 0, 0};
	bool timing_pending = false;
};

namespace {

int fail(tokscore_ctx *c, int status, const std::string &msg) {
	if (c) c->err = msg;
	return status;
}

#define HIP_TRY(ctx, call)                                                                              \
	do {                                                                                                \
		hipError_t _e = (call);                                                                         \
		if (_e != hipSuccess)                                                                           \
			return fail(ctx, TOKSCORE_E_HIP, std::string(#call) + ": " + hipGetErrorString(_e));        \
	} while (0)

void free_corpus(tokscore_ctx *c) {
	free_candidates(&c->cands);
	if (c->own_corpus) {
		if (c->d_bytes) (void)hipFree(c->d_bytes);
		if (c->d_off) (void)hipFree(c->d_off);
	}
	c->d_bytes = nullptr;
	c->d_off = nullptr;
	if (c->d_prep) (void)hipFree(c->d_prep);
	c->d_prep = nullptr;
	c->prep_ok = false;
	c->prep_total = 0;
	c->own_corpus = false;
	c->corpus_set = false;
	c->plan_valid = false;
	c->dfa_invalid = true;
}

// The stream scan's copy of the corpus (kernels_stream.hip: prep_corpus_kernel), built once per
// corpus + vocabulary (with the plan): codes of the lower-cased bytes, code 0 after every document.
// prep_ok stays false when a document holds a NUL byte or the copy would not fit 32-bit offsets:
// the tile scan is used then.
int build_prepared(tokscore_ctx *c) {
	c->prep_ok = false;
	if (c->d_prep) (void)hipFree(c->d_prep);
	c->d_prep = nullptr;
	c->prep_total = c->total + c->D;
	if (c->D == 0 || c->prep_total + kCorpusPad >= 0xFFFF0000ull || env_u64("TOKSCORE_NOPREP", 0)) return TOKSCORE_OK;
	uint32_t *d_flags = nullptr;
	uint32_t flags = 1;
	hipError_t e = hipMalloc((void **)&c->d_prep, c->prep_total + kCorpusPad);
	if (e == hipSuccess) e = hipMalloc((void **)&d_flags, sizeof(uint32_t));
	if (e == hipSuccess) e = hipMemsetAsync(c->d_prep + (c->prep_total & ~15ull), 0, kCorpusPad + (c->prep_total & 15), c->stream);
	if (e == hipSuccess) e = hipMemsetAsync(d_flags, 0, sizeof(uint32_t), c->stream);
	if (e == hipSuccess) {
		CorpusDev C{c->d_bytes, c->d_off, c->D, c->total};
		ByteCodes M;
		memcpy(M.code, c->simg.code, 256);
		launch_prep_corpus(C, M, c->d_prep, c->prep_total, d_flags, c->stream);
		e = hipMemcpyAsync(&flags, d_flags, sizeof flags, hipMemcpyDeviceToHost, c->stream);
	}
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	if (d_flags) (void)hipFree(d_flags);
	if (e != hipSuccess) {
		(void)hipGetLastError();
		if (c->d_prep) (void)hipFree(c->d_prep);
		c->d_prep = nullptr;
		return TOKSCORE_OK; // not fatal: the tile scan needs no prepared copy
	}
	c->prep_ok = flags == 0;
	if (!c->prep_ok) {
		(void)hipFree(c->d_prep);
		c->d_prep = nullptr;
	}
	return TOKSCORE_OK;
}

// TOKSCORE_TRACE_HOST=1: host milliseconds between marks, on stderr (where a first evaluation goes)
struct HostTrace {
	bool on = getenv("TOKSCORE_TRACE_HOST") != nullptr;
	std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
	void mark(const char *what) {
		if (!on) return;
		const auto n = std::chrono::steady_clock::now();
		fprintf(stderr, "tokscore: %-36s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
		t = n;
	}
};

void free_dfa(tokscore_ctx *c) {
	void **bufs[] = {(void **)&c->d_dhot, (void **)&c->d_dgof, (void **)&c->d_dslot, (void **)&c->d_dev_miss, (void **)&c->d_dev_arr, (void **)&c->d_dpool, (void **)&c->d_dcodes,
	                 &c->d_dtcodes, (void **)&c->d_dstate, (void **)&c->d_dtag, (void **)&c->d_dcold, &c->d_dstream_m, &c->d_dstream_a,
	                 (void **)&c->d_dfill_m, (void **)&c->d_dfill_a, (void **)&c->d_dcursor, (void **)&c->d_dslab};
	for (void **b : bufs) {
		if (*b) (void)hipFree(*b);
		*b = nullptr;
	}
	c->dfa_state = 0;
	c->dfa_used_last = false;
	c->dfa_prior.clear();
	c->dfa_evals = c->dfa_relayouts = 0;
	c->dfa_second_hits = 0;
}

void free_plan(tokscore_ctx *c) {
	// nothing enqueued by an earlier asynchronous evaluation may still use these buffers
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	if (c->side) (void)hipStreamSynchronize(c->side);
	if (c->d_chunk_begin) (void)hipFree(c->d_chunk_begin);
	if (c->d_chunk_state) (void)hipFree(c->d_chunk_state);
	if (c->d_chunk_first_doc) (void)hipFree(c->d_chunk_first_doc);
	c->d_chunk_first_doc = nullptr;
	if (c->d_stream) (void)hipFree(c->d_stream);
	if (c->d_cursor) (void)hipFree(c->d_cursor);
	if (c->d_fill) (void)hipFree(c->d_fill);
	c->d_fill = nullptr;
	if (c->d_sel) (void)hipFree(c->d_sel);
	c->d_sel = nullptr;
	if (c->d_sel_list) (void)hipFree(c->d_sel_list);
	c->d_sel_list = nullptr;
	if (c->d_index) (void)hipFree(c->d_index);
	c->d_index = nullptr;
	c->index_state = 0;
	if (c->d_slab) (void)hipFree(c->d_slab);
	if (c->d_pbeg) (void)hipFree(c->d_pbeg);
	c->d_pbeg = nullptr;
	if (c->d_block_stream) (void)hipFree(c->d_block_stream);
	c->d_block_stream = nullptr;
	if (c->d_stream_order) (void)hipFree(c->d_stream_order);
	c->d_stream_order = nullptr;
	if (c->d_chunk_first) (void)hipFree(c->d_chunk_first);
	c->d_chunk_first = nullptr;
	if (c->d_range_state) (void)hipFree(c->d_range_state);
	c->d_range_state = nullptr;
	if (c->d_tprep) (void)hipFree(c->d_tprep);
	c->d_tprep = nullptr;
	if (c->d_tprep_raw) (void)hipFree(c->d_tprep_raw);
	c->d_tprep_raw = nullptr;
	c->d_chunk_begin = nullptr;
	c->d_chunk_state = nullptr;
	c->d_stream = nullptr;
	c->d_cursor = nullptr;
	c->d_slab = nullptr;
	c->plan_valid = false;
	c->trie_epoch++;
}

// Work chunks, the unit the scan's work queue hands out -- the same unit of independence the
// reference parallelises over (main.cpp:38-43), cut finer.  With `uniform` the cuts are at
// multiples of chunk_bytes (ensure_plan then moves each one forward to a guaranteed token
// start, so chunks stay equal in size and need no repair).  Otherwise a cut is snapped
// forward to a document start when one is within half a chunk (its entry is then certain);
// the rest are cut mid-text and repaired by the fix-up kernels.
void build_chunks(tokscore_ctx *c, uint64_t chunk_bytes, bool uniform) {
	std::vector<uint64_t> &b = c->h_chunk_begin;
	b.clear();
	b.push_back(0);
	const std::vector<uint64_t> &off = c->h_off;
	uint64_t pos = 0;
	while (pos + chunk_bytes + chunk_bytes / 2 < c->total) {
		const uint64_t target = pos + chunk_bytes;
		uint64_t next = target;
		if (!uniform) {
			auto it = std::lower_bound(off.begin(), off.end(), target);
			if (it != off.end() && *it - target <= chunk_bytes / 2 && *it < c->total) next = *it;
		}
		b.push_back(next);
		pos = next;
	}
	b.push_back(c->total);
}

TrieDev trie_dev(const tokscore_ctx *c) {
	TrieDev T;
	T.slots = reinterpret_cast<const uint2 *>(c->d_slots);
	T.nslots = (uint32_t)c->packed.slots.size();
	T.unknown_id = c->packed.unknown_id;
	T.lds_slots = c->tile.lds_slots;
	T.root_base = c->packed.slots.empty() ? 0 : PackedTrie::base(c->packed.slots[0]);
	return T;
}

StreamTrieDev stream_trie_dev(const tokscore_ctx *c) {
	StreamTrieDev T;
	T.lt = c->d_slt;
	T.base = c->d_sbase;
	T.nslots = (uint32_t)c->simg.lt.size();
	T.root_base = c->simg.root_base;
	return T;
}

StreamWork stream_work(const tokscore_ctx *c, bool selected) {
	StreamWork W;
	W.prep = c->d_prep;
	W.pbeg = c->d_pbeg;
	W.nstreams = c->nstreams;
	W.prep_total = (uint32_t)c->prep_total;
	W.range_bytes = c->range_bytes;
	W.nranges = c->nranges;
	W.entry = c->d_range_state;
	W.exit_a = c->d_range_state + c->nranges;
	W.tprep = c->d_tprep;
	W.halo = c->halo;
	W.round = c->rplan.round;
	W.sel_list = selected ? c->d_sel_list : nullptr;
	W.sel_count = selected ? c->d_sel_list + c->nchunks : nullptr;
	W.chunk_first = c->d_chunk_first;
	W.streams_per_chunk = c->widest_chunk; // (a chunk's first cut may have moved forward, a chunk without one joined the one before)
	return W;
}

int ensure_trie(tokscore_ctx *c) {
	if (!c->vocab_set) return fail(c, TOKSCORE_E_STATE, "vocabulary not set");
	HostTrace tr;
	if (!c->packed_valid || (!c->sample_used && !c->sample.empty())) {
		if (!c->sample.empty() && !c->sample_used) {
			c->trie.AddHotnessSample(c->sample.data(), c->sample.size());
			c->sample_used = true;
			tr.mark("ensure_trie: hotness sample");
		}
		std::string e;
		if (!c->trie.Pack(&c->packed, &e)) return fail(c, TOKSCORE_E_VOCAB, e);
		tr.mark("ensure_trie: double array packed");
		c->packed_valid = true;
		c->device_trie_valid = false;
		c->plan_valid = false;
		// (the automaton's image does not depend on the packing: set_vocab and set_corpus invalidate it themselves, and
		// this may run after it was built -- the first toggle, move or count call after evaluations on the automaton)
		c->simg_valid = c->trie.PackStream(&c->simg);
		tr.mark("ensure_trie: stream image packed");
	}
	if (c->hostonly) return TOKSCORE_OK;
	if (!c->device_trie_valid && c->simg_valid) {
		// the stream scan's image: label[] (padded to 16 bytes), info[], slot_id[]
		const uint64_t n = c->simg.lt.size();
		const uint64_t npad = (n + 15) & ~15ull;
		if (npad > c->d_simg_cap) {
			if (c->d_slt) (void)hipFree(c->d_slt);
			if (c->d_sbase) (void)hipFree(c->d_sbase);
			if (c->d_slot_id) (void)hipFree(c->d_slot_id);
			c->d_slt = nullptr, c->d_sbase = nullptr, c->d_slot_id = nullptr;
			c->d_simg_cap = 0;
			HIP_TRY(c, hipMalloc((void **)&c->d_slt, npad));
			HIP_TRY(c, hipMalloc((void **)&c->d_sbase, npad * sizeof(uint16_t)));
			HIP_TRY(c, hipMalloc((void **)&c->d_slot_id, npad * sizeof(uint32_t)));
			c->d_simg_cap = npad;
		}
		HIP_TRY(c, hipMemsetAsync(c->d_slt, 0, npad, c->stream));
		HIP_TRY(c, hipMemsetAsync(c->d_sbase, 0, npad * sizeof(uint16_t), c->stream));
		HIP_TRY(c, hipMemcpyAsync(c->d_slt, c->simg.lt.data(), n, hipMemcpyHostToDevice, c->stream));
		HIP_TRY(c, hipMemcpyAsync(c->d_sbase, c->simg.base.data(), n * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
		HIP_TRY(c, hipMemcpyAsync(c->d_slot_id, c->simg.slot_id.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
	}
	if (!c->device_trie_valid) {
		const uint64_t n = c->packed.slots.size();
		if (n > c->d_slots_cap) {
			if (c->d_slots) (void)hipFree(c->d_slots);
			c->d_slots = nullptr;
			HIP_TRY(c, hipMalloc((void **)&c->d_slots, n * sizeof(uint64_t)));
			c->d_slots_cap = n;
		}
		HIP_TRY(c, hipMemcpyAsync(c->d_slots, c->packed.slots.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice,
		                          c->stream));
		HIP_TRY(c, hipStreamSynchronize(c->stream));
		c->device_trie_valid = true;
		c->trie_epoch++;
		c->pending.clear();
		c->pending_s.clear();
		c->state_valid = false;
	}
	if (!c->d_vocab_valid) {
		if (c->d_vocab_bytes) (void)hipFree(c->d_vocab_bytes);
		c->d_vocab_bytes = nullptr;
		HIP_TRY(c, hipMalloc((void **)&c->d_vocab_bytes, c->h_vocab_bytes.size() + 16));
		if (!c->h_vocab_bytes.empty())
			HIP_TRY(c, hipMemcpy(c->d_vocab_bytes, c->h_vocab_bytes.data(), c->h_vocab_bytes.size(), hipMemcpyHostToDevice));
		c->d_vocab_valid = true;
	}
	return TOKSCORE_OK;
}

// Bring the device trie up to date with the host image: one 8-byte slot per pending toggle,
// enqueued on s (in order after whatever still reads the old state there).
int apply_pending(tokscore_ctx *c, hipStream_t s) {
	if (c->pending.empty() && c->pending_s.empty()) return TOKSCORE_OK;
	c->trie_epoch++;
	if (c->device_trie_valid) {
		const bool staged = c->h_patch && s == c->stream && !c->patch_in_flight && c->pending.size() <= kMaxPatch;
		size_t k = 0;
		for (const auto &pd : c->pending) {
			const uint64_t *src = &c->packed.slots[pd.second];
			if (staged) {
				// (the previous batch was consumed: every caller waits for its stream before it returns)
				c->h_patch[k] = *src;
				src = &c->h_patch[k++];
			}
			HIP_TRY(c, hipMemcpyAsync(c->d_slots + pd.second, src, sizeof(uint64_t), hipMemcpyHostToDevice, s));
		}
		// the same toggles in the stream image (kept up to date on the host by tokscore_toggle):
		// token-end bit and the id the slot counts for
		const bool staged_s = staged && c->pending_s.size() <= kMaxPatch;
		k = 0;
		for (const int64_t t : c->pending_s) {
			if (!c->simg_valid || !c->d_slt) break;
			const uint8_t *src_lt = &c->simg.lt[t];
			const uint32_t *src_id = &c->simg.slot_id[t];
			if (staged_s) {
				c->h_patch_lt[k] = *src_lt, c->h_patch_id[k] = *src_id;
				src_lt = &c->h_patch_lt[k], src_id = &c->h_patch_id[k];
				k++;
			}
			HIP_TRY(c, hipMemcpyAsync(c->d_slt + t, src_lt, sizeof(uint8_t), hipMemcpyHostToDevice, s));
			HIP_TRY(c, hipMemcpyAsync(c->d_slot_id + t, src_id, sizeof(uint32_t), hipMemcpyHostToDevice, s));
		}
		if (staged) c->patch_in_flight = true; // until the next wait on c->stream (fetch_out)
	}
	c->pending.clear();
	c->pending_s.clear();
	return TOKSCORE_OK;
}

// Boundaries (corpus offsets, b[0] = 0 .. b[n] = total) moved forward to guaranteed token starts
// (kernels_generic.hip: snap_chunks_kernel); syn[j] = 0 where none was found within `window`.
int snap_boundaries(tokscore_ctx *c, std::vector<uint64_t> &b, uint32_t window, std::vector<uint32_t> &syn) {
	const size_t nb = b.size();
	syn.assign(nb, 1);
	if (nb < 3) return TOKSCORE_OK;
	uint64_t *d_old = nullptr, *d_new = nullptr;
	uint32_t *d_syn = nullptr;
	hipError_t e = hipMalloc((void **)&d_old, nb * sizeof(uint64_t));
	if (e == hipSuccess) e = hipMalloc((void **)&d_new, nb * sizeof(uint64_t));
	if (e == hipSuccess) e = hipMalloc((void **)&d_syn, nb * sizeof(uint32_t));
	// (both copies from the host: a device-to-device hipMemcpy is not ordered against c->stream,
	// and the kernel below must not be overtaken by it)
	if (e == hipSuccess) e = hipMemcpy(d_old, b.data(), nb * sizeof(uint64_t), hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(d_new, b.data(), nb * sizeof(uint64_t), hipMemcpyHostToDevice);
	if (e == hipSuccess) {
		CorpusDev C{c->d_bytes, c->d_off, c->D, c->total};
		launch_snap_chunks(trie_dev(c), C, d_old, d_new, d_syn, (uint32_t)(nb - 1), (uint32_t)c->packed.max_len, window, c->stream);
		e = hipStreamSynchronize(c->stream);
	}
	if (e == hipSuccess) e = hipMemcpy(b.data(), d_new, nb * sizeof(uint64_t), hipMemcpyDeviceToHost);
	if (e == hipSuccess) e = hipMemcpy(syn.data() + 1, d_syn + 1, (nb - 2) * sizeof(uint32_t), hipMemcpyDeviceToHost);
	if (d_old) (void)hipFree(d_old);
	if (d_new) (void)hipFree(d_new);
	if (d_syn) (void)hipFree(d_syn);
	if (e != hipSuccess) return fail(c, TOKSCORE_E_HIP, std::string("snap_chunks: ") + hipGetErrorString(e));
	return TOKSCORE_OK;
}

// Plan of the stream scan (kernels_stream.hip).  Streams: the corpus cut every stream_bytes, every
// cut moved forward to a guaranteed token start; a cut that could not be moved to one is dropped
// (its stream joins the one before).  Chunks, the unit the incremental re-score selects, are
// streams_per_chunk consecutive streams.  *ok = false: not for this vocabulary / corpus, nothing kept.
int plan_stream(tokscore_ctx *c, bool *ok) {
	*ok = false;
	if (!c->simg_valid || c->total == 0) return TOKSCORE_OK;
	StreamScanPlan sp;
	if (!plan_stream_scan((uint32_t)c->simg.lt.size(), c->packed.max_len, (uint32_t)c->num_cus, &sp)) return TOKSCORE_OK;
	HostTrace tr;
	int st = build_prepared(c);
	if (st || !c->prep_ok) return st;
	tr.mark("plan_stream: prepared corpus");
	uint64_t sbytes = env_u64("TOKSCORE_STREAM", 256);
	sbytes = std::max<uint64_t>(sbytes, std::max<uint64_t>(64, 4 * c->packed.max_len));
	const uint64_t spc = std::max<uint64_t>(1, std::min<uint64_t>(64, env_u64("TOKSCORE_SPC", 16)));
	std::vector<uint64_t> sb; // stream boundaries, corpus offsets
	{
		std::vector<uint64_t> keep;
		keep.swap(c->h_chunk_begin);
		build_chunks(c, sbytes, true);
		sb.swap(c->h_chunk_begin);
		c->h_chunk_begin.swap(keep);
	}
	std::vector<uint32_t> syn;
	const uint32_t window = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(sbytes - 1, 8192));
	tr.mark("plan_stream: stream boundaries");
	st = snap_boundaries(c, sb, window, syn);
	if (st) return st;
	tr.mark("plan_stream: boundaries snapped");
	const size_t ns = sb.size() - 1;
	size_t unsynced = 0;
	for (size_t j = 1; j < ns; j++) unsynced += syn[j] ? 0 : 1;
	// prepared offsets: position p of document d (off[d] <= p < off[d+1]) sits at p + d; the corpus end at total + D
	std::vector<uint32_t> pb(ns + 1);
	{
		size_t d = 0; // largest d with off[d] <= sb[j]
		for (size_t j = 0; j <= ns; j++) {
			while (d < c->D && c->h_off[d + 1] <= sb[j]) d++;
			pb[j] = (uint32_t)(sb[j] + d);
		}
		pb[ns] = (uint32_t)c->prep_total;
		for (size_t j = ns; j-- > 1;)
			if (!syn[j]) pb[j] = pb[j + 1]; // dropped cut: the stream before it runs on to the next kept one
	}
	// chunks: from every spc-th stream boundary, or the first kept cut after it (a chunk must begin
	// where some lane begins a stream, or re-scanning it alone would not be exact).  Without a kept
	// cut in its range a chunk is empty and its bytes belong to the chunk before it.
	c->h_chunk_begin.clear();
	std::vector<uint32_t> cfirst;
	for (size_t j = 0; j < ns; j += spc) {
		size_t k = j;
		while (k < ns && k < j + spc && !syn[k]) k++;
		cfirst.push_back(k == ns || k == j + spc ? 0xFFFFFFFFu : (uint32_t)k);
	}
	{
		uint32_t next = (uint32_t)ns;
		for (size_t j = cfirst.size(); j-- > 0;) {
			if (cfirst[j] == 0xFFFFFFFFu) cfirst[j] = next;
			next = cfirst[j];
		}
	}
	uint32_t widest = 1;
	for (size_t j = 0; j < cfirst.size(); j++) {
		c->h_chunk_begin.push_back(cfirst[j] < ns ? sb[cfirst[j]] : c->total);
		widest = std::max<uint32_t>(widest, (j + 1 < cfirst.size() ? cfirst[j + 1] : (uint32_t)ns) - cfirst[j]);
	}
	// (a lane per stream slot of every selected chunk: a corpus with a very long stretch without
	// any guaranteed token start gets no incremental re-scores)
	c->chunks_synced = widest <= 8 * spc;
	c->widest_chunk = widest;
	cfirst.push_back((uint32_t)ns);
	c->h_chunk_begin.push_back(c->total);
	c->nchunks = (uint32_t)(c->h_chunk_begin.size() - 1);
	tr.mark("plan_stream: offsets and chunks");
	HIP_TRY(c, hipMalloc((void **)&c->d_chunk_first, cfirst.size() * sizeof(uint32_t)));
	HIP_TRY(c, hipMemcpy(c->d_chunk_first, cfirst.data(), cfirst.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	c->chunk_bytes = sbytes * spc;
	c->uniform_chunks = true;
	c->unsynced = (uint32_t)unsynced;
	c->nstreams = (uint32_t)ns;
	c->streams_per_chunk = (uint32_t)spc;
	c->splan = sp;
	// full evaluation: as many equal ranges as the grid has lanes (none shorter than four tokens)
	StreamScanPlan rp = sp;
	const bool transposed = !env_u64("TOKSCORE_NOTRANSPOSE", 0) &&
	                        plan_ranges_scan((uint32_t)c->simg.lt.size(), c->packed.max_len, (uint32_t)c->num_cus, &rp);
	if (!transposed) rp = sp;
	c->rplan = rp;
	const uint64_t lanes = (uint64_t)rp.blocks * rp.waves_per_block * 64;
	uint64_t rb = (c->prep_total + lanes - 1) / lanes;
	rb = std::max<uint64_t>(rb, std::max<uint64_t>(64, 4 * c->packed.max_len));
	rb = std::max<uint64_t>(rb, env_u64("TOKSCORE_RANGE", 0));
	rb = (rb + 7) & ~7ull;
	c->range_bytes = (uint32_t)rb;
	c->nranges = (uint32_t)((c->prep_total + rb - 1) / rb);
	HIP_TRY(c, hipMalloc((void **)&c->d_range_state, (size_t)3 * c->nranges * sizeof(uint32_t)));
	HIP_TRY(c, hipMemset(c->d_range_state, 0, (size_t)3 * c->nranges * sizeof(uint32_t)));
	if (transposed) {
		// each range is followed by the first bytes of the next: a walk may look max_len bytes past a
		// token that itself ends max_len - 1 bytes past the range
		c->halo = rp.halo;
		const uint64_t nblocks = ((uint64_t)c->nranges + 63) / 64;
		const uint64_t tbytes = nblocks * (rb + c->halo) * 64;
		// (the scan addresses the transposed copy with 32-bit offsets)
		if (tbytes < 0xFFFF0000ull && hipMalloc((void **)&c->d_tprep, tbytes) == hipSuccess) {
			launch_transpose_ranges(c->d_prep, c->prep_total + kCorpusPad, (uint32_t)rb, c->halo, (uint32_t)nblocks, c->d_tprep, c->stream);
			HIP_TRY(c, hipStreamSynchronize(c->stream));
		} else {
			(void)hipGetLastError();
			c->d_tprep = nullptr; // no room: the ring scan over the linear copy does the full evaluations too
			c->rplan = sp;
		}
	}
	tr.mark("plan_stream: ranges transposed");
	HIP_TRY(c, hipMalloc((void **)&c->d_pbeg, (ns + 1) * sizeof(uint32_t)));
	HIP_TRY(c, hipMemcpy(c->d_pbeg, pb.data(), (ns + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
	{
		// the streams longest first, for the walk over all of them (tokscore_score_toggles)
		if (getenv("TOKSCORE_TRACE_HOST")) {
			uint32_t longest = 0, over1k = 0;
			for (size_t j = 0; j < ns; j++) {
				longest = std::max(longest, pb[j + 1] - pb[j]);
				over1k += pb[j + 1] - pb[j] > 1024 ? 1 : 0;
			}
			fprintf(stderr, "plan_stream: %zu streams, longest %u bytes, %u over 1 KiB, %zu cuts dropped\n", ns, longest, over1k, unsynced);
		}
		// (by length in buckets, the longest first, streams of one length in corpus order: what a stable sort by
		// descending length gives, without its second and a half per ten million streams)
		std::vector<uint32_t> order(ns);
		{
			constexpr uint32_t kCap = 1u << 16; // lengths from here on share the first bucket, sorted among themselves
			std::vector<uint32_t> count(kCap + 2, 0), big;
			auto len = [&](size_t j) { return pb[j + 1] - pb[j]; };
			for (size_t j = 0; j < ns; j++) {
				if (len(j) >= kCap) big.push_back((uint32_t)j);
				else count[len(j)]++;
			}
			std::stable_sort(big.begin(), big.end(), [&](uint32_t x, uint32_t y) { return len(x) > len(y); });
			std::copy(big.begin(), big.end(), order.begin());
			size_t at = big.size();
			std::vector<size_t> first(kCap, 0);
			for (uint32_t l = kCap; l-- > 0;) first[l] = at, at += count[l];
			for (size_t j = 0; j < ns; j++)
				if (len(j) < kCap) order[first[len(j)]++] = (uint32_t)j;
		}
		tr.mark("plan_stream: streams by length");
		HIP_TRY(c, hipMalloc((void **)&c->d_stream_order, std::max<size_t>(1, ns) * sizeof(uint32_t)));
		HIP_TRY(c, hipMemcpy(c->d_stream_order, order.data(), ns * sizeof(uint32_t), hipMemcpyHostToDevice));
	}
	HIP_TRY(c, hipMalloc((void **)&c->d_chunk_begin, c->h_chunk_begin.size() * sizeof(uint64_t)));
	HIP_TRY(c, hipMemcpy(c->d_chunk_begin, c->h_chunk_begin.data(), c->h_chunk_begin.size() * sizeof(uint64_t),
	                     hipMemcpyHostToDevice));
	c->stream_ready = 0;
	HIP_TRY(c, hipMalloc((void **)&c->d_cursor, kCursorWords * sizeof(uint32_t)));
	HIP_TRY(c, hipMalloc((void **)&c->d_sel, ((size_t)(c->nchunks + 31) / 32 + 1) * sizeof(uint32_t)));
	HIP_TRY(c, hipMalloc((void **)&c->d_sel_list, ((size_t)c->nchunks + 1) * sizeof(uint32_t)));
	// token stream: at most one entry per prepared byte (every token and every separator is at
	// least one byte), so this capacity cannot be exceeded whatever the input
	const uint64_t batches = stream_scan_capacity_batches(c->prep_total, sp.blocks * 16);
	if (batches > 0x7FFFFFF0ull) return fail(c, TOKSCORE_E_CAPACITY, "corpus too large for one token stream");
	c->stream_cap_batches = (uint32_t)batches;
	HIP_TRY(c, hipMalloc(&c->d_stream, batches * kStreamBatch * 1024));
	HIP_TRY(c, hipMalloc((void **)&c->d_fill, batches * sizeof(uint32_t)));
	c->slab_blocks = (uint32_t)std::min<uint64_t>(256, std::max<uint64_t>(16, c->total / (256 * 1024)));
	c->slab_blocks = (uint32_t)std::min<uint64_t>(1024, std::max<uint64_t>(1, env_u64("TOKSCORE_SLAB_BLOCKS", c->slab_blocks)));
	HIP_TRY(c, hipMalloc((void **)&c->d_slab, (size_t)c->slab_blocks * (c->simg.lt.size() + 1) * sizeof(uint32_t)));
	tr.mark("plan_stream: buffers");
	*ok = true;
	return TOKSCORE_OK;
}

int ensure_plan(tokscore_ctx *c) {
	if (c->plan_valid) return TOKSCORE_OK;
	free_plan(c);
	const char *force = getenv("TOKSCORE_SCAN");
	if (!c->force_generic && (!force || !strcmp(force, "stream"))) {
		bool ok = false;
		int st = plan_stream(c, &ok);
		if (st) return st;
		if (ok) {
			c->scan_path = 2;
			c->tile = TileScanPlan{};
			c->plan_valid = true;
			return TOKSCORE_OK;
		}
		free_plan(c);
	}
	TrieDev T = trie_dev(c);
	bool tile_ok = plan_tile_scan(T, c->packed.max_len, (uint32_t)c->V, (uint32_t)c->num_cus, &c->tile);
	if (force && !strcmp(force, "generic")) tile_ok = false;
	if (c->force_generic) tile_ok = false;
	c->scan_path = tile_ok ? 1 : 0;
	if (!tile_ok) c->tile = TileScanPlan{};
	uint64_t dflt_chunk = 2048;
	const bool snap = !env_u64("TOKSCORE_NOSNAP", 0);
	if (tile_ok) {
		// Equal chunks, a whole number of them per resident wave (the queue then hands every
		// wave the same amount of work), each one LDS tile: a little under the tile's capacity,
		// because the snap below moves a boundary forward by some tens of bytes.
		const uint64_t waves = (uint64_t)c->tile.blocks * c->tile.waves_per_block;
		const uint64_t nominal = 64ull * (c->tile.seg_bytes - 8);
		const uint64_t per_wave = std::max<uint64_t>(1, (c->total + waves * nominal - 1) / (waves * nominal));
		dflt_chunk = std::max<uint64_t>(nominal / 4, (c->total + waves * per_wave - 1) / (waves * per_wave));
	}
	c->chunk_bytes = env_u64("TOKSCORE_CHUNK", dflt_chunk);
	if (c->chunk_bytes < 64) c->chunk_bytes = 64;
	if (c->chunk_bytes < 4 * c->packed.max_len) c->chunk_bytes = 4 * c->packed.max_len;
	build_chunks(c, c->chunk_bytes, snap);
	c->uniform_chunks = snap;
	c->nchunks = (uint32_t)(c->h_chunk_begin.size() - 1);
	HIP_TRY(c, hipMalloc((void **)&c->d_chunk_begin, c->h_chunk_begin.size() * sizeof(uint64_t)));
	HIP_TRY(c, hipMemcpy(c->d_chunk_begin, c->h_chunk_begin.data(), c->h_chunk_begin.size() * sizeof(uint64_t),
	                     hipMemcpyHostToDevice));
	HIP_TRY(c, hipMalloc((void **)&c->d_chunk_state, (size_t)3 * (c->nchunks + 1) * sizeof(uint32_t)));
	// Move the boundaries that fell inside a document forward to the next position where every
	// greedy chain starts a token (kernels_generic.hip: snap_chunks_kernel).  Chunks that begin
	// there need no entry repair; when all do, an evaluation launches no fix-up kernels at all.
	c->unsynced = 0;
	if (c->nchunks > 1 && snap) {
		uint64_t *d_new = nullptr;
		uint32_t *d_syn = nullptr;
		const size_t nb = c->h_chunk_begin.size();
		HIP_TRY(c, hipMalloc((void **)&d_new, nb * sizeof(uint64_t)));
		hipError_t e = hipMalloc((void **)&d_syn, nb * sizeof(uint32_t));
		std::vector<uint32_t> syn(nb, 1);
		if (e == hipSuccess) e = hipMemcpy(d_new, c->h_chunk_begin.data(), nb * sizeof(uint64_t), hipMemcpyHostToDevice);
		if (e == hipSuccess) {
			CorpusDev C{c->d_bytes, c->d_off, c->D, c->total};
			const uint32_t window = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(c->chunk_bytes - 1, 8192));
			launch_snap_chunks(T, C, c->d_chunk_begin, d_new, d_syn, c->nchunks, (uint32_t)c->packed.max_len, window, c->stream);
			e = hipStreamSynchronize(c->stream);
		}
		if (e == hipSuccess) e = hipMemcpy(c->h_chunk_begin.data(), d_new, nb * sizeof(uint64_t), hipMemcpyDeviceToHost);
		if (e == hipSuccess) e = hipMemcpy(syn.data() + 1, d_syn + 1, (c->nchunks - 1) * sizeof(uint32_t), hipMemcpyDeviceToHost);
		if (e == hipSuccess) e = hipMemcpy(c->d_chunk_begin, c->h_chunk_begin.data(), nb * sizeof(uint64_t), hipMemcpyHostToDevice);
		(void)hipFree(d_new);
		if (d_syn) (void)hipFree(d_syn);
		if (e != hipSuccess) return fail(c, TOKSCORE_E_HIP, std::string("snap_chunks: ") + hipGetErrorString(e));
		for (uint32_t j = 1; j < c->nchunks; j++) c->unsynced += syn[j] ? 0 : 1;
	} else {
		const std::vector<uint64_t> &off = c->h_off;
		for (uint32_t j = 1; j < c->nchunks; j++)
			c->unsynced += std::binary_search(off.begin(), off.end(), c->h_chunk_begin[j]) ? 0 : 1;
	}
	{
		std::vector<uint64_t> first(c->nchunks + 1);
		for (uint32_t j = 0; j <= c->nchunks; j++)
			first[j] = (uint64_t)(std::lower_bound(c->h_off.begin(), c->h_off.end(), c->h_chunk_begin[j]) - c->h_off.begin());
		HIP_TRY(c, hipMalloc((void **)&c->d_chunk_first_doc, first.size() * sizeof(uint64_t)));
		HIP_TRY(c, hipMemcpy(c->d_chunk_first_doc, first.data(), first.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
	}
	c->stream_ready = 0;
	HIP_TRY(c, hipMalloc((void **)&c->d_cursor, kCursorWords * sizeof(uint32_t)));
	HIP_TRY(c, hipMalloc((void **)&c->d_sel, ((size_t)(c->nchunks + 31) / 32 + 1) * sizeof(uint32_t)));
	HIP_TRY(c, hipMalloc((void **)&c->d_sel_list, ((size_t)c->nchunks + 1) * sizeof(uint32_t)));
	if (c->scan_path == 1) {
		// Token stream: one entry per emitted token plus the repair rounds' +/- entries, in
		// log blocks that are partly empty.  3 entries per corpus byte is far above anything
		// but adversarial input; an overflow is detected and the evaluation redone on the
		// generic path, never silently truncated.
		const uint64_t entry_bytes = c->tile.wide_ids ? 4 : 2;
		const uint64_t block_bytes = stream_block_bytes(c->tile.wide_ids);
		const uint64_t waves = (uint64_t)c->tile.blocks * c->tile.waves_per_block;
		// (TOKSCORE_STREAM_DIV: tests shrink the capacity to exercise the overflow report and the fallback)
		const uint64_t batches = (3 * c->total * entry_bytes) / (block_bytes * kStreamBatch) / std::max<uint64_t>(1, env_u64("TOKSCORE_STREAM_DIV", 1)) +
		                         (env_u64("TOKSCORE_STREAM_DIV", 1) > 1 ? 1 : 2 * waves + 16);
		if (batches > 0x7FFFFFF0ull) return fail(c, TOKSCORE_E_CAPACITY, "corpus too large for one token stream");
		c->stream_cap_batches = (uint32_t)batches;
		HIP_TRY(c, hipMalloc(&c->d_stream, batches * kStreamBatch * block_bytes));
		HIP_TRY(c, hipMalloc((void **)&c->d_fill, batches * sizeof(uint32_t)));
		c->slab_blocks = (uint32_t)std::min<uint64_t>(256, std::max<uint64_t>(16, c->total / (256 * 1024)));
		c->slab_blocks = (uint32_t)std::min<uint64_t>(1024, std::max<uint64_t>(1, env_u64("TOKSCORE_SLAB_BLOCKS", c->slab_blocks)));
		HIP_TRY(c, hipMalloc((void **)&c->d_slab, (size_t)c->slab_blocks * (c->V + 4) * sizeof(uint32_t)));
	}
	c->plan_valid = true;
	return TOKSCORE_OK;
}

// The automaton's image (host + device) from the vocabulary as it is.  false: not available.
bool upload_dfa_image(tokscore_ctx *c) {
	DfaOptions opt;
	opt.max_slots = c->dfa_max_slots;
	opt.lds_slots = dfa_lds_slots_max(c->dfa_lds_cap);
	opt.sample = c->sample.empty() ? nullptr : c->sample.data();
	opt.sample_bytes = c->sample.size();
	opt.corpus_bytes = c->total + c->D;
	opt.fail_mode = c->dfa_fail;
	// (a flattened image of that shape is turned away by ensure_dfa's policy: not built in the first place, unless the
	// policy is lifted or that form is asked for)
	if (c->dfa_form_knob != 1 && c->dfa_max_virtual_ppm < 1000000) opt.max_virtual_per_node = 8;
	c->dfa_prior_keys.clear(), c->dfa_prior_counts.clear();
	for (const auto &kv : c->dfa_prior) c->dfa_prior_keys.push_back(kv.first), c->dfa_prior_counts.push_back(kv.second);
	if (!c->dfa_prior.empty()) opt.prior_keys = c->dfa_prior_keys.data(), opt.prior_counts = c->dfa_prior_counts.data(), opt.prior_n = c->dfa_prior.size();
	std::string why;
	const auto t0 = std::chrono::steady_clock::now();
	// (a new image object each time: the old one lends its sample counts while the vocabulary is what it was built from)
	DfaImage fresh;
	opt.vocab_tag = c->toggle_version + 1;
	if (c->dimg.valid && (c->dimg.fail_code != 0) == c->dfa_fail) opt.reuse = &c->dimg;
	if (!c->trie.BuildDfa(&fresh, opt, &why)) return false;
	c->dimg = std::move(fresh);
	c->dfa_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	const DfaImage &g = c->dimg;
	const size_t n = g.nslots, nh = g.hot_slots, ne = g.ev_miss.size(), ng = g.gof.size();
	for (uint32_t **b : {&c->d_dhot, &c->d_dgof, &c->d_dslot, &c->d_dev_miss, &c->d_dev_arr, &c->d_dpool}) {
		if (*b) (void)hipFree(*b);
		*b = nullptr;
	}
	bool ok = hipMalloc((void **)&c->d_dhot, (nh + 16) * sizeof(uint32_t)) == hipSuccess &&
	          hipMalloc((void **)&c->d_dgof, (ng + 16) * sizeof(uint32_t)) == hipSuccess &&
	          hipMalloc((void **)&c->d_dslot, (2 * n + 16) * sizeof(uint32_t)) == hipSuccess &&
	          hipMalloc((void **)&c->d_dev_miss, (ne + 16) * sizeof(uint32_t)) == hipSuccess &&
	          hipMalloc((void **)&c->d_dev_arr, (ne + 16) * sizeof(uint32_t)) == hipSuccess &&
	          hipMalloc((void **)&c->d_dpool, (g.pool.size() + 16) * sizeof(uint32_t)) == hipSuccess;
	ok = ok && hipMemsetAsync(c->d_dhot + nh, 0xFF, 16 * sizeof(uint32_t), c->stream) == hipSuccess &&
	     hipMemcpyAsync(c->d_dhot, g.hot.data(), nh * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream) == hipSuccess &&
	     hipMemcpyAsync(c->d_dgof, g.gof.data(), ng * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream) == hipSuccess &&
	     hipMemcpyAsync(c->d_dslot, g.slot.data(), 2 * n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream) == hipSuccess &&
	     hipMemcpyAsync(c->d_dev_miss, g.ev_miss.data(), ne * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream) == hipSuccess &&
	     hipMemcpyAsync(c->d_dev_arr, g.ev_arr.data(), ne * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream) == hipSuccess &&
	     (g.pool.empty() || hipMemcpyAsync(c->d_dpool, g.pool.data(), g.pool.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream) == hipSuccess) &&
	     hipStreamSynchronize(c->stream) == hipSuccess;
	if (!ok) (void)hipGetLastError();
	return ok;
}

DfaDev dfa_dev(const tokscore_ctx *c) {
	DfaDev A;
	A.hot = c->d_dhot, A.gof = c->d_dgof, A.slot = (const uint2 *)c->d_dslot, A.ev_miss = c->d_dev_miss, A.ev_arr = c->d_dev_arr, A.pool = c->d_dpool;
	A.nslots = c->dimg.nslots;
	A.hot_slots = c->dimg.hot_slots;
	A.cold_base = c->dimg.cold_base;
	A.stride = c->dimg.stride;
	A.fail_code = c->dimg.fail_code;
	return A;
}

DfaWork dfa_work(const tokscore_ctx *c) {
	DfaWork W;
	W.tcodes = (const uint2 *)c->d_dtcodes;
	W.range_bytes = c->dplan.range_bytes;
	W.nwb = c->dplan.nwb;
	W.nranges = c->dplan.nwb * 64 * c->dplan.chains;
	W.entry = c->d_dstate;
	W.exit_a = c->d_dstate + W.nranges;
	W.exit_b = c->d_dstate + 2 * (size_t)W.nranges;
	W.exit_tag = c->d_dtag;
	W.cold_miss = c->d_dcold;
	W.cold_arr = c->d_dcold + c->dimg.nslots + 1;
	W.cold_hit = c->d_dcold + 2 * ((size_t)c->dimg.nslots + 1);
	// (the failure form sends a share of its lookups to the global array by design: their hits are counted only while
	// a re-layout can still use them)
	if (c->dfa_fail && !(c->dfa_relayout_on && c->dfa_relayouts < c->dfa_relayouts_max)) W.cold_hit = nullptr;
	return W;
}

// The failure form's image takes a second of host time for a trie of millions of nodes, and with more than 16 trie nodes
// per LDS slot its sample never kept 70 % of the lookups in LDS (config 4: 1.5 M nodes, 38 % beyond): not built then.
bool fail_form_worth_a_try(const tokscore_ctx *c) { return c->trie.node_count() <= 16ull * dfa_lds_slots_max(c->dfa_lds_cap); }

// Everything the automaton scan needs for the resident corpus and the vocabulary as it is; leaves
// dfa_state at 1 (ready) or -1 (not for this vocabulary / corpus: the plan's scan serves every evaluation).
void ensure_dfa(tokscore_ctx *c) {
	if (c->dfa_state != 0) return;
	c->dfa_state = -1;
	const uint64_t prep_total = c->total + c->D;
	if (!c->dfa_enabled || c->force_generic || c->D == 0 || c->total == 0 || c->V + 5 >= 0x7FFFFFFFull || prep_total + kCorpusPad >= 0xFFFF0000ull) return;
	const auto t0 = std::chrono::steady_clock::now();
	HostTrace tr;
	c->dfa_fail = c->dfa_form_knob == 2;
	if (!upload_dfa_image(c)) {
		if (c->dfa_fail || c->dfa_form_knob == 1 || !fail_form_worth_a_try(c)) return;
		c->dfa_fail = true; // (more virtual entries than the limit: the failure form has none)
		if (!upload_dfa_image(c)) return;
	}
	if (!c->dfa_fail) {
		// The automaton pays off while LDS answers nearly every lookup and arrivals on virtual entries are rare (the
		// loop keeps both behind one rarely taken branch).  A vocabulary whose tokens' prefixes are mostly not tokens
		// (the reference annealer's own output: a third of the steps arrive on a virtual entry, almost half need the
		// global array) is served better by the plan's scan: measured 9.8 ms against 1.0 ms per 100 MB.
		const DfaImage &g = c->dimg;
		const double steps = (double)std::max<uint64_t>(1, g.sample_steps);
		const double limit_cold = (double)c->dfa_max_cold_ppm * 1e-6, limit_virtual = (double)c->dfa_max_virtual_ppm * 1e-6;
		if (g.sample_steps != 0 && ((double)g.sample_cold / steps > limit_cold || (double)g.sample_virtual / steps > limit_virtual)) {
			if (c->dfa_form_knob == 1 || !fail_form_worth_a_try(c)) return;
			c->dfa_fail = true;
			if (!upload_dfa_image(c)) return;
		}
	}
	if (c->dfa_fail && c->dfa_form_knob != 2) {
		const DfaImage &g = c->dimg;
		if (g.sample_steps != 0 && (double)g.sample_cold / (double)g.sample_steps > (double)c->dfa_fail_max_cold_ppm * 1e-6) return;
	}
	tr.mark("ensure_dfa: image(s) built + uploaded");
	DfaScanPlan &P = c->dplan;
	// (the failure form waits for the global array in most steps: four ranges per lane share each wait)
	if (!plan_dfa_scan(c->dimg.hot_slots, prep_total, (uint32_t)c->num_cus, c->dfa_fail && c->dfa_chains == 0 ? 4 : c->dfa_chains, &P)) return;
	const uint64_t nranges = (uint64_t)P.nwb * 64 * P.chains;
	const uint64_t covered = nranges * P.range_bytes; // >= prep_total: the rest reads as separators
	const uint64_t lin_bytes = std::max<uint64_t>(covered, prep_total) + kCorpusPad;
	const uint32_t ncold = c->dimg.nslots;
	c->dstream_cap = 0;
	const uint64_t cap = dfa_stream_capacity_batches(P);
	if (cap > 0x7FFFFFF0ull) return;
	uint32_t *d_flags = nullptr;
	uint32_t flags = 1;
	c->dslab_blocks = (uint32_t)std::min<uint64_t>(256, std::max<uint64_t>(16, c->total / (256 * 1024)));
	c->dslab_blocks = (uint32_t)std::min<uint64_t>(256, std::max<uint64_t>(2, env_u64("TOKSCORE_DFA_SLAB_BLOCKS", c->dslab_blocks)));
	bool ok = hipMalloc((void **)&c->d_dcodes, lin_bytes) == hipSuccess && hipMalloc((void **)&d_flags, sizeof(uint32_t)) == hipSuccess &&
	          hipMalloc(&c->d_dtcodes, covered) == hipSuccess &&
	          hipMalloc((void **)&c->d_dstate, 3 * nranges * sizeof(uint2)) == hipSuccess &&
	          hipMalloc((void **)&c->d_dtag, nranges * sizeof(uint32_t)) == hipSuccess &&
	          hipMalloc((void **)&c->d_dcold, (3 * (size_t)ncold + 3) * sizeof(uint32_t)) == hipSuccess &&
	          hipMalloc(&c->d_dstream_m, cap * kStreamBatch * 1024) == hipSuccess &&
	          hipMalloc(&c->d_dstream_a, cap * kStreamBatch * 1024) == hipSuccess &&
	          hipMalloc((void **)&c->d_dfill_m, cap * sizeof(uint32_t)) == hipSuccess &&
	          hipMalloc((void **)&c->d_dfill_a, cap * sizeof(uint32_t)) == hipSuccess &&
	          hipMalloc((void **)&c->d_dcursor, 16 * sizeof(uint32_t)) == hipSuccess &&
	          hipMalloc((void **)&c->d_dslab, (size_t)c->dslab_blocks * (P.lds_slots + 2) * sizeof(uint32_t)) == hipSuccess;
	ok = ok && hipMemsetAsync(c->d_dcodes, 0, lin_bytes, c->stream) == hipSuccess &&
	     hipMemsetAsync(d_flags, 0, sizeof(uint32_t), c->stream) == hipSuccess &&
	     hipMemsetAsync(c->d_dtag, 0, nranges * sizeof(uint32_t), c->stream) == hipSuccess &&
	     hipMemsetAsync(c->d_dcold, 0, (3 * (size_t)ncold + 3) * sizeof(uint32_t), c->stream) == hipSuccess &&
	     hipMemsetAsync(c->d_dcursor, 0, 16 * sizeof(uint32_t), c->stream) == hipSuccess;
	tr.mark("ensure_dfa: buffers allocated");
	if (ok) {
		CorpusDev C{c->d_bytes, c->d_off, c->D, c->total};
		ByteCodes M;
		memcpy(M.code, c->dimg.code, 256);
		launch_prep_corpus(C, M, c->d_dcodes, prep_total, d_flags, c->stream);
		launch_dfa_transpose(c->d_dcodes, prep_total, P.range_bytes, P.chains, P.nwb, c->d_dtcodes, c->stream);
		launch_dfa_reset_states(dfa_work(c), c->dimg.start_word(), c->stream);
		launch_dfa_warm_entries(dfa_dev(c), dfa_work(c), c->d_dcodes, c->dimg.start_word(), c->dfa_warm_bytes, c->stream);
		launch_dfa_prepare(nullptr, (uint32_t)c->V, c->D, c->total, StreamDev{c->d_dstream_m, (uint32_t)cap, c->d_dfill_m, c->d_dcursor},
		                   StreamDev{c->d_dstream_a, (uint32_t)cap, c->d_dfill_a, c->d_dcursor + 4}, P.blocks * P.waves_per_block,
		                   c->d_dcursor + 8, c->stream);
		ok = hipMemcpyAsync(&flags, d_flags, sizeof flags, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
		     hipStreamSynchronize(c->stream) == hipSuccess && hipGetLastError() == hipSuccess;
	}
	if (d_flags) (void)hipFree(d_flags);
	tr.mark("ensure_dfa: corpus prepared (synced)");
	if (!ok || flags != 0) { // (flags: a document holds a NUL byte -- the separator's code inside a document)
		(void)hipGetLastError();
		free_dfa(c);
		c->dfa_state = -1;
		return;
	}
	c->dstream_cap = (uint32_t)cap;
	c->dfa_version = c->toggle_version;
	c->dfa_stale_evals = 0;
	c->dfa_epoch = 0;
	c->dfa_state = 1;
	c->dfa_plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// A new image for the same corpus (the codes and the ranges stay: neither toggles nor another choice of
// what LDS holds change the trie's edges), every range's entry back to a guess.
bool rebuild_dfa(tokscore_ctx *c) {
	(void)hipDeviceSynchronize(); // (an evaluation on a caller's stream may still read what is replaced here)
	const uint32_t old_codes = c->dimg.ncodes;
	if (!upload_dfa_image(c) || c->dimg.ncodes != old_codes) return false;
	DfaScanPlan P = c->dplan;
	const uint32_t lds = c->dimg.hot_slots;
	// (the cold counters and the slab are sized by the image)
	if (c->d_dcold) (void)hipFree(c->d_dcold);
	if (c->d_dslab) (void)hipFree(c->d_dslab);
	c->d_dcold = nullptr, c->d_dslab = nullptr;
	P.lds_slots = lds;
	P.lds_bytes = ((lds + 3) & ~3u) * 4;
	c->dplan = P;
	const uint32_t ncold = c->dimg.nslots;
	bool ok = hipMalloc((void **)&c->d_dcold, (3 * (size_t)ncold + 3) * sizeof(uint32_t)) == hipSuccess &&
	          hipMalloc((void **)&c->d_dslab, (size_t)c->dslab_blocks * (P.lds_slots + 2) * sizeof(uint32_t)) == hipSuccess &&
	          hipMemsetAsync(c->d_dcold, 0, (3 * (size_t)ncold + 3) * sizeof(uint32_t), c->stream) == hipSuccess &&
	          hipMemsetAsync(c->d_dtag, 0, (size_t)P.nwb * 64 * P.chains * sizeof(uint32_t), c->stream) == hipSuccess;
	if (!ok) {
		(void)hipGetLastError();
		return false;
	}
	launch_dfa_reset_states(dfa_work(c), c->dimg.start_word(), c->stream);
	launch_dfa_warm_entries(dfa_dev(c), dfa_work(c), c->d_dcodes, c->dimg.start_word(), c->dfa_warm_bytes, c->stream);
	c->dfa_version = c->toggle_version;
	c->dfa_stale_evals = 0;
	c->dfa_epoch = 0;
	c->dfa_evals = 0;
	return true;
}

// After toggles: the counts of the entries LDS did not hold stay (they are keyed by trie node and code, which a
// toggle does not renumber), so the new image starts from what the corpus is known to use.
bool refresh_dfa(tokscore_ctx *c) { return rebuild_dfa(c); }

// After an evaluation on the automaton: which entries of the global array the corpus really uses (the
// host's sample saw 2 MB of it).  Where that is more than a few lookups in 100,000, LDS is laid out
// again with those entries counted in (results never depend on it: only where a lookup is answered).
// (a second re-layout was measured to change nothing: 7.5 k second looks per 20 MB after one, 7.9 k after two)
constexpr uint32_t kDfaRelayoutsDefault = 1;
constexpr uint32_t kDfaArrivalBlocks = 16; // workgroups of the histogram launch that bin the arrival stream (a few percent of the miss stream's entries at most)
bool relayout_dfa(tokscore_ctx *c) {
	(void)hipDeviceSynchronize();
	const DfaImage &g = c->dimg;
	const size_t n = g.nslots - g.hot_slots;
	if (n == 0) return true;
	std::vector<uint32_t> hits(n);
	const DfaWork W = dfa_work(c);
	if (hipMemcpy(hits.data(), W.cold_hit, n * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) {
		(void)hipGetLastError();
		return false;
	}
	uint64_t total = 0;
	for (uint32_t h : hits) total += h;
	c->dfa_second_hits = total;
	c->dfa_relayouts++;
	if (total * 40000 < (c->total + c->D) * (uint64_t)c->dfa_evals) return true; // rare enough as it is
	for (size_t i = 0; i < n; i++) {
		if (!hits[i] || g.ent_key[i] == DfaImage::kEvNone) continue;
		uint32_t &p = c->dfa_prior[g.ent_key[i]];
		const uint64_t v = (uint64_t)p + (hits[i] + c->dfa_evals - 1) / c->dfa_evals;
		p = (uint32_t)std::min<uint64_t>(v, 0xFFFFFFFFull);
	}
	return rebuild_dfa(c);
}

int ensure_out(tokscore_ctx *c) {
	const uint64_t n = c->V + 5;
	if (n > c->d_out_cap) {
		if (c->d_out) (void)hipFree(c->d_out);
		c->d_out = nullptr;
		HIP_TRY(c, hipMalloc((void **)&c->d_out, n * sizeof(unsigned long long)));
		if (c->h_out) (void)hipHostFree(c->h_out);
		c->h_out = nullptr;
		HIP_TRY(c, hipHostMalloc((void **)&c->h_out, (n + 1) * sizeof(uint64_t), hipHostMallocDefault));
		c->d_out_cap = n;
	}
	if (!c->h_patch) {
		HIP_TRY(c, hipHostMalloc((void **)&c->h_patch, kMaxPatch * 16, hipHostMallocDefault));
		c->h_patch_id = reinterpret_cast<uint32_t *>(c->h_patch + kMaxPatch);
		c->h_patch_lt = reinterpret_cast<uint8_t *>(c->h_patch_id + kMaxPatch);
	}
	return TOKSCORE_OK;
}

// d_out and the scan's overflow flag into h_out, one wait.
int fetch_out(tokscore_ctx *c, hipStream_t s) {
	const uint64_t n = c->V + 5;
	c->h_out[n] = 0;
	HIP_TRY(c, hipMemcpyAsync(c->h_out, c->d_out, n * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
	const uint32_t *cursor = c->dfa_used_last ? c->d_dcursor : (c->scan_path >= 1 ? c->d_cursor : nullptr);
	if (cursor) HIP_TRY(c, hipMemcpyAsync(c->h_out + n, cursor, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
	HIP_TRY(c, hipStreamSynchronize(s));
	c->patch_in_flight = false;
	return TOKSCORE_OK;
}

// Enqueue one full evaluation into d_out (V+5 u64) on stream s.
int enqueue_score(tokscore_ctx *c, unsigned long long *d_out, hipStream_t s) {
	HostTrace tr;
	int st = TOKSCORE_OK;
	if (!c->vocab_set) return fail(c, TOKSCORE_E_STATE, "vocabulary not set");
	if (!c->corpus_set) return fail(c, TOKSCORE_E_STATE, "corpus not set");
	c->state_valid = false;
	// the automaton scan, while its image is the vocabulary as it is; after toggles the plan's own scan
	// (which patches one slot per toggle) until two full evaluations in a row met no new toggle
	if (c->dfa_invalid) {
		(void)hipDeviceSynchronize();
		free_dfa(c);
		c->dfa_invalid = false;
	}
	ensure_dfa(c);
	tr.mark("enqueue_score: ensure_dfa");
	bool use_dfa = c->dfa_state == 1 && c->dfa_enabled;
	if (use_dfa && c->dfa_version != c->toggle_version) {
		use_dfa = false;
		if (++c->dfa_stale_evals >= 2) {
			if (refresh_dfa(c)) use_dfa = true;
			else {
				free_dfa(c);
				c->dfa_state = -1;
			}
		}
	} else if (use_dfa && c->dfa_evals >= 1 && c->dfa_relayouts < c->dfa_relayouts_max && c->dfa_relayout_on) {
		if (!relayout_dfa(c)) {
			free_dfa(c);
			c->dfa_state = -1;
			use_dfa = false;
		}
	}
	if (use_dfa) c->dfa_evals++;
	// The packed trie of the other scan paths (hotness sample, double array, stream image, their device copies: 100 ms
	// of host time for the bench vocabulary) and their plan: only when this evaluation runs on them, or toggles wait
	// to be applied to them -- the automaton is built from the trie itself.  Every other entry point that needs them
	// (toggle, moves, counting, incremental re-scores, tokenize) asks for them itself.
	if (!use_dfa || c->packed_valid) { // (once they exist they are kept up to date as ever: nothing to do when they are)
		st = ensure_trie(c);
		if (st) return st;
		tr.mark("enqueue_score: ensure_trie");
	}
	if (!use_dfa) {
		st = ensure_plan(c);
		if (st) return st;
	}
	st = apply_pending(c, s); // (the device trie and stream image of the other paths: kept up to date whichever path runs)
	if (st) return st;
	TrieDev T = trie_dev(c);
	CorpusDev C{c->d_bytes, c->d_off, c->D, c->total};
	ChunkState S{c->d_chunk_begin, c->d_chunk_state, c->d_chunk_state + (c->nchunks + 1),
	             c->d_chunk_state + 2 * (size_t)(c->nchunks + 1), c->nchunks};
	// fix-up kernels only when some chunk starts at a speculated position
	const bool repair = c->unsynced != 0 && c->scan_path != 2; // (the stream scan drops cuts it could not snap)
	// events: scan start, scan end, [fix-up forked,] end.  Each record holds the stream for about
	// 5 us, so the evaluation start shares the scan-start event (prepare_kernel, 4 us, is outside)
	// and without fix-up kernels the "fix-up" interval is empty.
	hipEvent_t *ev = nullptr, *view = nullptr;
	const bool scan_only = c->timing_scan_only;
	if (c->timing_on && (!scan_only || c->timing_skipped++ % c->timing_every == 0)) {
		ev = c->ring[c->calls % tokscore_ctx::kRing];
		view = c->view[c->calls % tokscore_ctx::kRing];
		c->calls++;
		view[0] = view[1] = ev[1];
		view[2] = ev[2];
		view[3] = repair && !scan_only ? ev[3] : ev[2];
		view[4] = scan_only ? ev[2] : ev[4];
		c->ev = view;
	}
	auto mark = [&](int k) -> hipError_t { return ev && (!scan_only || k == 1 || k == 2) ? hipEventRecord(ev[k], s) : hipSuccess; };
	c->dfa_used_last = use_dfa;
	if (use_dfa) {
		// scan (clears and presets the output itself), then fix-up + histograms, then the reduction
		// (which also leaves the cursors ready for the next evaluation)
		const StreamDev Qm{c->d_dstream_m, c->dstream_cap, c->d_dfill_m, c->d_dcursor};
		const StreamDev Qa{c->d_dstream_a, c->dstream_cap, c->d_dfill_a, c->d_dcursor + 4};
		const DfaDev A = dfa_dev(c);
		const DfaWork W = dfa_work(c);
		launch_dfa_scan(A, W, Qm, Qa, c->dplan, s, OutInit{d_out, (uint32_t)c->V, c->D, c->total}, ev ? ev[1] : nullptr, ev ? ev[2] : nullptr);
		// (the histogram's workgroups take a CU each, LDS full of bins: no more of them than CUs, both streams together)
		const uint32_t arr_blocks = std::min<uint32_t>(kDfaArrivalBlocks, std::max<uint32_t>(1, c->dslab_blocks / 16));
		launch_dfa_count(A, W, Qm, Qa, c->d_dcodes, c->d_dcursor + 8, ++c->dfa_epoch, d_out, (uint32_t)c->V + 3, c->d_dslab,
		                 c->dslab_blocks - arr_blocks, arr_blocks, c->dplan.blocks * c->dplan.waves_per_block, s);
	} else if (c->scan_path == 2) {
		StreamDev Q{c->d_stream, c->stream_cap_batches, c->d_fill, c->d_cursor};
		const uint32_t first_batches = c->d_tprep ? c->rplan.blocks * c->rplan.waves_per_block : 0u;
		const bool ready = first_batches != 0 && c->stream_ready == first_batches;
		c->stream_ready = 0;
		if (!ready) launch_prepare(d_out, (uint32_t)c->V, c->D, c->total, Q, s, first_batches);
		const StreamWork W = stream_work(c, false);
		if (c->d_tprep) {
			// the scan's two marks come from its dispatch itself (kernels.h)
			OutInit init;
			if (ready) init = OutInit{d_out, (uint32_t)c->V, c->D, c->total};
			launch_scan_ranges(stream_trie_dev(c), W, Q, c->rplan, s, init, ev ? ev[1] : nullptr, ev ? ev[2] : nullptr);
		} else {
			HIP_TRY(c, mark(1));
			launch_scan_stream(stream_trie_dev(c), W, Q, c->splan, s);
			HIP_TRY(c, mark(2));
		}
		// ranges whose entry was not the previous range's exit (none while the vocabulary stays as it is)
		c->fix_epoch = c->fix_epoch % 65535u + 1u;
		if (c->fix_epoch == 1u && c->fix_epoch_wrapped && c->d_range_state && c->nranges) {
			// the 16-bit tag starts over: an exit_b word written exactly one cycle ago would carry the current tag
			// again.  Cleared here, before the fix-up that reads them (the scan does not touch exit_b).
			HIP_TRY(c, hipMemsetAsync(c->d_range_state + 2 * (size_t)c->nranges, 0, (size_t)c->nranges * sizeof(uint32_t), s));
		}
		c->fix_epoch_wrapped = true;
		launch_fixup_and_histogram_slots(stream_trie_dev(c), W, c->d_range_state + 2 * (size_t)c->nranges, c->d_cursor + 2, c->fix_epoch, Q,
		                                 (uint32_t)c->simg.lt.size(), c->d_slot_id, d_out, (uint32_t)c->V + 3, c->d_slab, c->slab_blocks, s,
		                                 first_batches);
		c->stream_ready = first_batches;
		c->entries_epoch = c->d_tprep && s == c->stream ? c->trie_epoch : 0;
	} else if (c->scan_path == 1) {
		StreamDev Q{c->d_stream, c->stream_cap_batches, c->d_fill, c->d_cursor};
		c->stream_ready = 0;
		launch_prepare(d_out, (uint32_t)c->V, c->D, c->total, Q, s);
		HIP_TRY(c, mark(1));
		launch_scan_tile(T, C, S, Q, c->tile, c->d_chunk_first_doc, c->packed.max_len, s);
		HIP_TRY(c, mark(2));
		if (repair) {
			// the chunk-boundary fix-up (a few serial walks, latency bound) and the histogram of
			// the token stream only share d_out, which both update with atomics: side by side
			HIP_TRY(c, hipEventRecord(c->ev_fork, s));
			HIP_TRY(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
			launch_fixup(T, C, S, d_out, c->side);
			HIP_TRY(c, hipEventRecord(c->ev_join, c->side));
			HIP_TRY(c, mark(3));
		}
		launch_histogram(Q, c->tile, (uint32_t)c->V, d_out, c->d_slab, c->slab_blocks, s);
		if (repair) {
			HIP_TRY(c, hipStreamWaitEvent(s, c->ev_join, 0));
			launch_finalize(d_out, (uint32_t)c->V, c->D, c->total, s); // totals again, repairs included
		}
	} else {
		HIP_TRY(c, hipMemsetAsync(d_out, 0, (c->V + 5) * sizeof(unsigned long long), s));
		HIP_TRY(c, mark(1));
		launch_scan_generic(T, C, S, d_out, s);
		HIP_TRY(c, mark(2));
		if (repair) {
			launch_fixup(T, C, S, d_out, s);
			HIP_TRY(c, mark(3));
		}
		launch_finalize(d_out, (uint32_t)c->V, c->D, c->total, s);
	}
	HIP_TRY(c, mark(4));
	HIP_TRY(c, hipGetLastError());
	c->timing_pending = ev != nullptr;
	return TOKSCORE_OK;
}

// Occurrence index for incremental re-scores and batched moves (kernels_generic.hip:
// index_all_kernel).  Skipped when it would take more than TOKSCORE_INDEX_MB of HBM (default 32768:
// config 2's per-GPU share, 2.5 GB at V = 32 k, needs 2.4 GB, and without it a batch of moves falls
// back to a re-score per move -- 147 instead of tens of thousands of moves/s) or more than half of
// what is free on the device; the re-score then searches the corpus for each toggled string instead.
void build_index(tokscore_ctx *c) {
	c->index_state = -1;
	std::vector<int32_t> slot_row;
	const uint32_t rows = c->trie.StringRows(c->packed.slots.size(), &slot_row, &c->tok_row);
	const uint32_t wpr = (c->nchunks + 31) / 32;
	const uint64_t bytes = (uint64_t)rows * wpr * sizeof(uint32_t);
	size_t free_b = 0, total_b = 0;
	if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
		(void)hipGetLastError();
		free_b = 0;
	}
	if (rows == 0 || bytes > env_u64("TOKSCORE_INDEX_MB", 32768) * (1ull << 20) || bytes > free_b / 2) return;
	int32_t *d_slot_row = nullptr;
	uint32_t *d_counts = nullptr;
	bool ok = hipMalloc((void **)&c->d_index, bytes) == hipSuccess &&
	          hipMalloc((void **)&d_slot_row, slot_row.size() * sizeof(int32_t)) == hipSuccess &&
	          hipMalloc((void **)&d_counts, rows * sizeof(uint32_t)) == hipSuccess;
	if (ok) {
		c->row_chunks.assign(rows, 0);
		TrieDev T = trie_dev(c);
		CorpusDev C{c->d_bytes, c->d_off, c->D, c->total};
		ok = hipMemsetAsync(c->d_index, 0, bytes, c->stream) == hipSuccess &&
		     hipMemcpyAsync(d_slot_row, slot_row.data(), slot_row.size() * sizeof(int32_t), hipMemcpyHostToDevice,
		                    c->stream) == hipSuccess;
		if (ok) {
			launch_index_all(T, C, d_slot_row, c->d_chunk_begin, c->nchunks, c->chunk_bytes, rows, c->d_index, d_counts,
			                 c->stream);
			ok = hipMemcpyAsync(c->row_chunks.data(), d_counts, rows * sizeof(uint32_t), hipMemcpyDeviceToHost,
			                    c->stream) == hipSuccess &&
			     hipStreamSynchronize(c->stream) == hipSuccess;
		}
	}
	if (d_slot_row) (void)hipFree(d_slot_row);
	if (d_counts) (void)hipFree(d_counts);
	if (ok) {
		c->index_state = 1;
	} else {
		(void)hipGetLastError();
		if (c->d_index) (void)hipFree(c->d_index);
		c->d_index = nullptr;
	}
}

int check_overflow(tokscore_ctx *c) { // after fetch_out
	if (c->scan_path < 1 && !c->dfa_used_last) return TOKSCORE_OK;
	if (c->h_out[c->V + 5] >> 32) {
		c->stream_ready = 0; // the word is cleared by the prepare launch, which a "ready" stream skips: not sticky
		return fail(c, TOKSCORE_E_CAPACITY, "token stream overflow (internal capacity bound violated)");
	}
	return TOKSCORE_OK;
}

// One batch of (token, index row, stream-image slot) for the search of the streams the tokens'
// strings occur in (launch_find_toggle_streams): the description goes to the device in one blob,
//   delta[n] (u64, zero) | items per list [kToggleLists] (zero) | rows[n] | pair_off[n+1] | slots[n] | str_off[n+1] | codes
// and the scratch buffers grow to what the batch can need.  dedupe: every stream listed once
// whichever tokens occur in it (incremental re-score); otherwise once per token (independent moves).
struct ToggleSpec {
	uint64_t token;
	uint32_t row, slot, chunks;
};
struct ToggleUpload {
	ToggleBatch B;
	uint64_t pairs = 0, list_cap = 0;
	const uint32_t *d_slots = nullptr;
	unsigned long long *d_delta = nullptr;
};
// list_cap_hint: entries per item list to provide when fewer than the worst case are expected (0: the
// worst case, which cannot overflow; otherwise the caller checks the counts and comes again)
int upload_toggle_batch(tokscore_ctx *c, const ToggleSpec *m, uint32_t n, bool dedupe, hipStream_t s, ToggleUpload *up,
                        uint64_t list_cap_hint = 0) {
	const uint64_t per_chunk = c->widest_chunk; // no chunk has more streams (plan_stream)
	uint64_t pairs = 0;
	for (uint32_t k = 0; k < n; k++) pairs += m[k].chunks;
	if (pairs > 0xFFFFFFF0ull / per_chunk / 2) return fail(c, TOKSCORE_E_CAPACITY, "a toggled token touches too many streams for one launch");
	// items: kToggleLists lists; list q takes what the workgroups q, q + kToggleLists, .. of the search find
	// (16 pairs per workgroup), so this many entries per list always suffice
	const uint64_t find_blocks = (pairs + 15) / 16;
	uint64_t list_cap = std::max<uint64_t>(1, (find_blocks + kToggleLists - 1) / kToggleLists * 16 * per_chunk);
	if (list_cap_hint) list_cap = std::min(list_cap, list_cap_hint);
	const uint64_t cap_items = list_cap * kToggleLists;
	std::vector<uint32_t> meta((size_t)2 * n + kToggleLists, 0);
	std::vector<uint8_t> codes;
	const size_t rows_at = meta.size();
	for (uint32_t k = 0; k < n; k++) meta.push_back(m[k].row);
	const size_t poff_at = meta.size();
	uint32_t acc = 0;
	for (uint32_t k = 0; k < n; k++) {
		meta.push_back(acc);
		acc += m[k].chunks;
	}
	meta.push_back(acc);
	const size_t slots_at = meta.size();
	for (uint32_t k = 0; k < n; k++) meta.push_back(m[k].slot);
	const size_t soff_at = meta.size();
	for (uint32_t k = 0; k < n; k++) {
		meta.push_back((uint32_t)codes.size());
		for (uint64_t i = c->h_vocab_off[m[k].token]; i < c->h_vocab_off[m[k].token + 1]; i++) codes.push_back(c->simg.code[c->h_vocab_bytes[i]]);
	}
	meta.push_back((uint32_t)codes.size());
	while (meta.size() % 4) meta.push_back(0); // the move table is read as 16-byte records
	const size_t moves_at = meta.size();
	for (uint32_t k = 0; k < n; k++) {
		const uint32_t at = meta[soff_at + k], len = meta[soff_at + k + 1] - at;
		uint32_t pat = 0;
		for (uint32_t i = 0; i < 4 && i < len; i++) pat |= (uint32_t)codes[at + i] << (8 * i);
		meta.push_back(at);
		meta.push_back(len);
		meta.push_back(pat);
		meta.push_back(len >= 4 ? 0xFFFFFFFFu : (1u << (8 * len)) - 1u);
	}
	const size_t codes_at = meta.size() * sizeof(uint32_t);
	const size_t blob = codes_at + ((codes.size() + 15) & ~(size_t)15) + 16;
	auto grow = [&](void **p, size_t *cap, size_t need, size_t elem) -> hipError_t {
		if (need <= *cap) return hipSuccess;
		if (*p) (void)hipFree(*p);
		*p = nullptr, *cap = 0;
		const hipError_t e = hipMalloc(p, need * elem);
		if (e == hipSuccess) *cap = need;
		return e;
	};
	HIP_TRY(c, grow((void **)&c->d_tg_meta, &c->tg_meta_cap, blob * 2, 1));
	HIP_TRY(c, grow((void **)&c->d_tg_pairs, &c->tg_pairs_cap, std::max<uint64_t>(1, pairs), sizeof(uint4)));
	HIP_TRY(c, grow((void **)&c->d_tg_items, &c->tg_items_cap, cap_items, sizeof(ToggleItem)));
	const size_t seen_words = ((size_t)c->nstreams + 31) / 32 + 1;
	if (dedupe) {
		HIP_TRY(c, grow((void **)&c->d_tg_seen, &c->tg_seen_cap, seen_words, sizeof(uint32_t)));
		HIP_TRY(c, hipMemsetAsync(c->d_tg_seen, 0, seen_words * sizeof(uint32_t), s));
	}
	// one copy from a pinned buffer (every caller waits for the stream before the next batch is built)
	if (blob > c->h_tg_meta_cap) {
		if (c->h_tg_meta) (void)hipHostFree(c->h_tg_meta);
		c->h_tg_meta = nullptr, c->h_tg_meta_cap = 0;
		HIP_TRY(c, hipHostMalloc((void **)&c->h_tg_meta, blob * 2, hipHostMallocDefault));
		c->h_tg_meta_cap = blob * 2;
	}
	memcpy(c->h_tg_meta, meta.data(), codes_at);
	if (!codes.empty()) memcpy(c->h_tg_meta + codes_at, codes.data(), codes.size());
	HIP_TRY(c, hipMemcpyAsync(c->d_tg_meta, c->h_tg_meta, codes_at + codes.size(), hipMemcpyHostToDevice, s));
	uint32_t *words = (uint32_t *)c->d_tg_meta;
	ToggleBatch &B = up->B;
	B.n = n;
	B.index = c->d_index;
	B.words_per_row = (c->nchunks + 31) / 32;
	B.rows = words + rows_at;
	B.pair_off = words + poff_at;
	B.str_off = words + soff_at;
	B.moves = reinterpret_cast<const uint4 *>(words + moves_at);
	B.chunk_first = nullptr, B.pbeg = nullptr; // (the launcher fills them in from its StreamWork)
	B.codes = c->d_tg_meta + codes_at;
	B.pairs = c->d_tg_pairs;
	B.items = c->d_tg_items;
	B.n_items = words + 2 * n;
	B.cap_items = (uint32_t)list_cap;
	B.seen = dedupe ? c->d_tg_seen : nullptr;
	up->pairs = pairs;
	up->list_cap = list_cap;
	up->d_slots = words + slots_at;
	up->d_delta = (unsigned long long *)c->d_tg_meta;
	return TOKSCORE_OK;
}

// tokscore_count_occurrences on the stream image: where every position's walk ended -> per-slot
// totals -> (host) every string's subtree sum.
int count_on_stream_image(tokscore_ctx *c, uint32_t max_cp, uint64_t *counts, uint64_t *positions) {
	hipStream_t s = c->stream;
	const uint32_t nslots = (uint32_t)c->simg.lt.size();
	if (!c->d_tprep_raw) {
		// the prepared corpus again, bytes as they are (this path does not lower-case), straight
		// into the transposed layout; the linear copy is only needed on the way
		uint8_t *d_raw = nullptr;
		uint32_t *d_flags = nullptr;
		const uint64_t nblocks = ((uint64_t)c->nranges + 63) / 64;
		const uint64_t tbytes = nblocks * ((uint64_t)c->range_bytes + c->halo) * 64;
		hipError_t e = hipMalloc((void **)&d_raw, c->prep_total + kCorpusPad);
		if (e == hipSuccess) e = hipMalloc((void **)&d_flags, sizeof(uint32_t));
		if (e == hipSuccess) e = hipMalloc((void **)&c->d_tprep_raw, tbytes);
		if (e == hipSuccess) e = hipMemsetAsync(d_raw + (c->prep_total & ~15ull), 0, kCorpusPad + (c->prep_total & 15), s);
		if (e == hipSuccess) e = hipMemsetAsync(d_flags, 0, sizeof(uint32_t), s);
		if (e == hipSuccess) {
			CorpusDev C{c->d_bytes, c->d_off, c->D, c->total};
			ByteCodes M;
			memcpy(M.code, c->simg.code, 256);
			launch_prep_corpus(C, M, d_raw, c->prep_total, d_flags, s, /*lower=*/false);
			launch_transpose_ranges(d_raw, c->prep_total + kCorpusPad, c->range_bytes, c->halo, (uint32_t)nblocks, c->d_tprep_raw, s);
			e = hipStreamSynchronize(s);
		}
		if (d_raw) (void)hipFree(d_raw);
		if (d_flags) (void)hipFree(d_flags);
		if (e != hipSuccess) {
			if (c->d_tprep_raw) (void)hipFree(c->d_tprep_raw);
			c->d_tprep_raw = nullptr;
			return fail(c, TOKSCORE_E_HIP, std::string("count_occurrences: ") + hipGetErrorString(e));
		}
	}
	if ((size_t)nslots + 2 > c->slot_ends_cap) {
		if (c->d_slot_ends) (void)hipFree(c->d_slot_ends);
		c->d_slot_ends = nullptr, c->slot_ends_cap = 0;
		HIP_TRY(c, hipMalloc((void **)&c->d_slot_ends, ((size_t)nslots + 2) * sizeof(unsigned long long)));
		c->slot_ends_cap = (size_t)nslots + 2;
	}
	StreamDev Q{c->d_stream, c->stream_cap_batches, c->d_fill, c->d_cursor};
	HIP_TRY(c, hipMemsetAsync(c->d_slot_ends, 0, ((size_t)nslots + 2) * sizeof(unsigned long long), s));
	c->stream_ready = 0;
	launch_prepare(nullptr, (uint32_t)c->V, c->D, c->total, Q, s, c->rplan.blocks * c->rplan.waves_per_block);
	StreamWork W = stream_work(c, false);
	W.tprep = c->d_tprep_raw;
	// events as for an evaluation (tokscore_last_timing: [1] = the counting kernel, [3] = histogram + reduction)
	hipEvent_t *ev = nullptr;
	if (c->timing_on) {
		ev = c->ring[c->calls % tokscore_ctx::kRing];
		hipEvent_t *view = c->view[c->calls % tokscore_ctx::kRing];
		c->calls++;
		view[0] = view[1] = ev[1];
		view[2] = view[3] = ev[2];
		view[4] = ev[4];
		c->ev = view;
		HIP_TRY(c, hipEventRecord(ev[1], s));
	}
	launch_count_ends(stream_trie_dev(c), W, Q, c->rplan, c->simg.first_cont_code, max_cp, c->d_slot_ends + nslots + 1, s);
	if (ev) HIP_TRY(c, hipEventRecord(ev[2], s));
	launch_histogram_slots(Q, nslots, nullptr, c->d_slot_ends, nslots, c->d_slab, c->slab_blocks, s);
	if (ev) HIP_TRY(c, hipEventRecord(ev[4], s));
	c->timing_pending = ev != nullptr;
	HIP_TRY(c, hipGetLastError());
	std::vector<uint64_t> ends((size_t)nslots + 2);
	HIP_TRY(c, hipMemcpyAsync(ends.data(), c->d_slot_ends, ends.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
	HIP_TRY(c, hipStreamSynchronize(s));
	c->trie.AccumulateEnds(ends.data(), max_cp, counts);
	if (positions) *positions = ends[(size_t)nslots + 1];
	return TOKSCORE_OK;
}

} // namespace

extern "C" {

int tokscore_abi_version(void) { return 100; }

int tokscore_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

const char *tokscore_strerror(int status) {
	switch (status) {
	case TOKSCORE_OK: return "ok";
	case TOKSCORE_E_ARG: return "invalid argument";
	case TOKSCORE_E_NO_DEVICE: return "no HIP device (libtokscore has no CPU fallback)";
	case TOKSCORE_E_HIP: return "HIP runtime error";
	case TOKSCORE_E_STATE: return "vocabulary or corpus not set";
	case TOKSCORE_E_VOCAB: return "unusable vocabulary";
	case TOKSCORE_E_CAPACITY: return "capacity exceeded";
	case TOKSCORE_E_INTERNAL: return "internal error";
	}
	return "unknown status";
}

// why the last tokscore_create on this thread failed (there is no context to hold it)
static thread_local std::string t_create_error;

const char *tokscore_last_error(const tokscore_ctx *ctx) { return ctx ? ctx->err.c_str() : t_create_error.c_str(); }

int tokscore_create(tokscore_ctx **out, int device) {
	if (!out) return TOKSCORE_E_ARG;
	*out = nullptr;
	int n = tokscore_device_count();
	if (n <= 0 || device < 0 || device >= n) return TOKSCORE_E_NO_DEVICE;
	if (hipSetDevice(device) != hipSuccess) return TOKSCORE_E_NO_DEVICE;
	tokscore_ctx *c = new tokscore_ctx();
	c->device = device;
	t_create_error.clear();
	auto give_up = [&](const char *what, hipError_t e) {
		t_create_error = std::string(what) + ": " + hipGetErrorString(e);
		delete c;
		return TOKSCORE_E_HIP;
	};
	hipError_t e = static_scratch_address(device, &c->static_scratch);
	if (e != hipSuccess) return give_up("address of the library's static device scratch (is the gfx950 code object loaded?)", e);
	c->knob_noincr = env_u64("TOKSCORE_NOINCR", 0) != 0;
	c->knob_nobatch = env_u64("TOKSCORE_NOBATCH", 0) != 0;
	c->knob_search = env_u64("TOKSCORE_SEARCH", 1) != 0;
	c->knob_search_whole = (uint32_t)std::min<uint64_t>(kToggleOffsets, std::max<uint64_t>(1, env_u64("TOKSCORE_SEARCH_WHOLE", kToggleOffsets)));
	c->knob_search_moves = std::min<uint64_t>(0xFFFF, std::max<uint64_t>(1, env_u64("TOKSCORE_SEARCH_MOVES", 256)));
	c->sr_slots_fixed = getenv("TOKSCORE_SEARCH_SLOTS") != nullptr;
	c->sr_slots_hint = std::max<uint64_t>(1024, env_u64("TOKSCORE_SEARCH_SLOTS", 1u << 16));
	c->knob_no_mark_ranges = env_u64("TOKSCORE_NOMARKRANGES", 0) != 0;
	c->knob_incr_div = getenv("TOKSCORE_INCR_DIV") ? std::max<uint64_t>(1, env_u64("TOKSCORE_INCR_DIV", 3)) : 0;
	c->knob_toggle_items = std::max<uint64_t>(1, env_u64("TOKSCORE_TOGGLE_ITEMS", 32ull << 20));
	c->tg_list_hint = std::max<uint64_t>(1, env_u64("TOKSCORE_TOGGLE_LIST", 8192));
	c->dfa_enabled = env_u64("TOKSCORE_DFA", 1) != 0;
	c->dfa_max_slots = std::max<uint64_t>(1024, env_u64("TOKSCORE_DFA_SLOTS", 6u << 20));
	c->dfa_chains = (uint32_t)env_u64("TOKSCORE_DFA_CHAINS", 0);
	c->dfa_lds_cap = (uint32_t)env_u64("TOKSCORE_DFA_LDS", 0);
	c->dfa_relayout_on = env_u64("TOKSCORE_DFA_RELAYOUT", 1) != 0;
	c->dfa_relayouts_max = (uint32_t)env_u64("TOKSCORE_DFA_RELAYOUTS", kDfaRelayoutsDefault);
	c->fix_epoch = (uint32_t)(env_u64("TOKSCORE_FIX_EPOCH_START", 0) % 65536u); // (tests: start next to the 16-bit tag's wrap)
	c->fix_epoch_wrapped = c->fix_epoch != 0;
	c->dfa_max_cold_ppm = env_u64("TOKSCORE_DFA_MAX_COLD_PPM", c->dfa_max_cold_ppm);
	c->dfa_max_virtual_ppm = env_u64("TOKSCORE_DFA_MAX_VIRTUAL_PPM", c->dfa_max_virtual_ppm);
	c->dfa_fail_max_cold_ppm = env_u64("TOKSCORE_DFA_FAIL_MAX_COLD_PPM", c->dfa_fail_max_cold_ppm);
	c->dfa_form_knob = (uint32_t)env_u64("TOKSCORE_DFA_FORM", 0);
	c->dfa_warm_bytes = (uint32_t)std::min<uint64_t>(4096, env_u64("TOKSCORE_DFA_WARM", c->dfa_warm_bytes));
	c->timing_on = env_u64("TOKSCORE_TIMING", 1) != 0;
	c->timing_detail = env_u64("TOKSCORE_TIMING", 1) >= 2;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cus = prop.multiProcessorCount;
	if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return give_up("stream", e);
	if ((e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking)) != hipSuccess ||
	    (e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming)) != hipSuccess ||
	    (e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming)) != hipSuccess)
		return give_up("stream / events", e);
	for (auto &set : c->ring)
		for (auto &ev : set)
			if ((e = hipEventCreate(&ev)) != hipSuccess) return give_up("timing events", e);
	*out = c;
	return TOKSCORE_OK;
}

int tokscore_create_hostonly(tokscore_ctx **out) {
	if (!out) return TOKSCORE_E_ARG;
	tokscore_ctx *c = new tokscore_ctx();
	c->hostonly = true;
	*out = c;
	return TOKSCORE_OK;
}

void tokscore_destroy(tokscore_ctx *c) {
	if (!c) return;
	if (!c->hostonly) {
		(void)hipSetDevice(c->device);
		(void)hipStreamSynchronize(c->stream);
		if (c->side) (void)hipStreamSynchronize(c->side);
		free_dfa(c);
		free_plan(c);
		free_corpus(c);
		if (c->d_slots) (void)hipFree(c->d_slots);
		if (c->d_vocab_bytes) (void)hipFree(c->d_vocab_bytes);
		if (c->d_slt) (void)hipFree(c->d_slt);
		if (c->d_sbase) (void)hipFree(c->d_sbase);
		if (c->d_slot_id) (void)hipFree(c->d_slot_id);
		if (c->d_out) (void)hipFree(c->d_out);
		if (c->h_out) (void)hipHostFree(c->h_out);
		if (c->d_tg_meta) (void)hipFree(c->d_tg_meta);
		if (c->d_tg_pairs) (void)hipFree(c->d_tg_pairs);
		if (c->d_tg_items) (void)hipFree(c->d_tg_items);
		if (c->d_tg_seen) (void)hipFree(c->d_tg_seen);
		if (c->d_sr_table) (void)hipFree(c->d_sr_table);
		if (c->h_tg_meta) (void)hipHostFree(c->h_tg_meta);
		if (c->d_stream_tokens) (void)hipFree(c->d_stream_tokens);
		if (c->d_tok_starts) (void)hipFree(c->d_tok_starts);
		if (c->d_slot_ends) (void)hipFree(c->d_slot_ends);
		if (c->h_patch) (void)hipHostFree(c->h_patch);
		for (auto &set : c->ring)
			for (auto &e : set)
				if (e) (void)hipEventDestroy(e);
		if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
		if (c->ev_join) (void)hipEventDestroy(c->ev_join);
		if (c->side) (void)hipStreamDestroy(c->side);
		if (c->stream) (void)hipStreamDestroy(c->stream);
	}
	delete c;
}

int tokscore_set_vocab(tokscore_ctx *c, const uint8_t *bytes, const uint64_t *off, uint64_t V) {
	if (!c || !off || (!bytes && V && off[V] != off[0])) return fail(c, TOKSCORE_E_ARG, "null vocabulary buffers");
	if (V + 5 >= 0xFFFFFFFFull) return fail(c, TOKSCORE_E_VOCAB, "vocabulary too large");
	static const uint8_t kEmpty = 0;
	std::string e;
	c->vocab_set = false;
	c->packed_valid = false;
	c->device_trie_valid = false;
	c->plan_valid = false;
	c->dfa_invalid = true;
	c->pending.clear();
	c->pending_s.clear();
	c->simg_valid = false;
	c->state_valid = false;
	c->d_vocab_valid = false;
	if (!c->trie.Build(bytes ? bytes : &kEmpty, off, V, &e)) return fail(c, TOKSCORE_E_VOCAB, e);
	c->h_vocab_off.resize(V + 1);
	for (uint64_t i = 0; i <= V; i++) c->h_vocab_off[i] = off[i] - off[0];
	c->h_vocab_bytes.assign(bytes ? bytes + off[0] : &kEmpty, bytes ? bytes + off[V] : &kEmpty);
	c->V = V;
	c->vocab_set = true;
	c->sample_used = false;
	return TOKSCORE_OK;
}

int tokscore_vocab_max_len(const tokscore_ctx *c, uint64_t *max_len) {
	if (!c || !max_len) return TOKSCORE_E_ARG;
	if (!c->vocab_set) return TOKSCORE_E_STATE;
	*max_len = c->trie.max_len();
	return TOKSCORE_OK;
}

int tokscore_toggle(tokscore_ctx *c, uint64_t token_index, int enable) {
	if (!c) return TOKSCORE_E_ARG;
	if (!c->vocab_set) return fail(c, TOKSCORE_E_STATE, "vocabulary not set");
	if (token_index >= c->V) return fail(c, TOKSCORE_E_ARG, "token index out of range");
	if (!c->packed_valid) {
		int st = ensure_trie(c);
		if (st) return st;
	}
	uint64_t entry = 0;
	const int64_t slot = c->trie.Toggle(token_index, enable != 0, &c->packed, &entry);
	const int64_t sslot = c->simg_valid ? c->trie.PatchStream(token_index, &c->simg) : -1;
	if (sslot >= 0 && !c->hostonly && c->device_trie_valid) c->pending_s.push_back(sslot);
	// The device copy is patched by the next evaluation (apply_pending): an incremental
	// re-score first takes back what the OLD trie counted in the chunks the token occurs in.
	if (slot >= 0 && !c->hostonly && c->device_trie_valid) c->pending.emplace_back(token_index, slot);
	if (slot >= 0 || sslot >= 0) {
		c->toggle_version++;
		c->dfa_stale_evals = 0;
	}
	return TOKSCORE_OK;
}

static int adopt_offsets(tokscore_ctx *c, const uint64_t *off, uint64_t D) {
	c->h_off.resize(D + 1);
	const uint64_t o0 = off[0];
	for (uint64_t d = 0; d <= D; d++) {
		if (d && off[d] < off[d - 1]) return fail(c, TOKSCORE_E_ARG, "document offsets are not monotone");
		c->h_off[d] = off[d] - o0;
	}
	c->D = D;
	c->total = c->h_off[D];
	return TOKSCORE_OK;
}

int tokscore_set_corpus(tokscore_ctx *c, const uint8_t *bytes, const uint64_t *off, uint64_t D) {
	if (!c || !off) return fail(c, TOKSCORE_E_ARG, "null corpus buffers");
	if (c->hostonly) return fail(c, TOKSCORE_E_NO_DEVICE, tokscore_strerror(TOKSCORE_E_NO_DEVICE));
	(void)hipSetDevice(c->device);
	(void)hipDeviceSynchronize(); // (an evaluation on a caller's stream may still read what is freed here)
	free_dfa(c);
	free_plan(c);
	free_corpus(c);
	c->state_valid = false;
	int st = adopt_offsets(c, off, D);
	if (st) return st;
	if (c->total && !bytes) return fail(c, TOKSCORE_E_ARG, "null corpus bytes");
	const uint8_t *src = bytes ? bytes + off[0] : nullptr;
	HIP_TRY(c, hipMalloc((void **)&c->d_bytes, c->total + kCorpusPad));
	c->own_corpus = true;
	HIP_TRY(c, hipMalloc((void **)&c->d_off, (D + 1) * sizeof(uint64_t)));
	HIP_TRY(c, hipMemsetAsync(c->d_bytes + c->total, 0, kCorpusPad, c->stream));
	if (c->total) HIP_TRY(c, hipMemcpyAsync(c->d_bytes, src, c->total, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(c, hipMemcpyAsync(c->d_off, c->h_off.data(), (D + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
	// strided host sample that steers the hot-first trie layout (never the results)
	c->sample.clear();
	if (c->total) {
		const uint64_t piece = std::min<uint64_t>(kSampleBytes / kSamplePieces, c->total);
		const uint64_t pieces = std::min<uint64_t>(kSamplePieces, std::max<uint64_t>(1, c->total / piece));
		for (uint64_t p = 0; p < pieces; p++) {
			const uint64_t at = (c->total - piece) / std::max<uint64_t>(1, pieces) * p;
			c->sample.insert(c->sample.end(), src + at, src + at + piece);
		}
	}
	c->sample_used = false;
	c->force_generic = false;
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	c->corpus_set = true;
	return TOKSCORE_OK;
}

int tokscore_set_corpus_device(tokscore_ctx *c, const void *d_bytes, const void *d_off, uint64_t D,
                               uint64_t total_bytes) {
	if (!c || !d_off || (!d_bytes && total_bytes)) return fail(c, TOKSCORE_E_ARG, "null corpus buffers");
	if (c->hostonly) return fail(c, TOKSCORE_E_NO_DEVICE, tokscore_strerror(TOKSCORE_E_NO_DEVICE));
	(void)hipSetDevice(c->device);
	(void)hipDeviceSynchronize(); // (an evaluation on a caller's stream may still read what is freed here)
	free_dfa(c);
	free_plan(c);
	free_corpus(c);
	c->state_valid = false;
	std::vector<uint64_t> off(D + 1);
	HIP_TRY(c, hipMemcpy(off.data(), d_off, (D + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
	if (off[0] != 0 || off[D] != total_bytes) return fail(c, TOKSCORE_E_ARG, "device offsets must run from 0 to total_bytes");
	int st = adopt_offsets(c, off.data(), D);
	if (st) return st;
	c->d_bytes = (uint8_t *)d_bytes;
	c->d_off = (uint64_t *)d_off;
	c->own_corpus = false;
	c->sample.clear();
	if (c->total) {
		const uint64_t piece = std::min<uint64_t>(kSampleBytes / kSamplePieces, c->total);
		const uint64_t pieces = std::min<uint64_t>(kSamplePieces, std::max<uint64_t>(1, c->total / piece));
		c->sample.resize(piece * pieces);
		for (uint64_t p = 0; p < pieces; p++) {
			const uint64_t at = (c->total - piece) / std::max<uint64_t>(1, pieces) * p;
			HIP_TRY(c, hipMemcpy(c->sample.data() + p * piece, (const uint8_t *)d_bytes + at, piece, hipMemcpyDeviceToHost));
		}
	}
	c->sample_used = false;
	c->corpus_set = true;
	return TOKSCORE_OK;
}

int tokscore_score_async(tokscore_ctx *c, void *d_out, void *stream) {
	if (!c || !d_out) return fail(c, TOKSCORE_E_ARG, "null output");
	if (c->hostonly) return fail(c, TOKSCORE_E_NO_DEVICE, tokscore_strerror(TOKSCORE_E_NO_DEVICE));
	(void)hipSetDevice(c->device);
	return enqueue_score(c, (unsigned long long *)d_out, stream ? (hipStream_t)stream : c->stream);
}

int tokscore_async_status(tokscore_ctx *c, void *stream, int *overflowed) {
	if (!c || !overflowed) return fail(c, TOKSCORE_E_ARG, "null output");
	if (c->hostonly) return fail(c, TOKSCORE_E_NO_DEVICE, tokscore_strerror(TOKSCORE_E_NO_DEVICE));
	(void)hipSetDevice(c->device);
	hipStream_t s = stream ? (hipStream_t)stream : c->stream;
	uint32_t words[2] = {0, 0};
	const uint32_t *cursor = c->dfa_used_last ? c->d_dcursor : (c->plan_valid && c->scan_path >= 1 ? c->d_cursor : nullptr);
	if (cursor) HIP_TRY(c, hipMemcpyAsync(words, cursor, sizeof words, hipMemcpyDeviceToHost, s));
	HIP_TRY(c, hipStreamSynchronize(s));
	*overflowed = words[1] != 0;
	if (words[1] != 0) { // reported once: the next evaluation starts from a prepared stream (or, the automaton's, from the plan's scan)
		c->stream_ready = 0;
		if (c->dfa_used_last) {
			free_dfa(c);
			c->dfa_state = -1;
		}
	}
	return TOKSCORE_OK;
}

int tokscore_score(tokscore_ctx *c, uint64_t *counts, uint64_t *total_tokens, uint64_t *total_bytes) {
	if (!c || !counts) return fail(c, TOKSCORE_E_ARG, "null output");
	if (c->hostonly) return fail(c, TOKSCORE_E_NO_DEVICE, tokscore_strerror(TOKSCORE_E_NO_DEVICE));
	(void)hipSetDevice(c->device);
	int st = ensure_out(c);
	if (st) return st;
	HostTrace tr;
	st = enqueue_score(c, c->d_out, c->stream);
	if (st) return st;
	tr.mark("tokscore_score: enqueued");
	st = fetch_out(c, c->stream);
	if (st) return st;
	tr.mark("tokscore_score: result on the host");
	const uint64_t *host = c->h_out;
	st = check_overflow(c);
	if (st == TOKSCORE_E_CAPACITY && !c->force_generic) {
		// adversarial input blew the stream bound: redo this evaluation on the generic path
		c->force_generic = true;
		c->plan_valid = false;
		c->dfa_invalid = true;
		return tokscore_score(c, counts, total_tokens, total_bytes);
	}
	if (st) return st;
	c->state_valid = true; // d_out = this evaluation, the device trie = the vocabulary it used
	c->rescored_chunks = 0xFFFFFFFFu;
	memcpy(counts, host, (c->V + 3) * sizeof(uint64_t));
	if (total_tokens) *total_tokens = host[c->V + 3];
	if (total_bytes) *total_bytes = host[c->V + 4];
	return TOKSCORE_OK;
}

int tokscore_score_incremental(tokscore_ctx *c, uint64_t *counts, uint64_t *total_tokens, uint64_t *total_bytes) {
	// Re-score after toggles.  Chunk boundaries are token starts under every subset of enabled
	// tokens, so only the chunks that hold an occurrence of a toggled token's string can change:
	// those are scanned twice -- with the trie as it was (their old contribution, subtracted) and,
	// after the pending slot patches, as it is (added) -- on top of the counts the last
	// evaluation left in d_out.  Same result as tokscore_score, bit for bit; anything that does
	// not fit the fast path (no previous evaluation, fallback scan path, speculated boundaries,
	// many toggles, a token that occurs in more than a third of the chunks) is a plain full evaluation.
	if (!c || !counts) return fail(c, TOKSCORE_E_ARG, "null output");
	if (c->hostonly) return fail(c, TOKSCORE_E_NO_DEVICE, tokscore_strerror(TOKSCORE_E_NO_DEVICE));
	(void)hipSetDevice(c->device);
	if (!c->plan_valid && c->state_valid && c->corpus_set && c->vocab_set && !c->hostonly) {
		// full evaluations on the automaton scan need no plan: this call does (chunk boundaries, the occurrence index)
		int pst = ensure_trie(c);
		if (!pst) pst = ensure_plan(c);
		if (pst) return pst;
	}
	const bool eligible = c->state_valid && c->plan_valid && c->packed_valid && c->device_trie_valid && c->d_vocab_valid &&
	                      c->scan_path >= 1 && (c->scan_path == 2 ? c->chunks_synced : c->unsynced == 0) && c->uniform_chunks && c->pending.size() <= kMaxPatch &&
	                      !c->knob_noincr;
	if (!eligible) return tokscore_score(c, counts, total_tokens, total_bytes);
	hipStream_t s = c->stream;
	uint32_t nsel = 0;
	if (!c->pending.empty()) {
		TrieDev T = trie_dev(c);
		CorpusDev C{c->d_bytes, c->d_off, c->D, c->total};
		ChunkState S{c->d_chunk_begin, c->d_chunk_state, c->d_chunk_state + (c->nchunks + 1),
		             c->d_chunk_state + 2 * (size_t)(c->nchunks + 1), c->nchunks};
		StreamDev Q{c->d_stream, c->stream_cap_batches, c->d_fill, c->d_cursor};
		// events: start and end; with TOKSCORE_TIMING=2 also the subtract pass, the slot patch and
		// the add pass (every record holds the stream for about 5 us)
		hipEvent_t *ev = nullptr;
		if (c->timing_on) {
			ev = c->ring[c->calls % tokscore_ctx::kRing];
			hipEvent_t *view = c->view[c->calls % tokscore_ctx::kRing];
			for (int k = 0; k < 5; k++) view[k] = c->timing_detail ? ev[k] : ev[k == 4 ? 4 : 0];
			c->calls++;
			c->ev = view;
		}
		bool listed = false; // d_sel_list holds the selected chunks
		bool streams_listed = false; // by_stream holds the streams to scan (no chunk list)
		ToggleUpload by_stream;
		auto mark = [&](int k) -> hipError_t {
			return ev && (c->timing_detail || k == 0 || k == 4) ? hipEventRecord(ev[k], s) : hipSuccess;
		};
		HIP_TRY(c, mark(0));
		if (c->index_state == 0) build_index(c);
		// Two partial passes (about 0.11 ms each before the first stream is walked, then the longest
		// selected stream decides, which grows with their number) against one full evaluation (about
		// 0.05 ms + 0.95 ms per GB): on a 100 MB shard the full evaluation always wins, on a 1 GB
		// shard a token in up to a thousand chunks is re-scored faster.  TOKSCORE_INCR_DIV = d: the
		// plain rule "at most 1/d of the chunks" instead (tests).
		uint32_t limit;
		if (c->knob_incr_div) {
			limit = c->nchunks / (uint32_t)c->knob_incr_div;
		} else if (c->scan_path != 2) {
			// the tile scan (vocabularies the stream image cannot hold: config 5's CJK-heavy one) takes
			// 3 ms per GB for a full evaluation and 0.1 ms for a partial pass over a few thousand
			// chunks: round 1's rule, a third of the chunks.  (The estimate below, made for the stream
			// scan, sent 72 % of config 5's moves to full evaluations: 443 instead of 2,400 moves/s.)
			limit = c->nchunks / 3;
		} else {
			const double full_us = 50.0 + 0.95e-6 * (double)c->total; // 0.95 us per MB
			const double fixed_us = 2 * 110.0, per_chunk_us = 2 * 0.3;
			limit = full_us > fixed_us ? (uint32_t)std::min<double>((full_us - fixed_us) / per_chunk_us, c->nchunks / 3.0) : 0u;
		}
		if (c->index_state == 1) {
			// the index knows the chunks (and how many) without a search or a host sync
			uint32_t rows[8], nrows = 0;
			for (size_t k = 0; k < c->pending.size(); k++) {
				const int32_t r = c->tok_row[c->pending[k].first];
				bool seen = r < 0;
				for (uint32_t m = 0; m < nrows; m++) seen = seen || rows[m] == (uint32_t)r;
				if (!seen) {
					rows[nrows++] = (uint32_t)r;
					nsel += c->row_chunks[r]; // an upper bound when rows overlap
				}
			}
			if (nsel > limit) {
				if (ev) c->calls--; // this call's event set goes to the full evaluation
				return tokscore_score(c, counts, total_tokens, total_bytes);
			}
			if (nsel && c->scan_path == 2) {
				// the streams the toggled strings occur in (a chunk is sixteen of them): found once,
				// scanned by both passes
				ToggleSpec spec[8];
				uint32_t nspec = 0;
				for (size_t k = 0; k < c->pending.size() && nspec < nrows; k++) {
					const int32_t r = c->tok_row[c->pending[k].first];
					bool seen = r < 0;
					for (uint32_t m = 0; m < nspec; m++) seen = seen || spec[m].row == (uint32_t)r;
					if (!seen) spec[nspec++] = ToggleSpec{c->pending[k].first, (uint32_t)r, 0u, c->row_chunks[r]};
				}
				int st = upload_toggle_batch(c, spec, nspec, true, s, &by_stream);
				if (st == TOKSCORE_E_CAPACITY) return tokscore_score(c, counts, total_tokens, total_bytes); // too many streams for one launch: the full evaluation
				if (st) return st;
				StreamWork Wf = stream_work(c, false);
				launch_find_toggle_streams(by_stream.B, Wf, (uint32_t)by_stream.pairs, c->nchunks, s);
				streams_listed = true;
			} else if (nsel) launch_or_rows(c->d_index, rows, nrows, c->nchunks, c->d_sel, s);
		} else {
			HIP_TRY(c, hipMemsetAsync(c->d_sel, 0, ((size_t)(c->nchunks + 31) / 32 + 1) * sizeof(uint32_t), s));
			for (size_t k = 0; k < c->pending.size(); k++) {
				const uint64_t tok = c->pending[k].first;
				bool seen = false;
				for (size_t m = 0; m < k; m++) seen = seen || c->pending[m].first == tok;
				const uint64_t len = c->h_vocab_off[tok + 1] - c->h_vocab_off[tok];
				if (seen || len == 0) continue;
				launch_find_token_chunks(C, c->d_chunk_begin, c->nchunks, c->chunk_bytes,
				                         c->d_vocab_bytes + c->h_vocab_off[tok], (uint32_t)len, c->d_sel, s);
			}
			launch_compact_sel(c->d_sel, c->nchunks, c->d_sel_list, s);
			listed = true;
			HIP_TRY(c, hipMemcpyAsync(&nsel, c->d_sel_list + c->nchunks, sizeof nsel, hipMemcpyDeviceToHost, s));
			HIP_TRY(c, hipStreamSynchronize(s));
			if (nsel > limit) {
				if (ev) c->calls--;
				return tokscore_score(c, counts, total_tokens, total_bytes);
			}
		}
		// the scan hands out no more stream batches than it takes chunks, and a wave of the
		// histogram takes whole batches (any grid is correct): no more histogram workgroups, and
		// slab rows to add up, than that needs
		if (nsel && !listed && !streams_listed) launch_compact_sel(c->d_sel, c->nchunks, c->d_sel_list, s);
		const uint32_t hist_blocks = std::max<uint32_t>(1, std::min<uint32_t>(c->slab_blocks, (nsel + 15) / 16));
		HIP_TRY(c, mark(1));
		// one partial pass: the selected chunks scanned under the device trie as it is now
		auto partial_pass = [&](bool flip) {
			c->stream_ready = 0;
			launch_prepare(nullptr, (uint32_t)c->V, c->D, c->total, Q, s);
			if (c->scan_path == 2) {
				StreamWork Wp = stream_work(c, !streams_listed);
				if (streams_listed) {
					Wp.tg_items = by_stream.B.items;
					Wp.tg_count = by_stream.B.n_items;
					Wp.tg_cap = by_stream.B.cap_items;
				}
				launch_scan_stream(stream_trie_dev(c), Wp, Q, c->splan, s);
				launch_histogram_slots(Q, (uint32_t)c->simg.lt.size(), c->d_slot_id, c->d_out, (uint32_t)c->V + 3, c->d_slab,
				                       hist_blocks, s, flip);
			} else {
				launch_scan_tile(T, C, S, Q, c->tile, c->d_chunk_first_doc, c->packed.max_len, s, c->d_sel_list,
				                 c->d_sel_list + c->nchunks);
				launch_histogram(Q, c->tile, (uint32_t)c->V, c->d_out, c->d_slab, hist_blocks, s, flip);
			}
		};
		if (nsel) partial_pass(/*flip=*/true);
		HIP_TRY(c, mark(2));
		c->state_valid = false; // until the second half is in
		int st = apply_pending(c, s);
		if (st) return st;
		HIP_TRY(c, mark(3));
		if (nsel) partial_pass(/*flip=*/false);
		HIP_TRY(c, mark(4));
		HIP_TRY(c, hipGetLastError());
		c->timing_pending = ev != nullptr;
	}
	int st = fetch_out(c, s);
	if (st) return st;
	const uint64_t *host = c->h_out;
	if (nsel) {
		// (two passes share the overflow flag: prepare clears it, so this sees the second one;
		// a partial scan holds at most a third of what a full one is sized for)
		st = check_overflow(c);
		if (st) {
			c->force_generic = true;
			c->plan_valid = false;
			c->dfa_invalid = true;
			return tokscore_score(c, counts, total_tokens, total_bytes);
		}
	}
	c->state_valid = true;
	c->rescored_chunks = nsel;
	memcpy(counts, host, (c->V + 3) * sizeof(uint64_t));
	if (total_tokens) *total_tokens = host[c->V + 3];
	if (total_bytes) *total_bytes = host[c->V + 4];
	return TOKSCORE_OK;
}

// tokscore_score_toggles when the batched path does not apply: every move as toggle, incremental
// re-score, toggle back (the same answers, one evaluation per move)
static int score_toggles_one_by_one(tokscore_ctx *c, const uint64_t *token_index, const int32_t *enable, uint64_t K,
                                    int64_t *delta) {
	std::vector<uint64_t> counts(c->V + 3);
	uint64_t base = 0, tt = 0;
	int st = tokscore_score_incremental(c, counts.data(), &base, nullptr);
	if (st) return st;
	bool moved = false;
	for (uint64_t k = 0; k < K; k++) {
		delta[k] = 0;
		const bool en = enable[k] != 0;
		if (c->trie.Enabled(token_index[k]) == en) continue;
		if ((st = tokscore_toggle(c, token_index[k], en))) return st;
		st = tokscore_score_incremental(c, counts.data(), &tt, nullptr);
		const int st2 = tokscore_toggle(c, token_index[k], !en);
		if (st || st2) return st ? st : st2;
		delta[k] = (int64_t)(tt - base);
		moved = true;
	}
	if (moved) st = tokscore_score_incremental(c, counts.data(), &tt, nullptr); // the state as it was
	return st;
}

// One batch of tokscore_score_toggles by ONE pass over the corpus: the moves' strings as a search automaton
// (dfa.h: BuildSearchImage) on the scan's loop and transposed corpus, its occurrences on token starts gathered
// per (move, stream), then the walks of the other path.  1: not this way (no automaton plan for this corpus, the
// batch's automaton does not fit LDS): the caller takes the occurrence index.  deltas -> res[0 .. n).
int search_batch(tokscore_ctx *c, const ToggleSpec *m, uint32_t n, hipStream_t s, int64_t *deltas);
// (the pass over the corpus costs the same for any number of strings, the automaton must fit LDS: 256 moves at a time,
// fewer where their automaton is too large)
int score_toggles_by_search(tokscore_ctx *c, const ToggleSpec *m, uint32_t n, hipStream_t s, int64_t *deltas) {
	uint32_t step = (uint32_t)std::max<uint64_t>(1, c->knob_search_moves);
	for (uint32_t first = 0; first < n;) {
		const uint32_t k = std::min(step, n - first);
		const int st = search_batch(c, m + first, k, s, deltas + first);
		if (st == 1 && first != 0 && k > 8) { // (the plan was there for the batches before: this one's automaton does not fit)
			step = k / 2;
			continue;
		}
		if (st) return st;
		first += k;
	}
	return TOKSCORE_OK;
}

// the stream at every 256th prepared byte (once per stream plan)
int ensure_block_stream(tokscore_ctx *c, hipStream_t s) {
	if (c->d_block_stream) return TOKSCORE_OK;
	HIP_TRY(c, hipMalloc((void **)&c->d_block_stream, ((size_t)c->prep_total / 256 + 2) * sizeof(uint32_t)));
	launch_stream_blocks(c->d_pbeg, c->nstreams, (uint32_t)c->prep_total, c->d_block_stream, s);
	return TOKSCORE_OK;
}

int search_batch(tokscore_ctx *c, const ToggleSpec *m, uint32_t n, hipStream_t s, int64_t *deltas) {
	if (!c->knob_search || n == 0 || n >= 0x10000u) return 1;
	if (c->dfa_invalid) { // (as enqueue_score: the corpus or the trie changed since the automaton's plan was made)
		(void)hipDeviceSynchronize();
		free_dfa(c);
		c->dfa_invalid = false;
	}
	ensure_dfa(c);
	if (c->dfa_state != 1 || !c->d_dtcodes || !c->d_dstream_a) return 1;
	// the strings in the automaton's codes
	std::vector<uint8_t> codes;
	std::vector<uint32_t> off(n + 1, 0);
	for (uint32_t k = 0; k < n; k++) {
		for (uint64_t i = c->h_vocab_off[m[k].token]; i < c->h_vocab_off[m[k].token + 1]; i++) codes.push_back(c->dimg.code[c->h_vocab_bytes[i]]);
		off[k + 1] = (uint32_t)codes.size();
	}
	const bool trace = getenv("TOKSCORE_TRACE_HOST") != nullptr;
	auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double t0 = now();
	SearchImage img;
	if (!BuildSearchImage(codes.data(), off.data(), n, c->dimg.ncodes, dfa_lds_slots_max(0), &img)) return 1;
	const double t_img = now();
	auto grow = [&](void **p, size_t *cap, size_t need) -> hipError_t {
		if (need <= *cap) return hipSuccess;
		if (*p) (void)hipFree(*p);
		*p = nullptr, *cap = 0;
		const hipError_t e = hipMalloc(p, need);
		if (e == hipSuccess) *cap = need;
		return e;
	};
	{
		const int st = ensure_block_stream(c, s);
		if (st) return st;
	}
	const DfaScanPlan &P = c->dplan;
	const uint32_t first_batches = P.blocks * P.waves_per_block;
	if (c->dstream_cap < first_batches + 16 || img.max_len > P.range_bytes || img.max_len > 256) return 1; // (a string crosses one range boundary at most; a record says in 8 bits how far behind it)
	// blob (32-bit words): delta u64[n] | flags[2] | cursor[2] | list counts[kToggleLists] | slots[n] | hot | out_first | out_list
	const size_t hot_words = ((size_t)img.hot_slots + 19) & ~(size_t)3; // (16 bytes of padding the kernel's 16-byte copies may read)
	const size_t flags_at = 2 * (size_t)n, cursor_at = flags_at + 4, lists_at = cursor_at + 2, slots_at = lists_at + kToggleLists;
	const size_t hot_at = (slots_at + n + 3) & ~(size_t)3, first_at = hot_at + hot_words, list_at = first_at + hot_words;
	const size_t words = list_at + img.out_list.size() + 4;
	for (;;) {
		uint64_t slots = 1024;
		// (a move's string is the token taken in about a thousand of a 100 MB shard's streams: four slots per item and more)
		const uint64_t want = c->sr_slots_fixed ? c->sr_slots_hint : std::max<uint64_t>(c->sr_slots_hint, (uint64_t)n * (c->prep_total / 32768 + 1024));
		while (slots < want) slots <<= 1;
		if (slots > (1ull << 28)) return 1;
		const uint32_t list_cap = (uint32_t)(slots / 32); // (items <= slots / 2 in a table that is not full, spread evenly over the lists: four times their share)
		const size_t table_bytes = slots * (sizeof(unsigned long long) + sizeof(uint32_t) + sizeof(ToggleItem)) + (size_t)kToggleLists * list_cap * sizeof(uint32_t);
		HIP_TRY(c, grow((void **)&c->d_sr_table, &c->sr_table_bytes, table_bytes));
		HIP_TRY(c, grow((void **)&c->d_tg_meta, &c->tg_meta_cap, words * sizeof(uint32_t)));
		if (words * sizeof(uint32_t) > c->h_tg_meta_cap) {
			if (c->h_tg_meta) (void)hipHostFree(c->h_tg_meta);
			c->h_tg_meta = nullptr, c->h_tg_meta_cap = 0;
			HIP_TRY(c, hipHostMalloc((void **)&c->h_tg_meta, words * sizeof(uint32_t) * 2, hipHostMallocDefault));
			c->h_tg_meta_cap = words * sizeof(uint32_t) * 2;
		}
		uint32_t *h = reinterpret_cast<uint32_t *>(c->h_tg_meta);
		memset(h, 0, hot_at * sizeof(uint32_t));
		h[cursor_at] = first_batches;
		for (uint32_t k = 0; k < n; k++) h[slots_at + k] = m[k].slot;
		memcpy(h + hot_at, img.hot.data(), img.hot_slots * sizeof(uint32_t));
		for (size_t i = img.hot_slots; i < hot_words; i++) h[hot_at + i] = DfaImage::kEmptyWord;
		memcpy(h + first_at, img.out_first.data(), img.hot_slots * sizeof(uint32_t));
		for (size_t i = img.hot_slots; i < hot_words; i++) h[first_at + i] = SearchImage::kNone;
		if (!img.out_list.empty()) memcpy(h + list_at, img.out_list.data(), img.out_list.size() * sizeof(uint32_t));
		HIP_TRY(c, hipMemcpyAsync(c->d_tg_meta, h, words * sizeof(uint32_t), hipMemcpyHostToDevice, s));
		uint32_t *d = reinterpret_cast<uint32_t *>(c->d_tg_meta);
		unsigned long long *keys = reinterpret_cast<unsigned long long *>(c->d_sr_table);
		uint32_t *count = reinterpret_cast<uint32_t *>(keys + slots);
		ToggleItem *items = reinterpret_cast<ToggleItem *>(count + slots);
		HIP_TRY(c, hipMemsetAsync(c->d_sr_table, 0, slots * (sizeof(unsigned long long) + sizeof(uint32_t)), s));
		SearchDev A{d + hot_at, d + first_at, d + list_at, img.hot_slots, img.stride, img.max_len};
		DfaWork Wd = dfa_work(c);
		const StreamDev Q{c->d_dstream_a, c->dstream_cap, c->d_dfill_a, d + cursor_at};
		uint32_t *list_index = reinterpret_cast<uint32_t *>(items + slots);
		SearchItems H{keys, count, items, (uint32_t)(slots - 1), d + flags_at, list_index, d + lists_at, list_cap, c->d_tok_starts, c->d_pbeg, c->d_block_stream, c->nstreams, (uint32_t)c->prep_total};
		launch_dfa_search(A, Wd, Q, c->d_dcodes, P, s);
		launch_search_items(A, Q, H, (uint32_t)c->num_cus, s);
		StreamWork W = stream_work(c, false);
		W.tg_items = items;
		W.tg_count = d + lists_at;
		W.tg_cap = list_cap;
		W.tg_index = list_index;
		W.tg_n = count;
		W.tg_moves = n;
		W.tg_whole = c->knob_search_whole;
		W.tg_known = getenv("TOKSCORE_LOCAL_BOTH") ? nullptr : c->d_tok_starts;
		W.tg_slot = d + slots_at;
		W.tg_base = c->d_stream_tokens;
		W.tg_delta = reinterpret_cast<unsigned long long *>(d);
		launch_local_toggles(stream_trie_dev(c), W, (uint32_t)c->num_cus, (uint64_t)list_cap * kToggleLists, s);
		launch_walk_toggles(stream_trie_dev(c), W, c->splan, (uint64_t)list_cap * kToggleLists, s);
		HIP_TRY(c, hipGetLastError());
		HIP_TRY(c, hipMemcpyAsync(c->h_tg_meta, c->d_tg_meta, slots_at * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
		const double t_l = now();
		HIP_TRY(c, hipStreamSynchronize(s));
		if (trace)
			fprintf(stderr, "score_toggles by search: automaton %.0f us (%u states, %u entries, %u slots), launches %.0f, wait %.0f; table %llu slots, records %u\n",
			        t_img - t0, img.states, img.entries, img.hot_slots, t_l - t_img, now() - t_l, (unsigned long long)slots, h[flags_at + 2]);
		uint64_t made = 0;
		uint32_t most = 0;
		for (uint32_t q = 0; q < kToggleLists; q++) made += h[lists_at + q], most = std::max(most, h[lists_at + q]);
		const bool full = h[flags_at] != 0 || most > list_cap;
		const uint32_t overflow = h[cursor_at + 1];
		if (overflow) return fail(c, TOKSCORE_E_CAPACITY, "search records overflow (internal capacity bound violated)");
		if (made * 4 > slots) c->sr_slots_hint = std::max<uint64_t>(c->sr_slots_hint, made * 8); // (the next batch)
		if (trace) fprintf(stderr, "score_toggles by search: %llu items, longest list %u of %u%s\n", (unsigned long long)made, most, list_cap, full ? ", again with a larger table" : "");
		if (full) {
			c->sr_slots_hint = std::max<uint64_t>(2 * slots, made * 8);
			continue;
		}
		const uint64_t *res = reinterpret_cast<const uint64_t *>(c->h_tg_meta);
		for (uint32_t k = 0; k < n; k++) deltas[k] = (int64_t)res[k];
		c->toggle_items += made;
		c->search_batches++;
		return TOKSCORE_OK;
	}
}

int tokscore_score_toggles(tokscore_ctx *c, const uint64_t *token_index, const int32_t *enable, uint64_t K,
                           int64_t *delta_total_tokens) {
	if (!c || (K && (!token_index || !enable || !delta_total_tokens))) return fail(c, TOKSCORE_E_ARG, "null buffers");
	if (c->hostonly) return fail(c, TOKSCORE_E_NO_DEVICE, tokscore_strerror(TOKSCORE_E_NO_DEVICE));
	if (!c->vocab_set) return fail(c, TOKSCORE_E_STATE, "vocabulary not set");
	for (uint64_t k = 0; k < K; k++)
		if (token_index[k] >= c->V) return fail(c, TOKSCORE_E_ARG, "token index out of range");
	if (K == 0) return TOKSCORE_OK;
	(void)hipSetDevice(c->device);
	int st = ensure_trie(c);
	if (st) return st;
	if (!c->corpus_set) return fail(c, TOKSCORE_E_STATE, "corpus not set");
	st = ensure_plan(c);
	if (st) return st;
	bool batched = c->scan_path == 2 && c->chunks_synced && c->uniform_chunks && c->simg_valid &&
	               !c->knob_noincr && !c->knob_nobatch;
	// (the occurrence index is only built when a batch cannot be searched in one pass: score_toggles_by_search)
	bool by_search = batched && c->knob_search && c->dfa_enabled && c->dfa_state != -1;
	if (batched && !by_search && c->index_state == 0) build_index(c);
	batched = batched && (by_search || c->index_state == 1);
	if (!batched) return score_toggles_one_by_one(c, token_index, enable, K, delta_total_tokens);
	hipStream_t s = c->stream;
	if (!c->pending.empty() || !c->pending_s.empty()) {
		// toggles not evaluated yet: the moves are judged against the vocabulary as it is now
		std::vector<uint64_t> counts(c->V + 3);
		st = tokscore_score_incremental(c, counts.data(), nullptr, nullptr);
		if (st) return st;
	}
	const bool trace = getenv("TOKSCORE_TRACE_HOST") != nullptr;
	auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double t_begin = now();
	std::vector<ToggleSpec> moves;
	std::vector<uint64_t> move_k;
	for (uint64_t k = 0; k < K; k++) {
		delta_total_tokens[k] = 0;
		bool flips = false;
		const int64_t t = c->trie.PeekStream(token_index[k], enable[k] != 0, c->simg, &flips);
		if (t < 0 || !flips) continue; // nothing can change
		moves.push_back(ToggleSpec{token_index[k], 0u, (uint32_t)t, 0u});
		move_k.push_back(k);
	}
	if (moves.empty()) return TOKSCORE_OK;
	auto with_index_rows = [&]() { // the moves whose strings the corpus holds, with their rows of the occurrence index
		size_t kept = 0;
		for (size_t i = 0; i < moves.size(); i++) {
			const int32_t row = c->tok_row[moves[i].token];
			if (row < 0 || c->row_chunks[row] == 0) continue;
			moves[kept] = ToggleSpec{moves[i].token, (uint32_t)row, moves[i].slot, c->row_chunks[row]};
			move_k[kept++] = move_k[i];
		}
		moves.resize(kept), move_k.resize(kept);
	};
	if (!by_search) with_index_rows();
	if (moves.empty()) return TOKSCORE_OK;
	const double t_peek = now();
	// tokens per stream under the vocabulary as it is: one walk over all streams, kept until the
	// device trie changes
	if (c->stream_tokens_epoch != c->trie_epoch) {
		if ((size_t)c->nstreams + 1 > c->stream_tokens_cap) {
			if (c->d_stream_tokens) (void)hipFree(c->d_stream_tokens);
			c->d_stream_tokens = nullptr, c->stream_tokens_cap = 0;
			HIP_TRY(c, hipMalloc((void **)&c->d_stream_tokens, ((size_t)c->nstreams + 1) * sizeof(uint32_t)));
			c->stream_tokens_cap = (size_t)c->nstreams + 1;
		}
		const size_t start_words = (size_t)(c->prep_total / 64 + 2);
		if (start_words > c->tok_starts_cap) {
			if (c->d_tok_starts) (void)hipFree(c->d_tok_starts);
			c->d_tok_starts = nullptr, c->tok_starts_cap = 0;
			HIP_TRY(c, hipMalloc((void **)&c->d_tok_starts, start_words * sizeof(unsigned long long)));
			c->tok_starts_cap = start_words;
		}
		HIP_TRY(c, hipMemsetAsync(c->d_tok_starts, 0, start_words * sizeof(unsigned long long), s));
		StreamWork Wb = stream_work(c, false);
		Wb.tg_base = c->d_stream_tokens;
		Wb.tg_starts = c->d_tok_starts;
		Wb.tg_order = c->d_stream_order;
		if (c->d_tprep && c->entries_epoch == c->trie_epoch && !c->knob_no_mark_ranges) {
			// a full evaluation of this vocabulary has just left every range's true entry: walk the ranges
			HIP_TRY(c, hipMemsetAsync(c->d_stream_tokens, 0, ((size_t)c->nstreams + 1) * sizeof(uint32_t), s));
			st = ensure_block_stream(c, s);
			if (st) return st;
			Wb.block_stream = c->d_block_stream;
			launch_mark_ranges(stream_trie_dev(c), Wb, c->rplan, s);
		} else launch_count_streams(stream_trie_dev(c), Wb, c->splan, s);
		HIP_TRY(c, hipGetLastError());
		c->stream_tokens_epoch = c->trie_epoch;
	}
	if (by_search) {
		std::vector<int64_t> d(moves.size());
		st = score_toggles_by_search(c, moves.data(), (uint32_t)moves.size(), s, d.data());
		if (st == TOKSCORE_OK) {
			for (size_t i = 0; i < moves.size(); i++) delta_total_tokens[move_k[i]] = d[i];
			if (trace) fprintf(stderr, "score_toggles: one-pass search, %zu moves, %.0f us\n", moves.size(), now() - t_begin);
			return TOKSCORE_OK;
		}
		if (st != 1) return st;
		// not this way: the occurrence index
		if (c->index_state == 0) build_index(c);
		if (c->index_state != 1) return score_toggles_one_by_one(c, token_index, enable, K, delta_total_tokens);
		with_index_rows();
		if (moves.empty()) return TOKSCORE_OK;
	}
	// (one launch addresses at most 2^32 - 16 (pair, stream) items: a batch never asks for more than that)
	const uint64_t budget = std::max<uint64_t>(1, std::min<uint64_t>(c->knob_toggle_items, 0xFFFFFFF0ull / std::max<uint64_t>(1, c->widest_chunk) / 2));
	const StreamWork W0 = stream_work(c, false);
	size_t first = 0;
	while (first < moves.size()) {
		// a batch: as many moves as the item budget holds (one at least)
		size_t last = first;
		uint64_t pairs = 0;
		while (last < moves.size() && (last == first || pairs + moves[last].chunks <= budget)) pairs += moves[last++].chunks;
		const uint32_t n = (uint32_t)(last - first);
		ToggleUpload up;
		const double t_up0 = now();
		st = upload_toggle_batch(c, moves.data() + first, n, false, s, &up, c->tg_list_hint);
		if (st == TOKSCORE_E_CAPACITY) // a single move whose token occurs in more streams than one launch addresses: move by move, as the header promises
			return score_toggles_one_by_one(c, token_index, enable, K, delta_total_tokens);
		if (st) return st;
		const double t_up1 = now();
		up.B.starts = c->d_tok_starts;
		StreamWork W = W0;
		W.tg_items = up.B.items;
		W.tg_count = up.B.n_items;
		W.tg_cap = up.B.cap_items;
		W.tg_slot = up.d_slots;
		W.tg_base = c->d_stream_tokens;
		W.tg_delta = up.d_delta;
		launch_find_toggle_streams(up.B, W, (uint32_t)up.pairs, c->nchunks, s);
		launch_local_toggles(stream_trie_dev(c), W, (uint32_t)c->num_cus, up.list_cap * kToggleLists, s);
		launch_walk_toggles(stream_trie_dev(c), W, c->splan, up.list_cap * kToggleLists, s); // (the few items without a list)
		HIP_TRY(c, hipGetLastError());
		// (results come back into the pinned buffer the description left from: the upload is done by then)
		const uint64_t *res = reinterpret_cast<const uint64_t *>(c->h_tg_meta);
		HIP_TRY(c, hipMemcpyAsync(c->h_tg_meta, c->d_tg_meta, ((size_t)n + kToggleLists / 2) * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
		const double t_l = now();
		HIP_TRY(c, hipStreamSynchronize(s));
		if (trace) fprintf(stderr, "score_toggles: peek %.0f us, base %.0f, upload %.0f, launches %.0f, wait %.0f\n", t_peek - t_begin, t_up0 - t_peek, t_up1 - t_up0, t_l - t_up1, now() - t_l);
		const uint32_t *per_list = reinterpret_cast<const uint32_t *>(res + n);
		uint32_t most = 0;
		for (uint32_t q = 0; q < kToggleLists; q++) most = std::max(most, per_list[q]);
		if (most > up.B.cap_items) {
			// the lists were too short for this batch (what did not fit was counted, not stored):
			// the same batch again with room for it
			c->tg_list_hint = 2ull * most;
			continue;
		}
		for (uint32_t q = 0; q < kToggleLists; q++) c->toggle_items += per_list[q];
		for (uint32_t m = 0; m < n; m++) delta_total_tokens[move_k[first + m]] = (int64_t)res[m];
		first = last;
	}
	return TOKSCORE_OK;
}

int tokscore_count_occurrences(tokscore_ctx *c, uint32_t max_cp, uint64_t *counts, uint64_t *positions) {
	if (!c || !counts) return fail(c, TOKSCORE_E_ARG, "null output");
	if (max_cp == 0) return fail(c, TOKSCORE_E_ARG, "max_cp must be at least 1");
	if (c->hostonly) return fail(c, TOKSCORE_E_NO_DEVICE, tokscore_strerror(TOKSCORE_E_NO_DEVICE));
	(void)hipSetDevice(c->device);
	int st = ensure_trie(c);
	if (st) return st;
	if (!c->corpus_set) return fail(c, TOKSCORE_E_STATE, "corpus not set");
	st = ensure_out(c);
	if (st) return st;
	st = apply_pending(c, c->stream);
	if (st) return st;
	c->state_valid = false; // d_out is reused below
	// On the stream image (kernels_stream.hip: count_ends_kernel) when the plan has one with
	// transposed ranges and every token fits the walk; TOKSCORE_COUNT=tile keeps the older kernel.
	{
		const char *force = getenv("TOKSCORE_COUNT");
		if (!(force && !strcmp(force, "tile")) && c->total != 0) {
			st = ensure_plan(c);
			if (st) return st;
			if (c->scan_path == 2 && c->d_tprep && c->simg_valid) return count_on_stream_image(c, max_cp, counts, positions);
		}
	}
	const uint32_t n_ids = (uint32_t)c->V + 3;
	TrieDev T = trie_dev(c);
	CorpusDev C{c->d_bytes, c->d_off, c->D, c->total};
	CountPlan plan;
	if (!plan_count(T, c->packed.max_len, max_cp, n_ids, &plan))
		return fail(c, TOKSCORE_E_CAPACITY, "tokens too long for the counting kernel's LDS tile");
	const uint32_t tiles = count_tiles(c->total);
	const uint32_t blocks = count_blocks(c->total, (uint32_t)c->num_cus);
	// a slab cell is an int32 partial of one workgroup; tiles are handed out dynamically, so
	// leave room for a workgroup that gets four times its share
	if (c->total / blocks >= 0x7FFFFFFFull / 4) return fail(c, TOKSCORE_E_CAPACITY, "corpus too large for one counting launch");
	static const uint32_t walks = [] {
		const char *e = getenv("TOKSCORE_COUNT_WALKS");
		return e && atoi(e) == 4 ? 4u : 2u; // 2: 1.82 ms, 4: 1.90 ms on the 100 MB bench shard
	}();
	uint32_t *d_slab = nullptr, *d_tiles = nullptr;
	uint64_t *d_first = nullptr;
	HIP_TRY(c, hipMalloc((void **)&d_slab, (size_t)blocks * (n_ids + 1) * sizeof(uint32_t)));
	hipError_t e = hipMalloc((void **)&d_tiles, count_head_words(plan) * sizeof(uint32_t));
	if (e == hipSuccess) e = hipMalloc((void **)&d_first, std::max<size_t>(1, tiles) * sizeof(uint64_t));
	std::vector<uint64_t> host(c->V + 5);
	do {
		if (e != hipSuccess) break;
		if ((e = hipMemsetAsync(c->d_out, 0, (c->V + 5) * sizeof(unsigned long long), c->stream)) != hipSuccess) break;
		if ((e = launch_count_all(T, C, plan, max_cp, n_ids, blocks, walks, d_tiles, d_first, d_slab, c->stream)) != hipSuccess) break;
		launch_slab_reduce(d_slab, blocks, n_ids + 1, c->d_out, c->stream);
		if ((e = hipGetLastError()) != hipSuccess) break;
		if ((e = hipMemcpyAsync(host.data(), c->d_out, host.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream)) != hipSuccess) break;
		e = hipStreamSynchronize(c->stream);
	} while (0);
	if (d_first) (void)hipFree(d_first);
	(void)hipFree(d_slab);
	if (d_tiles) (void)hipFree(d_tiles);
	if (e != hipSuccess) return fail(c, TOKSCORE_E_HIP, std::string("count_occurrences: ") + hipGetErrorString(e));
	memcpy(counts, host.data(), (c->V + 3) * sizeof(uint64_t));
	if (positions) *positions = host[c->V + 3];
	return TOKSCORE_OK;
}

int tokscore_count_occurrences_static(tokscore_ctx *c, uint32_t max_cp, void *d_out, void *stream) {
	if (!c || !d_out) return fail(c, TOKSCORE_E_ARG, "null output");
	if (max_cp == 0) return fail(c, TOKSCORE_E_ARG, "max_cp must be at least 1");
	if (c->hostonly) return fail(c, TOKSCORE_E_NO_DEVICE, tokscore_strerror(TOKSCORE_E_NO_DEVICE));
	(void)hipSetDevice(c->device);
	if (!c->vocab_set || !c->corpus_set) return fail(c, TOKSCORE_E_STATE, "vocabulary or corpus not set");
	// bringing the trie to the device allocates: that is set-up, done by any earlier evaluation
	if (!c->packed_valid || !c->device_trie_valid)
		return fail(c, TOKSCORE_E_STATE,
		            "the trie is not in device memory yet: run tokscore_count_occurrences once "
		            "before the first tokscore_count_occurrences_static");
	hipStream_t s = stream ? (hipStream_t)stream : c->stream;
	// Toggles since the last evaluation are patched into the device trie by the next evaluation, on the context's
	// stream, from staging memory that evaluation owns.  This call must not do that on the caller's stream (not
	// ordered against the context's, a pageable copy that cannot be captured) nor behind the back of an
	// incremental re-score, which would find nothing pending and return the counts of before the toggle.
	if (!c->pending.empty() || !c->pending_s.empty())
		return fail(c, TOKSCORE_E_STATE,
		            "toggles are pending: run an evaluation (tokscore_score* or tokscore_count_occurrences) before "
		            "tokscore_count_occurrences_static, which never changes the context's state");
	const uint32_t n_ids = (uint32_t)c->V + 3;
	TrieDev T = trie_dev(c);
	CorpusDev C{c->d_bytes, c->d_off, c->D, c->total};
	CountPlan plan;
	if (!plan_count(T, c->packed.max_len, max_cp, n_ids, &plan))
		return fail(c, TOKSCORE_E_CAPACITY, "tokens too long for the counting kernel's LDS tile");
	// the static array, cut up: queue heads | first document of every tile | one row of partial
	// counts per workgroup -- as many workgroups as fit, at most what the allocating call takes
	const auto up256 = [](size_t n) { return (n + 255) & ~(size_t)255; };
	const size_t tiles = count_tiles(c->total);
	const size_t heads_bytes = up256(count_head_words(plan) * sizeof(uint32_t));
	const size_t first_bytes = up256(std::max<size_t>(1, tiles) * sizeof(uint64_t));
	const size_t row_bytes = ((size_t)n_ids + 1) * sizeof(uint32_t);
	// (TOKSCORE_STATIC_LIMIT: use less of the array, for the tests of the two refusals below)
	const size_t limit = (size_t)std::min<uint64_t>(kStaticScratchBytes, env_u64("TOKSCORE_STATIC_LIMIT", kStaticScratchBytes));
	char msg[256];
	if (heads_bytes + first_bytes + row_bytes > limit) {
		snprintf(msg, sizeof msg,
		         "count_occurrences_static: %zu bytes of scratch needed for %llu corpus bytes and %llu vocabulary entries, "
		         "the static buffer holds %zu; use tokscore_count_occurrences",
		         heads_bytes + first_bytes + row_bytes, (unsigned long long)c->total, (unsigned long long)c->V, limit);
		return fail(c, TOKSCORE_E_CAPACITY, msg);
	}
	const uint32_t blocks = (uint32_t)std::min<size_t>(count_blocks(c->total, (uint32_t)c->num_cus),
	                                                  (limit - heads_bytes - first_bytes) / row_bytes);
	if (c->total / blocks >= 0x7FFFFFFFull / 4) {
		snprintf(msg, sizeof msg, "count_occurrences_static: %llu corpus bytes over the %u workgroups the static buffer has "
		                          "room for overflow a 32-bit partial count; use tokscore_count_occurrences",
		         (unsigned long long)c->total, blocks);
		return fail(c, TOKSCORE_E_CAPACITY, msg);
	}
	uint32_t *d_heads = (uint32_t *)c->static_scratch;
	uint64_t *d_first = (uint64_t *)(c->static_scratch + heads_bytes);
	uint32_t *d_slab = (uint32_t *)(c->static_scratch + heads_bytes + first_bytes);
	// nothing here relies on what the previous call left behind: launch_count_all sets the heads,
	// tile_first and every slab cell of the `blocks` rows are written before they are read
	hipError_t e = hipMemsetAsync(d_out, 0, ((size_t)c->V + 4) * sizeof(unsigned long long), s);
	if (e == hipSuccess) e = launch_count_all(T, C, plan, max_cp, n_ids, blocks, 2, d_heads, d_first, d_slab, s);
	if (e == hipSuccess) {
		launch_slab_reduce(d_slab, blocks, n_ids + 1, (unsigned long long *)d_out, s);
		e = hipGetLastError();
	}
	if (e != hipSuccess) return fail(c, TOKSCORE_E_HIP, std::string("count_occurrences_static: ") + hipGetErrorString(e));
	return TOKSCORE_OK;
}

int tokscore_extract_candidates(tokscore_ctx *c, uint32_t max_cp, uint64_t *n_tokens, uint64_t *positions) {
	if (!c || !n_tokens) return fail(c, TOKSCORE_E_ARG, "null output");
	if (c->hostonly) return fail(c, TOKSCORE_E_NO_DEVICE, tokscore_strerror(TOKSCORE_E_NO_DEVICE));
	if (!c->corpus_set) return fail(c, TOKSCORE_E_STATE, "corpus not set");
	(void)hipSetDevice(c->device);
	free_candidates(&c->cands);
	CorpusDev C{c->d_bytes, c->d_off, c->D, c->total};
	std::string err;
	const int rc = extract_candidates(C, max_cp, &c->cands, &err, c->stream);
	if (rc == 1) return fail(c, TOKSCORE_E_HIP, err);
	if (rc == 2) return fail(c, TOKSCORE_E_CAPACITY, err);
	*n_tokens = c->cands.n;
	if (positions) *positions = c->cands.positions;
	return TOKSCORE_OK;
}

int tokscore_candidates_copy(tokscore_ctx *c, uint32_t *chr, uint64_t *uses, uint32_t *lpar, uint32_t *rpar) {
	if (!c) return fail(c, TOKSCORE_E_ARG, "null context");
	if (c->hostonly) return fail(c, TOKSCORE_E_NO_DEVICE, tokscore_strerror(TOKSCORE_E_NO_DEVICE));
	(void)hipSetDevice(c->device);
	const uint64_t n = c->cands.n;
	hipError_t e = hipSuccess;
	if (n) {
		if (chr) e = hipMemcpyAsync(chr, c->cands.chr, n * sizeof *chr, hipMemcpyDeviceToHost, c->stream);
		if (e == hipSuccess && uses) e = hipMemcpyAsync(uses, c->cands.uses, n * sizeof *uses, hipMemcpyDeviceToHost, c->stream);
		if (e == hipSuccess && lpar) e = hipMemcpyAsync(lpar, c->cands.lpar, n * sizeof *lpar, hipMemcpyDeviceToHost, c->stream);
		if (e == hipSuccess && rpar) e = hipMemcpyAsync(rpar, c->cands.rpar, n * sizeof *rpar, hipMemcpyDeviceToHost, c->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	}
	free_candidates(&c->cands);
	if (e != hipSuccess) return fail(c, TOKSCORE_E_HIP, std::string("candidates_copy: ") + hipGetErrorString(e));
	return TOKSCORE_OK;
}

int tokscore_probe_read_bandwidth(tokscore_ctx *c, uint64_t bytes, uint32_t reps, double *gb_per_s) {
	if (!c || !gb_per_s || reps == 0) return fail(c, TOKSCORE_E_ARG, "null output or zero repetitions");
	if (c->hostonly) return fail(c, TOKSCORE_E_NO_DEVICE, tokscore_strerror(TOKSCORE_E_NO_DEVICE));
	bytes &= ~(uint64_t)4095;
	if (bytes < (1u << 20)) return fail(c, TOKSCORE_E_ARG, "probe buffer below 1 MiB");
	(void)hipSetDevice(c->device);
	uint8_t *buf = nullptr;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	float ms = 0.f;
	hipError_t e = hipMalloc((void **)&buf, bytes + 256);
	do {
		if (e != hipSuccess) break;
		if ((e = hipEventCreate(&e0)) != hipSuccess || (e = hipEventCreate(&e1)) != hipSuccess) break;
		if ((e = hipMemsetAsync(buf, 0, bytes + 256, c->stream)) != hipSuccess) break;
		uint32_t *sink = (uint32_t *)(buf + bytes);
		const uint32_t blocks = (uint32_t)std::min<uint64_t>((uint64_t)c->num_cus * 16, (bytes / 16 + 255) / 256);
		launch_read_probe(buf, bytes, blocks, sink, c->stream); // warm-up (page tables, clocks)
		if ((e = hipEventRecord(e0, c->stream)) != hipSuccess) break;
		for (uint32_t r = 0; r < reps; r++) launch_read_probe(buf, bytes, blocks, sink, c->stream);
		if ((e = hipEventRecord(e1, c->stream)) != hipSuccess) break;
		if ((e = hipEventSynchronize(e1)) != hipSuccess) break;
		if ((e = hipEventElapsedTime(&ms, e0, e1)) != hipSuccess) break;
		e = hipGetLastError();
	} while (0);
	if (e0) (void)hipEventDestroy(e0);
	if (e1) (void)hipEventDestroy(e1);
	if (buf) (void)hipFree(buf);
	if (e != hipSuccess) return fail(c, TOKSCORE_E_HIP, std::string("probe_read_bandwidth: ") + hipGetErrorString(e));
	*gb_per_s = ms > 0.f ? (double)bytes * reps / (ms * 1e-3) / 1e9 : 0.0;
	return TOKSCORE_OK;
}

int tokscore_tokenize(tokscore_ctx *c, const uint8_t *text, uint64_t n, uint64_t *ids_out, uint64_t cap,
                      uint64_t *n_ids) {
	if (!c || !ids_out || !n_ids || (!text && n)) return fail(c, TOKSCORE_E_ARG, "null buffers");
	if (c->hostonly) return fail(c, TOKSCORE_E_NO_DEVICE, tokscore_strerror(TOKSCORE_E_NO_DEVICE));
	(void)hipSetDevice(c->device);
	int st = ensure_trie(c);
	if (st) return st;
	if (!c->pending.empty()) {
		st = apply_pending(c, c->stream);
		if (st) return st;
		c->state_valid = false; // the device trie moved on without the counts
	}
	uint8_t *d_text = nullptr;
	unsigned long long *d_ids = nullptr;
	const uint64_t need = n + 2;
	HIP_TRY(c, hipMalloc((void **)&d_text, n + 16));
	hipError_t e = hipMalloc((void **)&d_ids, (need + 1) * sizeof(unsigned long long));
	if (e != hipSuccess) {
		(void)hipFree(d_text);
		return fail(c, TOKSCORE_E_HIP, hipGetErrorString(e));
	}
	int status = TOKSCORE_OK;
	std::vector<uint64_t> host(need + 1);
	do {
		if (n && (e = hipMemcpyAsync(d_text, text, n, hipMemcpyHostToDevice, c->stream)) != hipSuccess) break;
		launch_tokenize_one(trie_dev(c), d_text, n, d_ids, need, d_ids + need, c->stream);
		if ((e = hipMemcpyAsync(host.data(), d_ids, (need + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream)) != hipSuccess) break;
		e = hipStreamSynchronize(c->stream);
	} while (0);
	(void)hipFree(d_text);
	(void)hipFree(d_ids);
	if (e != hipSuccess) return fail(c, TOKSCORE_E_HIP, hipGetErrorString(e));
	const uint64_t k = host[need];
	*n_ids = k;
	if (k > cap) status = fail(c, TOKSCORE_E_CAPACITY, "ids_out too small");
	memcpy(ids_out, host.data(), std::min(k, cap) * sizeof(uint64_t));
	return status;
}

int tokscore_get_info(tokscore_ctx *c, tokscore_info *info) {
	if (!c || !info) return TOKSCORE_E_ARG;
	memset(info, 0, sizeof *info);
	info->vocab_size = c->V;
	if (c->vocab_set) {
		info->max_len = c->trie.max_len();
		info->trie_nodes = c->trie.node_count();
	}
	if (c->packed_valid) {
		info->trie_slots = c->packed.slots.size();
		info->slot_bits = 23;
		info->id_bits = 32;
	}
	info->corpus_bytes = c->total;
	info->corpus_docs = c->D;
	if (c->plan_valid) {
		info->chunks = c->nchunks;
		info->scan_path = (uint32_t)c->scan_path;
		info->trie_lds_slots = c->scan_path == 2 ? (uint32_t)c->simg.lt.size() : c->tile.lds_slots;
	}
	info->rescored_chunks = c->rescored_chunks;
	info->toggle_streams = c->toggle_items;
	info->base_scan_path = info->scan_path;
	info->search_passes = (uint32_t)std::min<uint64_t>(c->search_batches, 0xFFFFFFFFull);
	if (c->dfa_state == 1 && !c->dfa_invalid) {
		info->dfa_slots = c->dimg.nslots;
		info->dfa_lds_slots = c->dimg.hot_slots;
		info->dfa_lds_used = c->dimg.hot_used;
		info->dfa_virtual_slots = c->dimg.virtual_slots;
		info->dfa_active = c->dfa_used_last ? 1u : 0u;
		info->dfa_chains = c->dplan.chains;
		info->dfa_range_bytes = c->dplan.range_bytes;
		info->dfa_hot_fraction = c->dimg.hot_fraction();
		info->dfa_build_ms = c->dfa_build_ms;
		info->dfa_plan_ms = c->dfa_plan_ms;
		info->dfa_second_hits = c->dfa_second_hits;
		info->dfa_relayouts = c->dfa_relayouts;
		info->dfa_form = c->dfa_fail ? 2u : 1u;
		if (c->dfa_used_last) info->scan_path = 2;
	}
	return TOKSCORE_OK;
}

int tokscore_last_timing(tokscore_ctx *c, float ms[4]) {
	if (!c || !ms) return TOKSCORE_E_ARG;
	if (c->hostonly) return TOKSCORE_E_NO_DEVICE;
	if (c->timing_pending) {
		(void)hipSetDevice(c->device);
		HIP_TRY(c, hipEventSynchronize(c->ev[4]));
		HIP_TRY(c, elapsed_ms(&c->last_ms[0], c->ev[0], c->ev[4]));
		HIP_TRY(c, elapsed_ms(&c->last_ms[1], c->ev[1], c->ev[2]));
		HIP_TRY(c, elapsed_ms(&c->last_ms[2], c->ev[2], c->ev[3]));
		HIP_TRY(c, elapsed_ms(&c->last_ms[3], c->ev[3], c->ev[4]));
		c->timing_pending = false;
	}
	memcpy(ms, c->last_ms, sizeof c->last_ms);
	return TOKSCORE_OK;
}

int tokscore_set_timing(tokscore_ctx *c, int mode, uint32_t every) {
	if (!c || mode < 0 || mode > 2) return TOKSCORE_E_ARG;
	c->timing_on = mode != 0;
	c->timing_scan_only = mode == 2;
	c->timing_every = every ? every : 1;
	c->timing_skipped = 0;
	return TOKSCORE_OK;
}

int tokscore_drain_timings(tokscore_ctx *c, float *ms, uint32_t cap, uint32_t *n) {
	if (!c || !ms || !n) return TOKSCORE_E_ARG;
	if (c->hostonly) return TOKSCORE_E_NO_DEVICE;
	(void)hipSetDevice(c->device);
	uint64_t first = c->drained;
	if (c->calls - first > (uint64_t)tokscore_ctx::kRing) first = c->calls - tokscore_ctx::kRing;
	uint32_t k = 0;
	for (uint64_t i = first; i < c->calls && k < cap; i++, k++) {
		hipEvent_t *e = c->view[i % tokscore_ctx::kRing];
		HIP_TRY(c, hipEventSynchronize(e[4]));
		HIP_TRY(c, elapsed_ms(&ms[4 * k + 0], e[0], e[4]));
		HIP_TRY(c, elapsed_ms(&ms[4 * k + 1], e[1], e[2]));
		HIP_TRY(c, elapsed_ms(&ms[4 * k + 2], e[2], e[3]));
		HIP_TRY(c, elapsed_ms(&ms[4 * k + 3], e[3], e[4]));
	}
	c->drained = first + k;
	*n = k;
	return TOKSCORE_OK;
}

int tokscore_hosttrie_longest(tokscore_ctx *c, const uint8_t *text, uint64_t n, uint64_t *len, uint64_t *id) {
	if (!c || !len || !id || (!text && n)) return TOKSCORE_E_ARG;
	if (!c->vocab_set) return fail(c, TOKSCORE_E_STATE, "vocabulary not set");
	if (!c->packed_valid) {
		std::string e;
		if (!c->trie.Pack(&c->packed, &e)) return fail(c, TOKSCORE_E_VOCAB, e);
		c->packed_valid = true;
		c->device_trie_valid = false;
	}
	VocabTrie::Longest(c->packed, text, n, len, id);
	return TOKSCORE_OK;
}

int tokscore_hostsearch_check(const uint8_t *codes, const uint32_t *off, uint32_t n, uint32_t ncodes, uint32_t max_slots,
                              const uint8_t *text, uint64_t text_n, uint64_t *pairs, uint64_t cap, uint64_t *n_found,
                              uint64_t stats[3]) {
	if (!off || !n_found || (!codes && n && off[n]) || (!text && text_n) || (!pairs && cap)) return TOKSCORE_E_ARG;
	SearchImage img;
	if (!BuildSearchImage(codes, off, n, ncodes, max_slots, &img)) return TOKSCORE_E_CAPACITY;
	std::vector<std::pair<uint32_t, uint64_t>> found;
	SearchRun(img, text, text_n, &found);
	*n_found = found.size();
	for (uint64_t i = 0; i < found.size() && i < cap; i++) pairs[2 * i] = found[i].first, pairs[2 * i + 1] = found[i].second;
	if (stats) stats[0] = img.hot_slots, stats[1] = img.states, stats[2] = img.entries;
	return TOKSCORE_OK;
}

int tokscore_hostdfa_check(tokscore_ctx *c, const uint8_t *bytes, const uint64_t *off, uint64_t D, uint64_t max_slots,
                           uint32_t lds_slots, uint64_t *counts, uint64_t *total_tokens, uint64_t stats[8]) {
	if (!c || !off || !counts || !total_tokens || (!bytes && D && off[D] != off[0])) return TOKSCORE_E_ARG;
	if (!c->vocab_set) return fail(c, TOKSCORE_E_STATE, "vocabulary not set");
	DfaImage img;
	DfaOptions opt;
	if (max_slots) opt.max_slots = max_slots;
	if (lds_slots) opt.lds_slots = lds_slots;
	opt.sample = bytes ? bytes + off[0] : nullptr;
	opt.sample_bytes = std::min<uint64_t>(off[D] - off[0], kSampleBytes);
	opt.fail_mode = env_u64("TOKSCORE_HOSTDFA_FAIL", 0) != 0; // the failure form (dfa.h: DfaImage::fail_code) instead of the flattened one
	std::string why;
	if (!c->trie.BuildDfa(&img, opt, &why)) return fail(c, TOKSCORE_E_VOCAB, why);
	uint64_t global_reads = 0;
	std::vector<uint32_t> hits;
	DfaRun(img, bytes, off, D, c->V, counts, total_tokens, &global_reads, &hits);
	// TOKSCORE_HOSTDFA_RELAYOUT=n: what relayout_dfa does after an evaluation, n times -- the image built
	// again with the first run's hits in the global array counted in; results and stats are the last run's
	std::map<uint32_t, uint32_t> prior;
	std::vector<uint32_t> pk, pc;
	for (uint64_t r = env_u64("TOKSCORE_HOSTDFA_RELAYOUT", 0); r > 0; r--) {
		for (size_t i = 0; i < hits.size(); i++)
			if (hits[i] && img.ent_key[i] != DfaImage::kEvNone) prior[img.ent_key[i]] += hits[i];
		pk.clear(), pc.clear();
		for (const auto &kv : prior) pk.push_back(kv.first), pc.push_back(kv.second);
		opt.prior_keys = pk.data(), opt.prior_counts = pc.data(), opt.prior_n = pk.size();
		opt.corpus_bytes = off[D] - off[0] + D;
		if (!c->trie.BuildDfa(&img, opt, &why)) return fail(c, TOKSCORE_E_VOCAB, why);
		DfaRun(img, bytes, off, D, c->V, counts, total_tokens, &global_reads, &hits);
	}
	if (stats) {
		stats[0] = img.nslots, stats[1] = img.real_slots, stats[2] = img.virtual_slots, stats[3] = img.ncodes;
		stats[4] = img.hot_used, stats[5] = global_reads, stats[6] = img.sample_steps, stats[7] = img.sample_virtual;
	}
	return TOKSCORE_OK;
}

} // extern "C"
