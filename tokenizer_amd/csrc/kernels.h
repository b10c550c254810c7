// kernels.h -- launch entry points of the gfx950 kernels (host-callable wrappers).
#pragma once
#include <string>

#include "device_common.h"

namespace tokscore {

// kernels_generic.hip
void launch_scan_generic(const TrieDev &T, const CorpusDev &C, const ChunkState &S, unsigned long long *counts,
                         hipStream_t stream);
void launch_fixup(const TrieDev &T, const CorpusDev &C, const ChunkState &S, unsigned long long *counts,
                  hipStream_t stream);
// Moves interior chunk boundaries forward to positions where every greedy chain starts a
// token (synced[j] = 1), so that those chunks need no entry repair.
void launch_snap_chunks(const TrieDev &T, const CorpusDev &C, const uint64_t *begin_in, uint64_t *begin_out,
                        uint32_t *synced, uint32_t nchunks, uint32_t max_len, uint32_t window, hipStream_t stream);
// Incremental re-score: bit j of sel = 1 for every chunk that holds an occurrence of the byte string
// tok[0 .. len) in the lower-cased corpus (chunks cut every chunk_bytes, each boundary moved
// forward by less than chunk_bytes).
void launch_find_token_chunks(const CorpusDev &C, const uint64_t *begin, uint32_t nchunks, uint64_t chunk_bytes,
                              const uint8_t *tok, uint32_t len, uint32_t *sel, hipStream_t stream);
// list[0 .. n) = the chunks whose bit is set in sel (any order), list[nchunks] = n
void launch_compact_sel(const uint32_t *sel, uint32_t nchunks, uint32_t *list, hipStream_t stream);
// Occurrence index, built once per plan: index[row][chunk bit] = string `row` occurs in the chunk
// (slot_row[t] = row of the string that ends at slot t, or -1); row_chunks[row] = its chunk count.
// launch_or_rows: sel = OR of up to 8 rows.  (sel is a chunk bitmap, (nchunks + 31) / 32 words.)
void launch_index_all(const TrieDev &T, const CorpusDev &C, const int32_t *slot_row, const uint64_t *begin,
                      uint32_t nchunks, uint64_t chunk_bytes, uint32_t rows, uint32_t *index, uint32_t *row_chunks,
                      hipStream_t stream);
void launch_or_rows(const uint32_t *index, const uint32_t *rows, uint32_t n, uint32_t nchunks, uint32_t *sel,
                    hipStream_t stream);
void launch_read_probe(const void *buf, uint64_t bytes, uint32_t blocks, uint32_t *sink, hipStream_t stream);
void launch_finalize(unsigned long long *out, uint32_t V, uint64_t D, uint64_t total_bytes, hipStream_t stream);
void launch_tokenize_one(const TrieDev &T, const uint8_t *text, uint64_t n, unsigned long long *ids, uint64_t cap,
                         unsigned long long *n_ids, hipStream_t stream);

// kernels_tile.hip -- LDS tile scan + token-stream histogram
struct TileScanPlan {
	uint32_t waves_per_block; // waves (64 lanes) per workgroup
	uint32_t seg_bytes;       // bytes per lane segment (4 * odd)
	uint32_t lds_slots;       // trie slots staged in LDS
	uint32_t lds_bytes;       // dynamic LDS per workgroup
	uint32_t blocks;          // persistent workgroups
	uint32_t wide_ids;        // 0: 16-bit stream entries, 1: 32-bit
	uint32_t hist_range;      // most ids one histogram pass keeps bins for (0: what LDS holds)
};
// Token stream and work-queue words shared by the tile scan and the histogram.
constexpr uint32_t kStreamBatch = 32;     // stream blocks handed out per allocation
constexpr uint32_t kWorkQueues = 64;      // sharded chunk-queue heads
constexpr uint32_t kQueueHeadWord = 32;   // first head at cursor[32], one per 128-byte line
constexpr uint32_t kQueueHeadStride = 32;
constexpr uint32_t kCursorWords = kQueueHeadWord + kWorkQueues * kQueueHeadStride;
struct StreamDev {
	void *blocks;               // token stream, blocks of stream_block_bytes()
	uint32_t capacity_batches;  // batches of kStreamBatch blocks
	uint32_t *fill;             // [capacity_batches] blocks written per batch
	uint32_t *cursor;           // [0] batches handed out, [1] overflow flag, [kQueueHeadWord + q*stride] queue heads
};
uint32_t stream_block_bytes(uint32_t wide_ids);
bool plan_tile_scan(const TrieDev &T, uint64_t max_len, uint32_t V, uint32_t num_cus, TileScanPlan *plan);
void launch_scan_tile(const TrieDev &T, const CorpusDev &C, const ChunkState &S, const StreamDev &Q,
                      const TileScanPlan &plan, const uint64_t *chunk_first_doc, uint64_t max_len,
                      hipStream_t stream, const uint32_t *sel_list = nullptr, const uint32_t *sel_count = nullptr);
// Clears out[0 .. V+5), the stream cursor and fill words in one launch and presets
// out[0] = out[1] = D (START / END once per document) and out[V+4] = total_bytes.
// out == nullptr: leave the counts as they are (incremental re-score).
// first_batches: stream batches that belong to the scan's waves from the start (scan_ranges_kernel: one per wave)
void launch_prepare(unsigned long long *out, uint32_t V, uint64_t D, uint64_t total_bytes, const StreamDev &Q,
                    hipStream_t stream, uint32_t first_batches = 0);
// counts += histogram of the token stream; out[V+3] += number of tokens counted.
// flip: subtract what the stream adds and add what it subtracts.
void launch_histogram(const StreamDev &Q, const TileScanPlan &plan, uint32_t V, unsigned long long *counts,
                      uint32_t *slab, uint32_t slab_blocks, hipStream_t stream, bool flip = false);

// slab rows (u32 partials per workgroup, n_cols columns) -> counts[] (u64 atomics)
void launch_slab_reduce(const uint32_t *slab, uint32_t rows, uint32_t n_cols, unsigned long long *counts,
                        hipStream_t stream);

// kernels_stream.hip -- stream scan: one independent byte stream per lane, the whole trie in LDS
struct StreamTrieDev {      // trie.h: StreamImage
	const uint8_t *lt;     // [nslots padded to 16] label code | 0x80 = a token ends here
	const uint16_t *base;  // [nslots padded to 8]
	uint32_t nslots;
	uint32_t root_base;
};
// One stream a move's string occurs in (at a token start of the vocabulary as it is), with where:
// up to kToggleOffsets occurrence offsets from the stream's begin, in no order.  n = kToggleWhole:
// too many (or a stream of 64 KB and more): the whole stream is walked instead.
constexpr uint32_t kToggleOffsets = 11;
constexpr uint16_t kToggleWhole = 0xFFFFu;
struct ToggleItem {
	uint32_t stream, move;
	uint16_t n;
	uint16_t off[kToggleOffsets];
};
static_assert(sizeof(ToggleItem) == 32, "two items per cache-line half");
struct ByteCodes {
	uint8_t code[256];     // byte -> code (StreamImage::code)
};
struct StreamWork {
	const uint8_t *prep;       // prepared corpus: codes of the lower-cased bytes, code 0 after every document
	const uint32_t *pbeg;      // [nstreams + 1] stream boundaries (prepared offsets), all guaranteed token starts
	uint32_t nstreams;
	uint32_t prep_total;       // bytes of the prepared corpus
	uint32_t range_bytes;      // full evaluation: the corpus in nranges equal ranges, one lane each ...
	uint32_t nranges;
	uint32_t *entry;           // ... [nranges] entry overshoot the range's counted chain starts with
	uint32_t *exit_a;          // ... [nranges] exit overshoot, written by the scan
	const uint8_t *tprep;      // ... the ranges transposed in 8-byte pieces, 64 to a block of (range_bytes + halo) * 64 bytes
	uint32_t halo;             //     bytes of the next range stored after each (multiple of 8)
	uint32_t round;            //     lookups between two ring refills
	const uint32_t *sel_list;  // incremental re-score: selected chunks (or nullptr) ...
	const uint32_t *sel_count; // ... and their number
	const uint32_t *chunk_first; // [nchunks + 1] first stream of every chunk
	uint32_t streams_per_chunk;  // upper bound of the streams in one chunk
	// batched move evaluation (launch_walk_toggles): item = move << 32 | stream; nullptr = count
	// every stream's tokens into tg_base
	const ToggleItem *tg_items = nullptr;
	uint32_t *tg_base = nullptr;        // [nstreams] tokens per stream under the vocabulary as it is
	const uint32_t *tg_order = nullptr; // [nstreams] with tg_base: the streams by length, longest first (optional)
	const uint32_t *block_stream = nullptr; // optional, [prep_total / 256 + 1]: the stream that holds prepared offset 256 b (launch_stream_blocks)
	const unsigned long long *tg_known = nullptr; // local walks, optional: tg_starts of the vocabulary as it is (then only the chain with the move is walked)
	unsigned long long *tg_starts = nullptr; // [prep_total / 64 + 1] written with tg_base (zeroed by the caller): bit q = a token of the vocabulary as it is starts at q
	const uint32_t *tg_count = nullptr; // [kToggleLists] items appended per list (may exceed tg_cap: the surplus was not stored)
	uint32_t tg_cap = 0;                // per list
	const uint32_t *tg_index = nullptr; // optional: list entry -> item (the lists then hold indices into tg_items, tg_n)
	const uint32_t *tg_n = nullptr;     // optional, by item: the item's n - 1 (n above kToggleOffsets: the whole stream), instead of ToggleItem::n
	const uint32_t *tg_slot = nullptr;  // [moves] the slot whose token-end bit the move flips
	unsigned long long *tg_delta = nullptr; // [moves] tokens with the move - tokens without, mod 2^64
	uint32_t tg_whole = kToggleOffsets; // with tg_n: items of more occurrences than this are walked whole (launch_walk_toggles), the others locally
	uint32_t tg_moves = 0;              // optional: moves of the batch (the local walks then add up per workgroup first)
};
constexpr uint32_t kToggleLists = 64; // the items of a batch are kept in this many lists, one per lane of the walk
// One batch of independent moves (tokscore_score_toggles).  All pointers are device memory.
struct ToggleBatch {
	uint32_t n;                 // moves
	const uint32_t *index;      // occurrence index: rows of words_per_row words, one bit per chunk
	uint32_t words_per_row;
	const uint32_t *rows;       // [n] index row of the move's string
	const uint32_t *pair_off;   // [n + 1] first (move, chunk) pair of every move; pair_off[m + 1] - pair_off[m] = chunks in its row
	const uint32_t *str_off;    // [n + 1] the move's string in `codes`
	const uint8_t *codes;       // strings in the prepared corpus' code space
	uint4 *pairs;               // [pair_off[n]] scratch, one record per pair: first stream | streams (10 bits) + move << 10 |
	                            // prepared offsets of the chunk's begin and end (streams = 0: nothing there)
	const uint4 *moves;         // [n] per move: str_off | bytes | its first four codes as a dword | their mask
	const uint32_t *chunk_first, *pbeg; // StreamWork's, for the kernel that writes `pairs`
	ToggleItem *items;          // [kToggleLists][cap_items] scratch: one item for every stream the string occurs in
	uint32_t *n_items;          // [kToggleLists] items per list, zeroed by the launcher
	uint32_t cap_items;         // per list
	const unsigned long long *starts = nullptr; // optional (StreamWork::tg_starts): only occurrences at a token start of the
	                            // vocabulary as it is count -- a move changes nothing in a stream without one
	uint32_t *seen = nullptr;   // optional, one bit per stream, zeroed by the caller: a stream is listed once
	                            // whichever moves' strings occur in it (incremental re-score)
};
struct StreamScanPlan {
	uint32_t ring;            // bytes of a lane's LDS ring
	uint32_t round = 16;      // lookups between two refills
	uint32_t halo = 0;        // scan_ranges_kernel: bytes of the next range a block stores after each range
	uint32_t waves_per_block;
	uint32_t blocks;
	uint32_t lds_bytes;
};
void launch_prep_corpus(const CorpusDev &C, const ByteCodes &M, uint8_t *prep, uint64_t prep_total, uint32_t *flags,
                        hipStream_t stream, bool lower = true);
bool plan_stream_scan(uint32_t nslots, uint64_t max_len, uint32_t num_cus, StreamScanPlan *plan);
void launch_scan_stream(const StreamTrieDev &T, const StreamWork &W, const StreamDev &Q, const StreamScanPlan &plan,
                        hipStream_t stream);
// Row f1 on the stream image: one stream entry (the slot the walk ended on) per code point position
// of W.tprep, which here is the transposed prepared corpus WITHOUT lower-casing; *positions += their
// number.  Per-slot totals: launch_histogram_slots with slot_id = nullptr.
void launch_count_ends(const StreamTrieDev &T, const StreamWork &W, const StreamDev &Q, const StreamScanPlan &plan,
                       uint32_t first_cont, uint32_t max_cp, unsigned long long *positions, hipStream_t stream);
// Batched move evaluation.  launch_find_toggle_streams: the streams each move's string occurs in
// (B.items / B.n_items), from the occurrence index and a search of the chunks it lists;
// launch_walk_toggles: every listed stream walked with the move's token-end bit flipped, the
// difference to its token count under the vocabulary as it is (W.tg_base, filled by the same
// launcher when W.tg_items is null) added to W.tg_delta[move].  max_items bounds the work for the grid.
void launch_find_toggle_streams(const ToggleBatch &B, const StreamWork &W, uint32_t npairs, uint32_t nchunks,
                                hipStream_t stream);
void launch_walk_toggles(const StreamTrieDev &T, const StreamWork &W, const StreamScanPlan &plan, uint64_t max_items,
                         hipStream_t stream);
// W.tg_base / W.tg_starts for all streams (the vocabulary as it is), W.tg_order optional
// the same two results from the ranges of the full evaluation that has just run (W.entry exact, W.tprep,
// plan = the ranges plan): W.tg_base is ADDED to (zero it first), W.tg_starts as for launch_count_streams
void launch_mark_ranges(const StreamTrieDev &T, const StreamWork &W, const StreamScanPlan &plan, hipStream_t stream);
void launch_count_streams(const StreamTrieDev &T, const StreamWork &W, const StreamScanPlan &plan, hipStream_t stream);
// The items with occurrence offsets: from every occurrence the chain of the vocabulary as it is and
// the chain with the move are walked side by side until they meet again (a few dozen bytes), the
// difference in tokens is added to W.tg_delta[move].  launch_walk_toggles then only takes the
// kToggleWhole items.
void launch_local_toggles(const StreamTrieDev &T, const StreamWork &W, uint32_t num_cus, uint64_t max_items, hipStream_t stream);
// full evaluation over the transposed ranges (W.tprep)
bool plan_ranges_scan(uint32_t nslots, uint64_t max_len, uint32_t num_cus, StreamScanPlan *plan);
void launch_transpose_ranges(const uint8_t *prep, uint64_t prep_readable, uint32_t range_bytes, uint32_t halo, uint32_t nblocks,
                             uint8_t *tprep, hipStream_t stream);
// init (optional): the output vector the evaluation will add to, cleared and preset by the scan's
// own workgroups before they start (what prepare_kernel does in a launch of its own otherwise)
struct OutInit {
	unsigned long long *out = nullptr; // V + 5 words: counts | total_tokens | total_bytes
	uint32_t V = 0;
	unsigned long long D = 0, total_bytes = 0;
};
// ev_start / ev_stop (both or neither): events that take the kernel's own begin and end from its
// dispatch (hipExtLaunchKernel), not the times a record placed before and after it would read -- those
// include what the stream waits around a record, about 4 us of a 90 us kernel.
void launch_scan_ranges(const StreamTrieDev &T, const StreamWork &W, const StreamDev &Q, const StreamScanPlan &plan,
                        hipStream_t stream, const OutInit &init = OutInit{}, hipEvent_t ev_start = nullptr,
                        hipEvent_t ev_stop = nullptr);
// out[slot_id[s]] += occurrences of slot s in the stream, out[total_index] += entries on counted slots
void launch_histogram_slots(const StreamDev &Q, uint32_t nslots, const uint32_t *slot_id, unsigned long long *out,
                            uint32_t total_index, uint32_t *slab, uint32_t slab_blocks, hipStream_t stream, bool flip = false);
// After a full-evaluation scan: repairs the ranges whose entry was not the previous range's exit
// (counts straight into out), leaves the true entries in W.entry.  changed: two words, zero on entry;
// epoch: 1 .. 65535, a different one for every evaluation of a plan.
// next_first_batches != 0: the reduction leaves Q.cursor as prepare_kernel would set it for the next
// full evaluation (batches handed out = next_first_batches, the fix-up's two words zero; the
// overflow word is left alone), so that evaluation needs no prepare launch
void launch_fixup_and_histogram_slots(const StreamTrieDev &T, const StreamWork &W, uint32_t *exit_b, uint32_t *changed,
                                      uint32_t epoch, const StreamDev &Q, uint32_t nslots, const uint32_t *slot_id,
                                      unsigned long long *out, uint32_t total_index, uint32_t *slab, uint32_t slab_blocks,
                                      hipStream_t stream, uint32_t next_first_batches = 0);
void launch_fixup_stream(const StreamTrieDev &T, const uint32_t *slot_id, const StreamWork &W, uint32_t *exit_b,
                         uint32_t *changed, uint32_t epoch, unsigned long long *out, uint32_t total_index, hipStream_t stream);
uint64_t stream_scan_capacity_batches(uint64_t prep_total, uint32_t waves_total);

// as launch_histogram_slots, for the event streams of the automaton scan (kernels_dfa.hip): ev[s] is an
// event (dfa.h: an id, kEvNone, or a list in `pool`); out[total_index] += the tokens the events stand for
void launch_histogram_events(const StreamDev &Q, uint32_t nslots, const uint32_t *ev, const uint32_t *pool,
                             unsigned long long *out, uint32_t total_index, uint32_t *slab, uint32_t slab_blocks, hipStream_t stream);

// kernels_dfa.hip -- full evaluation on the one-transition-per-byte automaton (dfa.h)
struct DfaDev {
	const uint32_t *hot;     // [hot_slots, padded to 16 bytes] the LDS array (dfa.h: kArr | partial | 0 : 6 | label : 8 | base : 16)
	const uint2 *slot;       // [nslots] the global array: .x the entry's word, .y its target's global base; rows at bases >= hot_slots
	const uint32_t *gof;     // [hot_slots] by LDS base: the state's global base
	const uint32_t *ev_miss; // [nslots] by state key (LDS base, or global base under cold_base): leaving the state by a miss
	const uint32_t *ev_arr;  // [nslots] by entry index (either array): arriving on a virtual entry
	const uint32_t *pool;    // event lists
	uint32_t nslots;
	uint32_t hot_slots;      // keys and entries below it go to the streams, the others to counters
	uint32_t cold_base;      // the LDS base of the states without one of their own
	uint32_t stride;         // the depth-1 state of code c has its row at c * stride
	// != 0: the image is in the failure form (dfa.h: DfaImage::fail_code) -- a state is known by the entry it was
	// entered through (the second word of a range's entry / exit state is that slot index, gof and ev_miss are
	// indexed by it and cover both arrays), a miss leads to the row's entry under this pseudo-code unless the
	// state's word carries kRootFail.  Scanned by dfa_fscan_kernel.
	uint32_t fail_code = 0;
};
struct DfaWork {
	const uint2 *tcodes;     // the ranges' codes transposed (kernels_dfa.hip: dfa_transpose_kernel)
	uint32_t range_bytes;    // multiple of 8
	uint32_t nranges;        // nwb * 64 * chains
	uint32_t nwb;            // wave blocks: 64 lanes x `chains` ranges each
	uint2 *entry;            // [nranges] state (word, global base) the range's count starts from
	uint2 *exit_a;           // [nranges] state the scan ended the range in
	uint2 *exit_b;           // [nranges] ... after the fix-up, where exit_tag[g] == the evaluation's epoch
	uint32_t *exit_tag;
	uint32_t *cold_miss;     // [nslots - hot_slots] by key / entry hot_slots + i (zero between evaluations)
	uint32_t *cold_arr;
	uint32_t *cold_hit;      // ... hits in the global array, by entry (never cleared by an evaluation: the host reads it for the next layout)
};
struct DfaScanPlan {
	uint32_t blocks, waves_per_block, chains;
	uint32_t lds_slots, lds_bytes;
	uint32_t range_bytes, nwb;
};
// ---- one-pass search of a batch of strings (tokscore_score_toggles; dfa.h: SearchImage) ----
struct SearchDev {
	const uint32_t *hot;       // [hot_slots, padded to 16 bytes] the search automaton's LDS array
	const uint32_t *out_first; // [hot_slots] by entry: first record of out_list (0xFFFFFFFF: none)
	const uint32_t *out_list;  // string | bytes << 16 | 0x80000000 on an entry's last record
	uint32_t hot_slots, stride, max_len;
};
// (string, stream) -> item, open addressing; what launch_local_toggles / launch_walk_toggles then read as
// the kToggleLists lists list_index (StreamWork::tg_index; tg_n = count: an item's n)
struct SearchItems {
	unsigned long long *keys;  // [slots] 0 = free, else (string << 32 | stream) + 1
	uint32_t *count;           // [slots] occurrences on a token start seen for the key after the first (zeroed by the caller, with keys)
	ToggleItem *items;         // [slots] stream, move (= string), off[] filled here; n is `count`
	uint32_t mask;             // slots - 1, slots a power of two
	uint32_t *flags;           // [0] != 0: the table was full (the caller comes again with a larger one)
	uint32_t *list_index;      // [kToggleLists][list_cap] the slots in use, in the order they were claimed: what the walks read
	uint32_t *list_count;      // [kToggleLists] slots listed (may exceed list_cap: the surplus was not stored; zeroed by the caller)
	uint32_t list_cap;
	const unsigned long long *starts; // bit q: a token of the vocabulary as it is starts at prepared offset q
	const uint32_t *pbeg;      // [nstreams + 1] stream boundaries (prepared offsets)
	const uint32_t *block_stream; // [prep_total / 256 + 1] the stream that holds prepared offset 256 b
	uint32_t nstreams;
	uint32_t prep_total;       // bytes of the prepared corpus (`starts` has no bits from there on)
};
// block_stream from pbeg (once per plan)
void launch_stream_blocks(const uint32_t *pbeg, uint32_t nstreams, uint32_t prep_total, uint32_t *block_stream, hipStream_t stream);
// The automaton over all ranges of W.tcodes, every range from the start state and on over the first max_len - 1 bytes
// of the next (from `codes`, the linear prepared corpus): records (entry | bytes behind the boundary << 16, position of
// the string's last byte) go to Q (8 bytes each, Q.cursor[0] preset to the grid's waves, Q.cursor[1] = 0);
// launch_search_items turns them into items.
void launch_dfa_search(const SearchDev &A, const DfaWork &W, const StreamDev &Q, const uint8_t *codes, const DfaScanPlan &plan, hipStream_t stream);
void launch_search_items(const SearchDev &A, const StreamDev &Q, const SearchItems &H, uint32_t num_cus, hipStream_t stream);
// dfa_lds_slots_max(lds_cap): slots of the LDS array; lds_cap = 0, or fewer than fit (tests: most lookups
// and events then go the way of the global array)
bool plan_dfa_scan(uint32_t hot_slots, uint64_t prep_total, uint32_t num_cus, uint32_t chains_hint, DfaScanPlan *plan);
uint32_t dfa_lds_slots_max(uint32_t lds_cap);
uint64_t dfa_stream_capacity_batches(const DfaScanPlan &plan);
void launch_dfa_transpose(const uint8_t *prep, uint64_t prep_total, uint32_t range_bytes, uint32_t chains, uint64_t nwb,
                          void *tcodes, hipStream_t stream);
void launch_dfa_reset_states(const DfaWork &W, uint32_t start_word, hipStream_t stream);
// After launch_dfa_reset_states, for a new image: every range's entry state from a walk over the `lookback` bytes in
// front of the range (codes: the prepared corpus, linear), started in the separator's state.  A guess like the reset's,
// only a far better one -- the automaton forgets where it started within a few tokens -- so the first evaluation's
// fix-up finds most entries right instead of repairing every range (7 - 16 ms per image).  Nothing is counted.
void launch_dfa_warm_entries(const DfaDev &A, const DfaWork &W, const uint8_t *codes, uint32_t start_word, uint32_t lookback,
                             hipStream_t stream);
// clears out (when given) and presets START / END / total_bytes; both streams' cursors = first_batches; changed[0..1] = 0
void launch_dfa_prepare(unsigned long long *out, uint32_t V, uint64_t D, uint64_t total_bytes, const StreamDev &Qm,
                        const StreamDev &Qa, uint32_t first_batches, uint32_t *changed, hipStream_t stream);
void launch_dfa_scan(const DfaDev &A, const DfaWork &W, const StreamDev &Qm, const StreamDev &Qa, const DfaScanPlan &plan,
                     hipStream_t stream, const OutInit &init = OutInit{}, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// After the scan, two launches: the fix-up of the range entries and the histograms of both streams (miss_blocks +
// arrival_blocks workgroups, a slab row of 16 bits x hot_slots each), then the slab's column sums and the
// global array's counters through the event tables into out; the cursors go back to first_batches and
// changed[0..1] to zero for the next evaluation.  codes: the prepared corpus (linear), readable up to
// nranges * range_bytes.
void launch_dfa_count(const DfaDev &A, const DfaWork &W, const StreamDev &Qm, const StreamDev &Qa, const uint8_t *codes,
                      uint32_t *changed, uint32_t epoch, unsigned long long *out, uint32_t total_index, uint32_t *slab,
                      uint32_t miss_blocks, uint32_t arrival_blocks, uint32_t first_batches, hipStream_t stream);

// kernels_count.hip -- closed-vocabulary all-occurrence counting (GetTokens.cpp:21-27)
struct CountPlan {
	uint32_t halo;      // bytes staged past a tile (longest walk + slack)
	uint32_t nbins;     // ids counted per pass (LDS bins)
	uint32_t passes;    // passes over the corpus, one id range each
	uint32_t lds_slots; // trie slots staged in LDS
	uint32_t lds_bytes;
};
bool plan_count(const TrieDev &T, uint64_t max_len_bytes, uint32_t max_cp, uint32_t n_ids, CountPlan *plan);
uint32_t count_tiles(uint64_t total_bytes);
uint32_t count_blocks(uint64_t total_bytes, uint32_t num_cus);
size_t count_head_words(const CountPlan &plan);
// One pass per id range; every pass writes its id columns of all `blocks` slab rows
// (n_ids + 1 columns; the last one = code-point positions, written by the first pass).
// Scratch: heads[count_head_words(plan)] (queue heads, set here), tile_first[count_tiles()].
// walks = trie walks a lane keeps in flight (2 or 4).
hipError_t launch_count_all(const TrieDev &T, const CorpusDev &C, const CountPlan &plan, uint32_t max_cp, uint32_t n_ids,
                            uint32_t blocks, uint32_t walks, uint32_t *heads, uint64_t *tile_first, uint32_t *slab,
                            hipStream_t stream);

// kernels_extract.hip (row f3): the candidate set of the resident corpus, arrays in device memory
constexpr uint32_t kExtractMaxCp = 32;
struct CandidateArraysDev {
	uint32_t *chr = nullptr;  // first code point of the string (Trie.cpp:81-89)
	uint64_t *uses = nullptr; // occurrences (Trie.cpp:82)
	uint32_t *lpar = nullptr, *rpar = nullptr;
	uint64_t n = 0;           // tokens
	uint64_t positions = 0;   // code points in the corpus (the trie root's freq)
};
// 0 ok, 1 HIP error, 2 input outside what the kernels take (message in *err)
int extract_candidates(const CorpusDev &C, uint32_t max_cp, CandidateArraysDev *out, std::string *err, hipStream_t stream);
void free_candidates(CandidateArraysDev *c);

} // namespace tokscore
