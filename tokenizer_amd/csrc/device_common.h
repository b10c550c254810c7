// device_common.h -- structures and device helpers shared by the gfx950 kernels.
//
// All kernels restate SolutionFile::Tokenize (/root/reference/src/files/SolutionFile.cpp:83-100)
// on a packed byte trie (trie.h): at a position the walk follows lower-cased bytes
// while children exist and remembers the deepest node that carries an id; that is the
// reference's "first hit scanning len = min(max_len_, remaining) .. 1" (:90-96).  No id
// on the path means UNKNOWN and a one-byte advance (:92-94).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tokscore {

// Packed trie slot (see trie.h): w0 = label:8 | valid:1 | base:23, w1 = id.
constexpr uint32_t kSlotValid = 0x100u;
constexpr uint32_t kSlotKeyMask = 0x1FFu;
constexpr uint32_t kSlotBaseShift = 9;

struct TrieDev {
	const uint2 *slots;  // packed double array in HBM (L2-resident in practice)
	uint32_t nslots;
	uint32_t unknown_id; // V + 2
	uint32_t lds_slots;  // slots [0, lds_slots) are staged in LDS by the tile scan
	uint32_t root_base;  // base of slot 0
};

struct CorpusDev {
	const uint8_t *bytes; // raw (not lower-cased) document bytes, back to back
	const uint64_t *off;  // D + 1 document offsets, off[0] == 0, off[D] == total
	uint64_t D;
	uint64_t total;
};

// Per work-chunk boundary state.  A chunk [begin, end) owns the tokens that START in
// it.  entry = overshoot (bytes) of the chain coming in from the previous chunk that
// the chunk's currently counted chain assumed; exit = overshoot past `end` of that
// chain.  The scan speculates entry = 0; the fix-up kernels repair the rest.
struct ChunkState {
	const uint64_t *begin; // nchunks + 1 entries
	uint32_t *entry;       // assumed entry overshoot of the counted chain
	uint32_t *exit_a;      // exit overshoot written by the scan
	uint32_t *exit_b;      // exit overshoot after fix-up
	uint32_t nchunks;
};

// ---- helpers every kernel file uses: defined once, here ----
#define LDS_AS __attribute__((address_space(3)))
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// LDS is in order per wave; this only stops the compiler moving LDS accesses across.
__device__ __forceinline__ void wave_sync() {
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	__builtin_amdgcn_wave_barrier();
}
// A wave-uniform value (a count, a lane mask) that the compiler does not recognise as such.
__device__ __forceinline__ uint32_t uni32(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ unsigned long long uni64(unsigned long long x) {
	const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x);
	const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
	return ((unsigned long long)hi << 32) | lo;
}
// A-Z -> a-z on four packed bytes (bytes >= 0x80 untouched).
__device__ __forceinline__ uint32_t lower4(uint32_t x) {
	const uint32_t low7 = x & 0x7F7F7F7Fu;
	const uint32_t ge_A = low7 + 0x3F3F3F3Fu; // bit7 set iff low7 >= 'A' (0x41)
	const uint32_t gt_Z = low7 + 0x25252525u; // bit7 set iff low7 >  'Z' (0x5A)
	const uint32_t is_up = ge_A & ~gt_Z & ~x & 0x80808080u;
	return x | (is_up >> 2); // + 0x20
}
// first byte of a UTF-8 sequence (utf8/core.h:134-147)
__device__ __forceinline__ bool is_lead(uint32_t c) { return (c & 0xC0u) != 0x80u; }

// std::tolower on unsigned char in the "C" locale: only A-Z (SolutionFile.cpp:87-88).
__device__ __forceinline__ uint32_t lower8(uint32_t c) { return c + (((c - 65u) < 26u) ? 32u : 0u); }

// Smallest document offset strictly greater than pos (corpus end if none): the walk
// window is clipped there, as `input.size() - pos` clips it at :90.
__device__ __forceinline__ uint64_t next_boundary(const CorpusDev &C, uint64_t pos) {
	uint64_t lo = 0, hi = C.D + 1;
	while (lo < hi) {
		const uint64_t mid = (lo + hi) >> 1;
		if (C.off[mid] > pos) hi = mid;
		else lo = mid + 1;
	}
	return lo <= C.D ? C.off[lo] : C.total;
}

// One token of the greedy chain, trie and corpus both read from global memory.
__device__ __forceinline__ void walk_token_global(const TrieDev &T, const uint8_t *bytes, uint64_t pos,
                                                  uint64_t limit, uint32_t &len, uint32_t &id) {
	uint32_t base = T.root_base;
	uint32_t d = 0;
	len = 0;
	id = T.unknown_id;
	while (base != 0 && pos + d < limit) {
		const uint32_t c = lower8(bytes[pos + d]);
		const uint2 e = T.slots[base + c];
		if ((e.x & kSlotKeyMask) != (c | kSlotValid)) break;
		d++;
		base = e.x >> kSlotBaseShift;
		if (e.y) {
			len = d;
			id = e.y;
		}
	}
	if (len == 0) len = 1; // UNKNOWN advances one byte (SolutionFile.cpp:92-94)
}

} // namespace tokscore
