// arriba_amd/csrc/device/device_utils.hpp -- device-only helpers shared by the kernels.
#ifndef AGPU_DEVICE_UTILS_HPP
#define AGPU_DEVICE_UTILS_HPP 1

#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace agpu {

// Appends 0..2 items per thread to a global list with ONE atomic per workgroup (a list cursor that every wavefront or thread bumps on its
// own serialises at the L2: ~10 ns per atomic, i.e. milliseconds per launch at 10^5..10^6 appends).  Ballots give the position inside
// the wavefront, LDS the position of the wavefront inside the workgroup.  Returns the index of the thread's first item.  Every thread of
// the workgroup must call it.  wave_offset: LDS, BLOCK_THREADS / 64 words; block_base: LDS, one word.
template <int BLOCK_THREADS> __device__ __forceinline__ uint32_t block_append(uint32_t mine, uint32_t* cursor, uint32_t* wave_offset, uint32_t* block_base) {
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const unsigned long long one = __ballot(mine >= 1), two = __ballot(mine >= 2), before = (1ull << lane) - 1;
	if (lane == 0) wave_offset[wave] = (uint32_t) (__popcll(one) + __popcll(two));
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t total = 0;
		for (int w = 0; w < BLOCK_THREADS / 64; ++w) { uint32_t c = wave_offset[w]; wave_offset[w] = total; total += c; }
		*block_base = total ? atomicAdd(cursor, total) : 0;
	}
	__syncthreads();
	return *block_base + wave_offset[wave] + (uint32_t) (__popcll(one & before) + __popcll(two & before));
}

// The same for an arbitrary number of items per thread (wave inclusive scan with shuffles).
template <int BLOCK_THREADS> __device__ __forceinline__ uint32_t block_reserve(uint32_t mine, uint32_t* cursor, uint32_t* wave_offset, uint32_t* block_base) {
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t inclusive = mine;
	for (int offset = 1; offset < 64; offset <<= 1) {
		const uint32_t other = __shfl_up(inclusive, offset);
		if (lane >= (uint32_t) offset) inclusive += other;
	}
	if (lane == 63) wave_offset[wave] = inclusive;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t total = 0;
		for (int w = 0; w < BLOCK_THREADS / 64; ++w) { uint32_t c = wave_offset[w]; wave_offset[w] = total; total += c; }
		*block_base = total ? atomicAdd(cursor, total) : 0;
	}
	__syncthreads();
	return *block_base + wave_offset[wave] + inclusive - mine;
}

// Adds the per-thread tallies of a workgroup to a global counter with ONE atomic per workgroup; kernels that only count run as a capped
// grid (tally_grid) with a grid-stride loop, so a launch over millions of items issues ~10^3 atomics instead of one per wavefront.
// Every thread of the workgroup must call it.  block_sum: LDS, one word (any value on entry).
__device__ __forceinline__ void block_tally(uint32_t mine, unsigned int* counter, uint32_t* block_sum) {
	if (threadIdx.x == 0) *block_sum = 0;
	__syncthreads();
	for (int offset = 32; offset > 0; offset >>= 1) mine += __shfl_down(mine, offset);
	if ((threadIdx.x & 63) == 0 && mine) atomicAdd(block_sum, mine);
	__syncthreads();
	if (threadIdx.x == 0 && *block_sum) atomicAdd(counter, *block_sum);
}
// A launch takes fewer than 2^32 work-items.  A kernel that gives every ITEM a wavefront (a candidate with its read lists, a bucket of discordant mates, a group of candidates) would
// exceed that at 2^26 = 67 M items -- a sample of 2 x 10^8 fragments has that many candidates -- and the runtime refuses such a launch without a word to the caller (the bug of
// round 5 in agpu_get_candidate_read_lists_of).  These kernels take the index of their first item, and the host launches them in chunks: for_each_wave_chunk(items, launch(first, count)).
// ARRIBA_WAVE_CHUNK (tests/test_gpu_parity.py): a chunk of a few items, so that every such kernel of the path runs in many chunks on a small sample and must give the same files.
inline uint64_t wave_chunk_items() {
	const char* knob = getenv("ARRIBA_WAVE_CHUNK"); // (read at every launch: a test switches it inside one process)
	const long long asked = knob != nullptr ? atoll(knob) : 0;
	// (a multiple of 16 items: the kernels run workgroups of up to 16 wavefronts and check their item against the END OF THE LIST only -- a chunk that ended inside a workgroup
	//  would have its last items done again by the first workgroup of the next chunk; found by the test with chunks of 3)
	return asked > 0 ? ((uint64_t) asked + 15) & ~(uint64_t) 15 : (uint64_t) 1 << 24;
}
template <class Launch> inline void for_each_wave_chunk(uint64_t items, Launch launch) {
	const uint64_t chunk = wave_chunk_items();
	for (uint64_t first = 0; first < items; first += chunk) launch(first, items - first < chunk ? items - first : chunk);
}
const unsigned int TALLY_MAX_BLOCKS = 2048;
inline unsigned int tally_grid(uint64_t n, int block_threads) {
	const uint64_t blocks = (n + block_threads - 1) / block_threads;
	return (unsigned int) (blocks < 1 ? 1 : blocks > TALLY_MAX_BLOCKS ? TALLY_MAX_BLOCKS : blocks);
}

}

#endif
