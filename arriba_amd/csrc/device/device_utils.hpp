// arriba_amd/csrc/device/device_utils.hpp -- device-only helpers shared by the kernels.
#ifndef AGPU_DEVICE_UTILS_HPP
#define AGPU_DEVICE_UTILS_HPP 1

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
const unsigned int TALLY_MAX_BLOCKS = 2048;
inline unsigned int tally_grid(uint64_t n, int block_threads) {
	const uint64_t blocks = (n + block_threads - 1) / block_threads;
	return (unsigned int) (blocks < 1 ? 1 : blocks > TALLY_MAX_BLOCKS ? TALLY_MAX_BLOCKS : blocks);
}

}

#endif
