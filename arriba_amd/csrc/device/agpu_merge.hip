// arriba_amd/csrc/device/agpu_merge.hip -- merge_adjacent_fusions on the device (reference: source/merge_adjacent_fusions.cpp:19-108,
// called at source/arriba.cpp:420-423).  Three stable radix sorts bring the candidates of one gene pair, direction pair and diagonal together
// in the reference's coordinate order; one thread per cluster (merge_core.hpp) runs the reference's sequential sweep.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <string>
#include <rocprim/rocprim.hpp>
#include "agpu_context.hpp"
#include "merge_core.hpp"
#include "device_utils.hpp"

using namespace agpu;

namespace {

const int BLOCK = 256;
inline unsigned int grid_for(uint64_t n) { return (unsigned int) ((n + BLOCK - 1) / BLOCK); }

#define HIP_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_last_error(std::string(#call) + ": " + hipGetErrorString(e_)); return AGPU_ERR_DEVICE; } } while (0)
#define ALLOC(buffer, bytes) do { if (!(buffer).allocate(bytes)) { set_last_error("hipMalloc failed (" #buffer ")"); return AGPU_ERR_DEVICE; } } while (0)

__global__ void merge_coordinate_key_kernel(CandidateTable t, uint64_t* keys) {
	uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c < t.n) keys[c] = (uint64_t) (uint32_t) t.breakpoint1[c] << 32 | (uint32_t) t.breakpoint2[c]; // breakpoints are non-negative
}
// pass 1: the diagonal, pass 2: gene pair + directions of the candidate at sorted position j
__global__ void merge_sort_key_kernel(CandidateTable t, const uint32_t* order, int pass, uint32_t max_itd_length, uint64_t* keys) {
	uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
	if (j >= t.n) return;
	const uint32_t c = order[j];
	keys[j] = pass == 1 ? merge_diagonal_key(t, c) : merge_group_key(t, c, max_itd_length);
}
__global__ void merge_cluster_kernel(CandidateTable t, const uint32_t* order, const uint64_t* group_keys, int32_t max_distance, uint32_t max_itd_length, uint32_t* extra_split_list) {
	uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
	if (j >= t.n || group_keys[j] == ~0ull) return;
	if (j > 0 && merge_cluster_continues(t, order, j, max_distance, max_itd_length)) return; // not the first of its cluster
	uint32_t end = j + 1;
	while (end < t.n && merge_cluster_continues(t, order, end, max_distance, max_itd_length)) ++end;
	if (end - j > 1) merge_cluster(t, order, j, end, max_distance, max_itd_length, extra_split_list);
}
__global__ void count_unfiltered_kernel(CandidateTable t, unsigned int* remaining) {
	__shared__ uint32_t block_sum;
	uint32_t kept = 0;
	for (uint32_t c = blockIdx.x * BLOCK + threadIdx.x; c < t.n; c += gridDim.x * BLOCK) kept += t.filter[c] == FILTER_none;
	block_tally(kept, remaining, &block_sum);
}

}

extern "C" int agpu_merge_adjacent_fusions(agpu_ctx* ctx, int32_t max_distance, uint64_t* remaining) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	DeviceBuffer& keys_in = ctx->scratch("merge.keys_in"); DeviceBuffer& keys_out = ctx->scratch("merge.keys_out"); DeviceBuffer& order_a = ctx->scratch("merge.order_a"); DeviceBuffer& order_b = ctx->scratch("merge.order_b");
	DeviceBuffer& counter = ctx->scratch("merge.counter"); DeviceBuffer& scratch = ctx->scratch("merge.rocprim");
	const size_t C1 = std::max<uint32_t>(C, 1);
	ALLOC(keys_in, C1 * 8); ALLOC(keys_out, C1 * 8); ALLOC(order_a, C1 * 4); ALLOC(order_b, C1 * 4); ALLOC(counter, 16); ALLOC(ctx->cand_extra_split_list, C1 * 4);
	HIP_CHECK(hipMemsetAsync(counter.ptr, 0, 16, s));
	HIP_CHECK(hipMemsetAsync(ctx->cand_extra_split_list.ptr, 0, C1 * 4, s));
	(void) hipEventRecord(ctx->event_start, s);
	if (C > 0 && ctx->params.filter_enabled[FILTER_merge_adjacent]) {
		const CandidateTable& t = ctx->candidates;
		{ KernelTimer timer(ctx, "merge_coordinate_key_kernel", (uint64_t) C * 16); merge_coordinate_key_kernel<<<grid_for(C), BLOCK, 0, s>>>(t, keys_in.as<uint64_t>()); }
		size_t bytes = 0;
		HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), rocprim::counting_iterator<uint32_t>(0), order_a.as<uint32_t>(), C, 0, 64, s));
		if (bytes > scratch.capacity) ALLOC(scratch, bytes);
		{ KernelTimer timer(ctx, "rocprim::radix_sort_pairs(merge: coordinates)", (uint64_t) C * 24);
		  HIP_CHECK(rocprim::radix_sort_pairs(scratch.ptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), rocprim::counting_iterator<uint32_t>(0), order_a.as<uint32_t>(), C, 0, 64, s)); }
		uint32_t* order = order_a.as<uint32_t>(); uint32_t* next = order_b.as<uint32_t>();
		for (int pass = 1; pass <= 2; ++pass) { // stable: the coordinate order survives inside a diagonal, the diagonal order inside a gene pair
			const int end_bit = pass == 1 ? MERGE_DIAGONAL_KEY_BITS : 64;
			{ KernelTimer timer(ctx, "merge_sort_key_kernel", (uint64_t) C * 30); merge_sort_key_kernel<<<grid_for(C), BLOCK, 0, s>>>(t, order, pass, ctx->params.max_itd_length, keys_in.as<uint64_t>()); }
			HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), order, next, C, 0, end_bit, s));
			if (bytes > scratch.capacity) ALLOC(scratch, bytes);
			{ KernelTimer timer(ctx, pass == 1 ? "rocprim::radix_sort_pairs(merge: diagonals)" : "rocprim::radix_sort_pairs(merge: gene pairs)", (uint64_t) C * 24);
			  HIP_CHECK(rocprim::radix_sort_pairs(scratch.ptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), order, next, C, 0, end_bit, s)); }
			std::swap(order, next);
		}
		{ KernelTimer timer(ctx, "merge_cluster_kernel", (uint64_t) C * 40);
		  merge_cluster_kernel<<<grid_for(C), BLOCK, 0, s>>>(t, order, keys_out.as<uint64_t>(), max_distance, ctx->params.max_itd_length, ctx->cand_extra_split_list.as<uint32_t>()); }
	}
	if (C > 0) count_unfiltered_kernel<<<tally_grid(C, BLOCK), BLOCK, 0, s>>>(ctx->candidates, counter.as<unsigned int>());
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (uint64_t) C * 120;
	unsigned int kept = 0;
	HIP_CHECK(hipMemcpy(&kept, counter.ptr, 4, hipMemcpyDeviceToHost));
	if (remaining) *remaining = kept;
	return AGPU_OK;
}
