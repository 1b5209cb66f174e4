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
#define ALLOC(buffer, bytes) do { if (!(buffer).allocate(bytes)) { set_last_error("hipMalloc failed (" #buffer ")"); return AGPU_ERR_NO_MEMORY; } } while (0)

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
__global__ void merge_cluster_kernel(CandidateTable t, const uint32_t* order, const uint64_t* group_keys, int32_t max_distance, uint32_t max_itd_length, uint32_t* extra_split_list, ItdAppended appended) {
	uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
	if (j >= t.n || group_keys[j] == ~0ull) return;
	if (j > 0 && merge_cluster_continues(t, order, j, max_distance, max_itd_length)) return; // not the first of its cluster
	uint32_t end = j + 1;
	while (end < t.n && merge_cluster_continues(t, order, end, max_distance, max_itd_length)) ++end;
	if (end - j > 1) merge_cluster(t, order, j, end, max_distance, max_itd_length, extra_split_list, appended);
}
// after the sweep, only if an internal tandem duplication absorbed another one: the read lists with the appended entries merged in
__global__ void merged_list_size_kernel(CandidateTable t, ItdAppended appended, uint64_t* sizes) {
	const uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c >= t.n) return;
	for (uint32_t list = 0; list < 3; ++list)
		sizes[3 * (uint64_t) c + list] = t.list_offset[3 * (uint64_t) c + list + 1] - t.list_offset[3 * (uint64_t) c + list] + (list < 2 ? appended.length[2 * (uint64_t) c + list] : 0);
}
__global__ void merged_list_copy_kernel(CandidateTable t, ItdAppended appended, const uint64_t* new_offset, uint32_t* new_lists, uint32_t first_candidate) {
	// one wavefront per candidate: the lists of a hot candidate hold hundreds of entries
	const uint32_t c = first_candidate + ((blockIdx.x * BLOCK + threadIdx.x) >> 6), lane = threadIdx.x & 63;
	if (c >= t.n) return;
	// (implicit discordant lists -- fusion_core.hpp: CandidateTable::discordant_before --: only the split-read lists exist, packed; they move, the discordant lists keep their
	//  numbers of entries, so the entries in front of a candidate's stay what they were)
	for (uint32_t list = 0; list < (t.discordant_before != nullptr ? 2u : 3u); ++list) {
		const uint64_t own_begin = t.list_offset[3 * (uint64_t) c + list], own_length = t.list_offset[3 * (uint64_t) c + list + 1] - own_begin;
		uint32_t* out = new_lists + new_offset[3 * (uint64_t) c + list] - (t.discordant_before != nullptr ? t.discordant_before[c] : 0);
		for (uint64_t j = lane; j < own_length; j += 64) out[j] = list < 2 ? split_list_entry(t, c, own_begin + j) : t.read_lists[own_begin + j];
		if (list < 2) {
			const uint32_t* extra = appended.pool + appended.begin[2 * (uint64_t) c + list];
			for (uint32_t j = lane; j < appended.length[2 * (uint64_t) c + list]; j += 64) out[own_length + j] = extra[j];
		}
	}
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
	DeviceBuffer& appended_pool = ctx->scratch("merge.appended_pool"); DeviceBuffer& appended_begin = ctx->scratch("merge.appended_begin"); DeviceBuffer& appended_length = ctx->scratch("merge.appended_length");
	const size_t C1 = std::max<uint32_t>(C, 1);
	ALLOC(keys_in, C1 * 8); ALLOC(keys_out, C1 * 8); ALLOC(order_a, C1 * 4); ALLOC(order_b, C1 * 4); ALLOC(counter, 16); ALLOC(ctx->cand_extra_split_list, C1 * 4);
	HIP_CHECK(hipMemsetAsync(counter.ptr, 0, 16, s));
	HIP_CHECK(hipMemsetAsync(ctx->cand_extra_split_list.ptr, 0, C1 * 4, s));
	// entries that internal tandem duplications take over from the candidates they absorb: rare, a pool of 1 M entries + 1 per list entry (at most 64 M) is ample
	// (a candidate's region is rewritten whenever it absorbs again); counter[1] = pool cursor, counter[2] = overflow flag
	const uint32_t pool_capacity = (uint32_t) std::min<uint64_t>(64u << 20, (uint64_t) ctx->n_list_entries + (1u << 20));
	ALLOC(appended_pool, (size_t) pool_capacity * 4); ALLOC(appended_begin, C1 * 8); ALLOC(appended_length, C1 * 8);
	HIP_CHECK(hipMemsetAsync(appended_begin.ptr, 0, C1 * 8, s));
	HIP_CHECK(hipMemsetAsync(appended_length.ptr, 0, C1 * 8, s));
	ItdAppended appended;
	appended.pool = appended_pool.as<uint32_t>(); appended.capacity = pool_capacity; appended.cursor = counter.as<uint32_t>() + 1; appended.overflow = counter.as<uint32_t>() + 2;
	appended.begin = appended_begin.as<uint32_t>(); appended.length = appended_length.as<uint32_t>();
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
		  merge_cluster_kernel<<<grid_for(C), BLOCK, 0, s>>>(t, order, keys_out.as<uint64_t>(), max_distance, ctx->params.max_itd_length, ctx->cand_extra_split_list.as<uint32_t>(), appended); }
		// did an internal tandem duplication absorb another one?  Then the read lists are rebuilt with the appended entries behind the own ones.
		uint32_t pool_state[2] = { 0, 0 };
		HIP_CHECK(hipMemcpyAsync(pool_state, counter.as<uint32_t>() + 1, 8, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		if (pool_state[1]) { set_last_error("merge_adjacent_fusions: the pool of read-list entries appended to internal tandem duplications is exhausted"); return AGPU_ERR_CAPACITY; }
		if (pool_state[0] > 0) {
			DeviceBuffer& sizes = ctx->scratch("merge.list_sizes"); DeviceBuffer& new_offset = ctx->scratch("merge.list_offset"); DeviceBuffer& new_lists = ctx->scratch("merge.read_lists");
			ALLOC(sizes, (3 * (size_t) C + 1) * 8); ALLOC(new_offset, (3 * (size_t) C + 1) * 8);
			HIP_CHECK(hipMemsetAsync(sizes.as<uint64_t>() + 3 * (size_t) C, 0, 8, s));
			merged_list_size_kernel<<<grid_for(C), BLOCK, 0, s>>>(t, appended, sizes.as<uint64_t>());
			HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, sizes.as<uint64_t>(), new_offset.as<uint64_t>(), (uint64_t) 0, 3 * (size_t) C + 1, rocprim::plus<uint64_t>(), s));
			if (bytes > scratch.capacity) ALLOC(scratch, bytes);
			HIP_CHECK(rocprim::exclusive_scan(scratch.ptr, bytes, sizes.as<uint64_t>(), new_offset.as<uint64_t>(), (uint64_t) 0, 3 * (size_t) C + 1, rocprim::plus<uint64_t>(), s));
			uint64_t total = 0;
			HIP_CHECK(hipMemcpyAsync(&total, new_offset.as<uint64_t>() + 3 * (size_t) C, 8, hipMemcpyDeviceToHost, s));
			HIP_CHECK(hipStreamSynchronize(s));
			uint64_t discordant_entries = 0; // (implicit discordant lists: not stored)
			if (ctx->lists_implicit) HIP_CHECK(hipMemcpy(&discordant_entries, t.discordant_before + C, 8, hipMemcpyDeviceToHost));
			ALLOC(new_lists, std::max<size_t>(total - discordant_entries, 1) * 4);
			{ KernelTimer timer(ctx, "merged_list_copy_kernel", (uint64_t) total * 8);
			  for_each_wave_chunk(C, [&](uint64_t first, uint64_t count) { merged_list_copy_kernel<<<grid_for(count * 64), BLOCK, 0, s>>>(t, appended, new_offset.as<uint64_t>(), new_lists.as<uint32_t>(), (uint32_t) first); }); }
			HIP_CHECK(hipStreamSynchronize(s));
			ctx->cand_list_offset.swap(new_offset); ctx->cand_read_lists.swap(new_lists);
			ctx->candidates.list_offset = ctx->cand_list_offset.as<uint64_t>(); ctx->candidates.read_lists = ctx->cand_read_lists.as<uint32_t>();
			ctx->n_list_entries = total;
			{ const int status = recut_list_windows(ctx); if (status != AGPU_OK) return status; } // (the positions of the lists have moved)
			HIP_CHECK(hipMemsetAsync(ctx->cand_extra_split_list.ptr, 0, C1 * 4, s)); // the appended entries are part of the lists now
		}
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
