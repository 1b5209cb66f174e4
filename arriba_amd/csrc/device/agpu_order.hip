// arriba_amd/csrc/device/agpu_order.hip -- iteration order of the reference's candidate container on the device (hazard H2).
//
// The reference keeps its candidates in std::unordered_map<8-tuple, fusion_t> (source/common.hpp:286) with the tuple hash of
// source/common.hpp:294-314, and several stages depend on the order in which that container iterates (e.g. the partner dedup of
// source/filter_relative_support.cpp:22-29).  With libstdc++ that order is a pure function of the hash codes, the insertion order
// (== candidate index) and the bucket counts chosen by std::__detail::_Prime_rehash_policy:
//
//   * a node inserted into a non-empty bucket goes to the front of that bucket's chain; a node inserted into an empty bucket goes
//     to the front of the whole list (_M_insert_bucket_begin);
//   * a rehash walks the list in its current order and re-inserts every node by the same rule (_M_rehash_aux).
//
// Hence, for one bucket count B, if S is the sequence "list order at the rehash, then the nodes inserted until the next rehash",
// the new list is S grouped by bucket (hash % B), buckets by DEScending first occurrence in S, nodes within a bucket by descending
// position in S.  Each phase is therefore one scatter-min (first occurrence per bucket) + one radix sort; the phases double in size,
// so the total work is about three times the final size.  The bucket counts come from the host's own _Prime_rehash_policy object,
// i.e. from the very libstdc++ the reference would be linked against.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include <rocprim/rocprim.hpp>
#include "agpu_context.hpp"
#include "order_host.hpp"

using namespace agpu;

namespace {

const int BLOCK = 256;
inline unsigned int grid_for(uint64_t n) { return (unsigned int) ((n + BLOCK - 1) / BLOCK); }

#define HIP_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_last_error(std::string(#call) + ": " + hipGetErrorString(e_)); return AGPU_ERR_DEVICE; } } while (0)
#define ALLOC(buffer, bytes) do { if (!(buffer).allocate(bytes)) { set_last_error("hipMalloc failed (" #buffer ")"); return AGPU_ERR_NO_MEMORY; } } while (0)

// the reference's tuple hash: hash(element k) ^ (hash of the rest << 4), std::hash of an integer = its value (int sign-extends)
__global__ void candidate_hash_kernel(CandidateTable t, uint64_t* hashes) {
	uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c >= t.n) return;
	uint32_t flags = t.flags[c], contigs = t.contigs[c];
	uint64_t h = (flags & CFLAG_UPSTREAM2) ? 1 : 0;
	h = (uint64_t) ((flags & CFLAG_UPSTREAM1) ? 1 : 0) ^ h << 4;
	h = (uint64_t) (int64_t) t.breakpoint2[c] ^ h << 4;
	h = (uint64_t) (int64_t) t.breakpoint1[c] ^ h << 4;
	h = (uint64_t) (contigs & 0xFFFF) ^ h << 4;
	h = (uint64_t) (contigs >> 16) ^ h << 4;
	h = (uint64_t) t.gene2[c] ^ h << 4;
	h = (uint64_t) t.gene1[c] ^ h << 4;
	hashes[c] = h;
}

// element at position idx of the phase's sequence S: the old list order for idx < carried, then the new insertions
__device__ __forceinline__ uint32_t sequence_element(const uint32_t* order, uint32_t carried, uint32_t idx) { return idx < carried ? order[idx] : idx; }

__global__ void first_occurrence_kernel(const uint64_t* hashes, const uint32_t* order, uint32_t carried, uint32_t n, uint64_t buckets, uint32_t* first) {
	uint32_t idx = blockIdx.x * BLOCK + threadIdx.x;
	if (idx >= n) return;
	atomicMin(&first[hashes[sequence_element(order, carried, idx)] % buckets], idx);
}
__global__ void phase_key_kernel(const uint64_t* hashes, const uint32_t* order, uint32_t carried, uint32_t n, uint64_t buckets, const uint32_t* first, uint32_t bits, uint64_t* keys, uint32_t* elements) {
	uint32_t idx = blockIdx.x * BLOCK + threadIdx.x;
	if (idx >= n) return;
	uint32_t element = sequence_element(order, carried, idx);
	uint32_t bucket_first = first[hashes[element] % buckets];
	keys[idx] = (uint64_t) (n - 1 - bucket_first) << bits | (n - 1 - idx); // ascending key == descending first occurrence, then descending position
	elements[idx] = element;
}
__global__ void rank_from_order_kernel(const uint32_t* order, uint32_t n, uint32_t* rank) {
	uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
	if (i < n) rank[order[i]] = i;
}

}

extern "C" int agpu_candidate_iteration_order(agpu_ctx* ctx, uint32_t* iteration_rank) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	ALLOC(ctx->cand_iteration_rank, (size_t) std::max<uint32_t>(C, 1) * 4);
	ctx->iteration_order_done = false;
	if (C == 0) { ctx->iteration_order_done = true; return AGPU_OK; }
	const std::vector<RehashPhase> phases = rehash_schedule(C);
	DeviceBuffer& hashes = ctx->scratch("order.hashes"); DeviceBuffer& order_a = ctx->scratch("order.a"); DeviceBuffer& order_b = ctx->scratch("order.b");
	DeviceBuffer& keys_in = ctx->scratch("order.keys_in"); DeviceBuffer& keys_out = ctx->scratch("order.keys_out"); DeviceBuffer& elements = ctx->scratch("order.elements");
	DeviceBuffer& first = ctx->scratch("order.first"); DeviceBuffer& sort_scratch = ctx->scratch("order.rocprim");
	ALLOC(hashes, (size_t) C * 8); ALLOC(order_a, (size_t) C * 4); ALLOC(order_b, (size_t) C * 4); ALLOC(keys_in, (size_t) C * 8); ALLOC(keys_out, (size_t) C * 8); ALLOC(elements, (size_t) C * 4);
	ALLOC(first, (size_t) phases.back().buckets * 4);
	(void) hipEventRecord(ctx->event_start, s);
	{ KernelTimer timer(ctx, "candidate_hash_kernel", (uint64_t) C * (24 + 8)); candidate_hash_kernel<<<grid_for(C), BLOCK, 0, s>>>(ctx->candidates, hashes.as<uint64_t>()); }
	uint32_t* order = order_a.as<uint32_t>(); uint32_t* next_order = order_b.as<uint32_t>();
	for (size_t k = 0; k < phases.size(); ++k) {
		const uint32_t carried = phases[k].begin;                                        // nodes already in the list when the rehash happens
		const uint32_t n = (k + 1 < phases.size()) ? phases[k + 1].begin : C;            // nodes in the list when the phase ends
		uint32_t bits = 1;
		while ((1ull << bits) < n) ++bits;
		HIP_CHECK(hipMemsetAsync(first.ptr, 0xFF, (size_t) phases[k].buckets * 4, s));
		KernelTimer timer(ctx, "iteration_order_phase", (uint64_t) n * (8 + 4 + 4 + 12 + 24 + 4));
		first_occurrence_kernel<<<grid_for(n), BLOCK, 0, s>>>(hashes.as<uint64_t>(), order, carried, n, phases[k].buckets, first.as<uint32_t>());
		phase_key_kernel<<<grid_for(n), BLOCK, 0, s>>>(hashes.as<uint64_t>(), order, carried, n, phases[k].buckets, first.as<uint32_t>(), bits, keys_in.as<uint64_t>(), elements.as<uint32_t>());
		size_t bytes = 0;
		HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), elements.as<uint32_t>(), next_order, n, 0, 2 * bits, s));
		if (bytes > sort_scratch.capacity) ALLOC(sort_scratch, bytes + (bytes >> 2));
		HIP_CHECK(rocprim::radix_sort_pairs(sort_scratch.ptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), elements.as<uint32_t>(), next_order, n, 0, 2 * bits, s));
		std::swap(order, next_order);
	}
	rank_from_order_kernel<<<grid_for(C), BLOCK, 0, s>>>(order, C, ctx->cand_iteration_rank.as<uint32_t>());
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (uint64_t) C * (24 + 8 + 4);
	ctx->iteration_order_done = true;
	if (iteration_rank) HIP_CHECK(hipMemcpy(iteration_rank, ctx->cand_iteration_rank.ptr, (size_t) C * 4, hipMemcpyDeviceToHost));
	return AGPU_OK;
}
