// arriba_amd/csrc/device/agpu_multimappers.hip -- filter_multimappers on the device (reference: source/filter_multimappers.cpp:109-221,
// called at source/arriba.cpp:426-429).
//
//   support rank      four stable radix sorts order the candidates by fusion_has_more_support (a strict total order, :79-107)
//   best_rank_kernel  per multi-mapping read: the smallest rank among the candidates that list it (atomicMin)
//   group_kernel      per group of alignments of one read name: alignment scores (CIGAR walk against the genome), keep the best
//   recount_kernel    candidates lose the reads that became multi-mappers
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <string>
#include <rocprim/rocprim.hpp>
#include "agpu_context.hpp"
#include "multimapper_core.hpp"

using namespace agpu;

namespace {

const int BLOCK = 256;
inline unsigned int grid_for(uint64_t n) { return (unsigned int) ((n + BLOCK - 1) / BLOCK); }

#define HIP_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_last_error(std::string(#call) + ": " + hipGetErrorString(e_)); return AGPU_ERR_DEVICE; } } while (0)
#define ALLOC(buffer, bytes) do { if (!(buffer).allocate(bytes)) { set_last_error("hipMalloc failed (" #buffer ")"); return AGPU_ERR_DEVICE; } } while (0)

// sort keys of the candidate at position j of the current order, least significant criterion first (pass 0) to most significant (pass 3)
__global__ void support_key_kernel(AnnotationView ann, CandidateTable t, const uint32_t* order, int pass, uint64_t* keys) {
	uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
	if (j >= t.n) return;
	const uint32_t c = order ? order[j] : j;
	const uint32_t flags = t.flags[c];
	uint64_t key;
	if (pass == 0) key = (uint64_t) t.gene1[c] << 32 | t.gene2[c];
	else if (pass == 1) key = ((flags & CFLAG_UPSTREAM1) ? 2u : 0u) | ((flags & CFLAG_UPSTREAM2) ? 1u : 0u);
	else if (pass == 2) key = (uint64_t) (uint32_t) t.breakpoint1[c] << 32 | (uint32_t) t.breakpoint2[c];
	else {
		const uint64_t supporting_reads = t.split_reads1[c] + t.split_reads2[c] + t.discordant_mates[c];
		const uint64_t not_coding1 = (ann.gene_bits[t.gene1[c]] & GBIT_PROTEIN_CODING) ? 0 : 1, not_coding2 = (ann.gene_bits[t.gene2[c]] & GBIT_PROTEIN_CODING) ? 0 : 1;
		key = (0x1FFFFull - (supporting_reads < 0x1FFFFull ? supporting_reads : 0x1FFFFull)) << 34 | not_coding1 << 33 | not_coding2 << 32 | t.contigs[c]; // contigs = contig1 << 16 | contig2
	}
	keys[j] = key;
}
__global__ void rank_from_order_kernel(const uint32_t* order, uint32_t n, uint32_t* rank) {
	uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
	if (j < n) rank[order[j]] = j;
}
__global__ void best_rank_kernel(BatchView b, CandidateTable t, const uint32_t* rank, uint32_t* best_rank) {
	uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c >= t.n) return;
	const uint32_t* offsets = t.list_offset + 3 * (uint64_t) c;
	const uint32_t mine = rank[c];
	for (uint32_t k = offsets[0]; k < offsets[3]; ++k) {
		const uint32_t read = t.read_lists[k];
		if ((b.fbits[read] & FBIT_MULTIMAPPER) && best_rank[read] > mine) atomicMin(&best_rank[read], mine); // only groups of alignments are ever looked at
	}
}
__global__ void multimapper_group_kernel(BatchView b, AnnotationView ann, GenomeView genome, const uint32_t* best_rank, unsigned int* discarded) {
	uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	uint32_t mine = 0;
	if (i < b.n && (b.fbits[i] & FBIT_MULTIMAPPER) && (i == 0 || b.group[i - 1] != b.group[i])) mine = resolve_multimapper_group(b, ann, genome, best_rank, i);
	for (int offset = 32; offset > 0; offset >>= 1) mine += __shfl_down(mine, offset);
	if ((threadIdx.x & 63) == 0 && mine) atomicAdd(discarded, mine);
}
__global__ void multimapper_recount_kernel(BatchView b, CandidateTable t, unsigned int* remaining) {
	uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	const bool kept = c < t.n && recount_after_multimappers(b, t, c);
	unsigned long long ballot = __ballot(kept);
	if ((threadIdx.x & 63) == 0 && ballot) atomicAdd(remaining, (unsigned int) __popcll(ballot));
}

}

extern "C" int agpu_filter_multimappers(agpu_ctx* ctx, uint64_t* remaining, uint64_t* discarded_reads) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	if (ctx->global_n != 0 && ctx->global_n != ctx->n) { set_last_error("filter_multimappers is not sharded yet: the read lists of a candidate and the reads they name must be in one context"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	const uint64_t n = ctx->n;
	DeviceBuffer& keys_in = ctx->scratch("multimappers.keys_in"); DeviceBuffer& keys_out = ctx->scratch("multimappers.keys_out"); DeviceBuffer& order_a = ctx->scratch("multimappers.order_a");
	DeviceBuffer& order_b = ctx->scratch("multimappers.order_b"); DeviceBuffer& rank = ctx->scratch("multimappers.rank"); DeviceBuffer& best_rank = ctx->scratch("multimappers.best_rank");
	DeviceBuffer& counters = ctx->scratch("multimappers.counters"); DeviceBuffer& scratch = ctx->scratch("multimappers.rocprim");
	const size_t C1 = std::max<uint32_t>(C, 1), n1 = std::max<uint64_t>(n, 1);
	ALLOC(keys_in, C1 * 8); ALLOC(keys_out, C1 * 8); ALLOC(order_a, C1 * 4); ALLOC(order_b, C1 * 4); ALLOC(rank, C1 * 4); ALLOC(best_rank, n1 * 4); ALLOC(counters, 16);
	HIP_CHECK(hipMemsetAsync(best_rank.ptr, 0xFF, n1 * 4, s));
	HIP_CHECK(hipMemsetAsync(counters.ptr, 0, 16, s));
	(void) hipEventRecord(ctx->event_start, s);
	if (C > 0 && n > 0 && ctx->params.filter_enabled[FILTER_multimappers]) {
		const CandidateTable& t = ctx->candidates;
		uint32_t* order = nullptr; uint32_t* next = order_a.as<uint32_t>();
		const int end_bits[4] = { 64, 2, 64, 51 };
		for (int pass = 0; pass < 4; ++pass) { // least significant criterion first; every sort is stable
			support_key_kernel<<<grid_for(C), BLOCK, 0, s>>>(ctx->annotation, t, order, pass, keys_in.as<uint64_t>());
			size_t bytes = 0;
			if (order == nullptr) {
				HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), rocprim::counting_iterator<uint32_t>(0), next, C, 0, end_bits[pass], s));
				if (bytes > scratch.capacity) ALLOC(scratch, bytes);
				KernelTimer timer(ctx, "rocprim::radix_sort_pairs(support rank)", (uint64_t) C * 24);
				HIP_CHECK(rocprim::radix_sort_pairs(scratch.ptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), rocprim::counting_iterator<uint32_t>(0), next, C, 0, end_bits[pass], s));
			} else {
				HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), order, next, C, 0, end_bits[pass], s));
				if (bytes > scratch.capacity) ALLOC(scratch, bytes);
				KernelTimer timer(ctx, "rocprim::radix_sort_pairs(support rank)", (uint64_t) C * 24);
				HIP_CHECK(rocprim::radix_sort_pairs(scratch.ptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), order, next, C, 0, end_bits[pass], s));
			}
			order = next;
			next = (order == order_a.as<uint32_t>()) ? order_b.as<uint32_t>() : order_a.as<uint32_t>();
		}
		rank_from_order_kernel<<<grid_for(C), BLOCK, 0, s>>>(order, C, rank.as<uint32_t>());
		{ KernelTimer timer(ctx, "best_rank_kernel", (uint64_t) ctx->n_list_entries * 5 + (uint64_t) C * 16); best_rank_kernel<<<grid_for(C), BLOCK, 0, s>>>(ctx->batch, t, rank.as<uint32_t>(), best_rank.as<uint32_t>()); }
		{ KernelTimer timer(ctx, "multimapper_group_kernel", n * 6); multimapper_group_kernel<<<grid_for(n), BLOCK, 0, s>>>(ctx->batch, ctx->annotation, ctx->genome, best_rank.as<uint32_t>(), counters.as<unsigned int>()); }
		{ KernelTimer timer(ctx, "multimapper_recount_kernel", (uint64_t) ctx->n_list_entries * 5 + (uint64_t) C * 14); multimapper_recount_kernel<<<grid_for(C), BLOCK, 0, s>>>(ctx->batch, t, counters.as<unsigned int>() + 1); }
	}
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (uint64_t) C * 120 + (uint64_t) ctx->n_list_entries * 10 + n * 6;
	unsigned int host_counters[2] = { 0, 0 };
	HIP_CHECK(hipMemcpy(host_counters, counters.ptr, sizeof(host_counters), hipMemcpyDeviceToHost));
	if (remaining) *remaining = host_counters[1];
	if (discarded_reads) *discarded_reads = host_counters[0];
	return AGPU_OK;
}
