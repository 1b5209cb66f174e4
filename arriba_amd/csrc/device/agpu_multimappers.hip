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
#include <vector>
#include <rocprim/rocprim.hpp>
#include "agpu_context.hpp"
#include "multimapper_core.hpp"
#include "device_utils.hpp"

using namespace agpu;

namespace {

const int BLOCK = 256;
inline unsigned int grid_for(uint64_t n) { return (unsigned int) ((n + BLOCK - 1) / BLOCK); }

#define HIP_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_last_error(std::string(#call) + ": " + hipGetErrorString(e_)); return AGPU_ERR_DEVICE; } } while (0)
#define ALLOC(buffer, bytes) do { if (!(buffer).allocate(bytes)) { set_last_error("hipMalloc failed (" #buffer ")"); return AGPU_ERR_NO_MEMORY; } } while (0)

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
// ---- bitmaps over the reads ------------------------------------------------------------------------------------------------------------
// A byte per read (10 MB at 10 M reads) thrashes the L2 when 4 x 10^8 list entries look their read up; a bit per read (1.25 MB) stays
// resident in the L2 of every XCD.
enum { BITS_MULTIMAPPER = 0, BITS_DISCARDED = 1, BITS_FROM_BYTES = 2 };
__global__ void read_bitmap_kernel(BatchView b, const uint8_t* bytes, uint64_t n, int what, uint32_t* words) {
	const uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	bool bit = false;
	if (i < n) bit = what == BITS_MULTIMAPPER ? (b.fbits[i] & FBIT_MULTIMAPPER) != 0 : what == BITS_DISCARDED ? b.filter[i] == FILTER_multimappers : bytes[i] != 0;
	const unsigned long long ballot = __ballot(bit);
	if ((threadIdx.x & 63) == 0 && i < n) { words[i >> 5] = (uint32_t) ballot; if (i + 32 < ((n + 31) & ~31ull)) words[(i >> 5) + 1] = (uint32_t) (ballot >> 32); }
}
// ... of the replicated states of the reads of a sample whose reads are sharded over the ranks (agpu_sharded.hip): a bit where (byte & mask) == value
__global__ void state_bitmap_kernel(const uint8_t* bytes, uint64_t n, uint8_t mask, uint8_t value, uint32_t* words) {
	const uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	const bool bit = i < n && (bytes[i] & mask) == value;
	const unsigned long long ballot = __ballot(bit);
	if ((threadIdx.x & 63) == 0 && i < n) { words[i >> 5] = (uint32_t) ballot; if (i + 32 < ((n + 31) & ~31ull)) words[(i >> 5) + 1] = (uint32_t) (ballot >> 32); }
}
const int LIST_UNROLL = 4;
__device__ __forceinline__ bool bitmap_test(const uint32_t* words, uint32_t read) { return (words[read >> 5] >> (read & 31)) & 1u; }
__global__ void bitmap_popcount_kernel(const uint32_t* words, uint64_t n_words, uint32_t* counts) {
	const uint64_t w = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (w < n_words) counts[w] = __popc(words[w]);
}

// ---- the scans over the read lists ---------------------------------------------------------------------------------------------------
// One wavefront owns 64 consecutive candidates, i.e. one contiguous range of list entries.  The lanes stride over the entries (coalesced),
// test the read's bit, and only for the few hits find the candidate the entry belongs to: the lane whose candidate ends behind the entry
// (one ballot over the lanes' end offsets).  `global_index` != NULL: the lists belong to the candidates this rank built and the
// candidate table is the replicated one (sharded form).

// best[target(read)] = min rank over the candidates that list the multi-mapping read; target = the read itself, or its ordinal among the
// multi-mapping reads of the sample when word_prefix != NULL
// (first: the kernel looks at the candidates [first, n_listed) -- a window of them when the discordant lists are implicit, for_each_list_window)
__global__ void __launch_bounds__(BLOCK) list_best_rank_kernel(uint32_t first, uint32_t n_listed, const uint64_t* list_offset, const uint32_t* read_lists, const uint32_t* global_index, const uint32_t* rank,
                                                                 const uint32_t* multimapper_bits, const uint32_t* word_prefix, uint32_t* best) {
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t c0 = first + ((blockIdx.x * BLOCK + threadIdx.x) & ~63u);
	if (c0 >= n_listed) return;
	const uint32_t c = c0 + lane, last = min(c0 + 64, n_listed);
	const uint64_t my_end = c < n_listed ? list_offset[3 * (uint64_t) c + 3] : ~0ull;
	const uint32_t my_rank = c < n_listed ? rank[global_index ? global_index[c] : c] : 0;
	const uint64_t begin = list_offset[3 * (uint64_t) c0], end = list_offset[3 * (uint64_t) last];
	for (uint64_t base = begin; base < end; base += 64 * LIST_UNROLL) { // LIST_UNROLL x 64 entries in flight: the loop is bound by the latency of two dependent loads
		uint32_t read[LIST_UNROLL]; bool hit[LIST_UNROLL];
		AGPU_UNROLL for (int u = 0; u < LIST_UNROLL; ++u) { const uint64_t k = base + 64 * u + lane; read[u] = k < end ? read_lists[k] : 0xFFFFFFFFu; }
		AGPU_UNROLL for (int u = 0; u < LIST_UNROLL; ++u) hit[u] = read[u] != 0xFFFFFFFFu && bitmap_test(multimapper_bits, read[u]);
		AGPU_UNROLL for (int u = 0; u < LIST_UNROLL; ++u) {
			unsigned long long hits = __ballot(hit[u]);
			while (hits) {
				const int l = __ffsll((unsigned long long) hits) - 1;
				hits &= hits - 1;
				const uint64_t entry = base + 64 * u + l;
				const int owner = __ffsll((unsigned long long) __ballot(my_end > entry)) - 1;
				const uint32_t owner_rank = __shfl(my_rank, owner);
				if ((int) lane == l) {
					const uint32_t target = word_prefix ? word_prefix[read[u] >> 5] + __popc(multimapper_bits[read[u] >> 5] & ((1u << (read[u] & 31)) - 1)) : read[u];
					if (best[target] > owner_rank) atomicMin(&best[target], owner_rank);
				}
			}
		}
	}
}

// reference :188-211: the counters of a candidate lose the reads that became multi-mappers.  finalize: candidates left without supporting reads get
// the filter `multimappers`, the others are counted (single context); otherwise only the counters are lowered (owner in the sharded form).
__global__ void __launch_bounds__(BLOCK) list_recount_kernel(CandidateTable t, uint32_t first, uint32_t n_listed, const uint64_t* list_offset, const uint32_t* read_lists, const uint32_t* global_index,
                                                               const uint32_t* discarded_bits, bool finalize, unsigned int* remaining) {
	__shared__ uint32_t block_sum;
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t c0 = first + ((blockIdx.x * BLOCK + threadIdx.x) & ~63u);
	uint32_t kept = 0;
	if (c0 < n_listed) {
		const uint32_t c = c0 + lane, last = min(c0 + 64, n_listed);
		const bool valid = c < n_listed;
		const uint32_t g = valid ? (global_index ? global_index[c] : c) : 0;
		const uint64_t end1 = valid ? list_offset[3 * (uint64_t) c + 1] : ~0ull, end2 = valid ? list_offset[3 * (uint64_t) c + 2] : ~0ull, end3 = valid ? list_offset[3 * (uint64_t) c + 3] : ~0ull;
		uint32_t lost1 = 0, lost2 = 0, lost3 = 0;
		const uint64_t begin = list_offset[3 * (uint64_t) c0], end = list_offset[3 * (uint64_t) last];
		const uint64_t previous_end = __shfl_up((unsigned long long) end3, 1);
		const uint64_t my_begin = lane == 0 ? begin : previous_end;
		for (uint64_t base = begin; base < end; base += 64 * LIST_UNROLL) {
			uint32_t read[LIST_UNROLL]; bool hit[LIST_UNROLL];
			AGPU_UNROLL for (int u = 0; u < LIST_UNROLL; ++u) { const uint64_t k = base + 64 * u + lane; read[u] = k < end ? read_lists[k] : 0xFFFFFFFFu; }
			AGPU_UNROLL for (int u = 0; u < LIST_UNROLL; ++u) hit[u] = read[u] != 0xFFFFFFFFu && bitmap_test(discarded_bits, read[u]);
			AGPU_UNROLL for (int u = 0; u < LIST_UNROLL; ++u) {
				unsigned long long hits = __ballot(hit[u]);
				while (hits) {
					const uint64_t entry = base + 64 * u + (uint32_t) (__ffsll((unsigned long long) hits) - 1);
					hits &= hits - 1;
					if (entry >= my_begin && entry < end3) { if (entry < end1) ++lost1; else if (entry < end2) ++lost2; else ++lost3; }
				}
			}
		}
		if (valid && t.filter[g] == FILTER_none) {
			const uint32_t count1 = t.split_reads1[g], count2 = t.split_reads2[g], count3 = t.discordant_mates[g];
			if (count1 + count2 + count3 == 0) kept = 1; // candidates without supporting reads are not looked at
			else {
				const uint32_t new1 = count1 > lost1 ? count1 - lost1 : 0, new2 = count2 > lost2 ? count2 - lost2 : 0, new3 = count3 > lost3 ? count3 - lost3 : 0;
				if (lost1) t.split_reads1[g] = new1;
				if (lost2) t.split_reads2[g] = new2;
				if (lost3) t.discordant_mates[g] = new3;
				if (new1 + new2 + new3 == 0) { if (finalize) t.filter[g] = FILTER_multimappers; } else kept = 1;
			}
		}
	}
	if (finalize) block_tally(kept, remaining, &block_sum);
}

// ---- the groups of alignments -----------------------------------------------------------------------------------------------------------
// 3 % of the fragments are multi-mappers: they are compacted first, so that the CIGAR walks against the genome run in dense wavefronts.
__global__ void __launch_bounds__(1024) select_multimappers_kernel(BatchView b, uint32_t* selected, uint32_t* count) {
	__shared__ uint32_t wave_offset[16];
	__shared__ uint32_t block_base;
	const uint64_t i = blockIdx.x * 1024ull + threadIdx.x;
	const bool keep = i < b.n && (b.fbits[i] & FBIT_MULTIMAPPER);
	const uint32_t at = block_append<1024>(keep ? 1u : 0u, count, wave_offset, &block_base);
	if (keep) selected[at] = (uint32_t) i;
}
// one thread per summand of the alignment score (MATE1, MATE2, supplementary); the three threads of a fragment sit in three different wavefronts
// (part-major order), so a wavefront walks alignments of one kind
__global__ void multimapper_score_kernel(BatchView b, AnnotationView ann, GenomeView genome, const uint32_t* selected, const uint32_t* n_selected, int32_t* scores) {
	const uint64_t thread = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	const uint32_t n = *n_selected;
	if (thread >= 3ull * n) return;
	const int part = (int) (thread / n);
	const uint32_t i = selected[thread - (uint64_t) part * n];
	const int32_t score = alignment_score_part(b, ann, genome, i, part);
	if (score != 0) atomicAdd(&scores[i], score);
}
__global__ void multimapper_group_kernel(BatchView b, AnnotationView ann, GenomeView genome, const uint32_t* selected, const uint32_t* n_selected, const uint32_t* best_rank, const int32_t* scores, unsigned int* discarded) {
	__shared__ uint32_t block_sum;
	uint32_t mine = 0;
	const uint32_t n = *n_selected;
	for (uint32_t k = blockIdx.x * BLOCK + threadIdx.x; k < n; k += gridDim.x * BLOCK) {
		const uint32_t i = selected[k];
		if (i == 0 || b.group[i - 1] != b.group[i]) mine += resolve_multimapper_group(b, ann, genome, best_rank, i, scores);
	}
	block_tally(mine, discarded, &block_sum);
}

// rank[c] = position of candidate c under fusion_has_more_support (0 = most support): four stable radix sorts, least significant criterion first
int compute_support_rank(agpu_ctx* ctx, uint32_t* rank) {
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	DeviceBuffer& keys_in = ctx->scratch("multimappers.keys_in"); DeviceBuffer& keys_out = ctx->scratch("multimappers.keys_out"); DeviceBuffer& order_a = ctx->scratch("multimappers.order_a");
	DeviceBuffer& order_b = ctx->scratch("multimappers.order_b"); DeviceBuffer& scratch = ctx->scratch("multimappers.rocprim");
	ALLOC(keys_in, (size_t) C * 8); ALLOC(keys_out, (size_t) C * 8); ALLOC(order_a, (size_t) C * 4); ALLOC(order_b, (size_t) C * 4);
	const CandidateTable& t = ctx->candidates;
	uint32_t* order = nullptr; uint32_t* next = order_a.as<uint32_t>();
	const int end_bits[4] = { 64, 2, 64, 51 };
	for (int pass = 0; pass < 4; ++pass) {
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
	rank_from_order_kernel<<<grid_for(C), BLOCK, 0, s>>>(order, C, rank);
	return AGPU_OK;
}

// ---- the sharded form: the candidate table is replicated, the read lists are with the owners of the gene pairs, the reads with their shards ----

__global__ void multimapper_flag_kernel(BatchView b, uint8_t* flags) {
	uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i < b.n) flags[i] = (b.fbits[i] & FBIT_MULTIMAPPER) ? 1 : 0;
}
__global__ void local_best_rank_kernel(BatchView b, const uint32_t* multimapper_bits, const uint32_t* word_prefix, const uint32_t* best, uint32_t* best_rank) {
	uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i >= b.n) return;
	uint32_t value = NO_FUSION;
	if (b.fbits[i] & FBIT_MULTIMAPPER) { const uint32_t read = (uint32_t) (b.first_rank + i); value = best[word_prefix[read >> 5] + __popc(multimapper_bits[read >> 5] & ((1u << (read & 31)) - 1))]; }
	best_rank[i] = value;
}
__global__ void multimapper_discarded_kernel(BatchView b, uint8_t* discarded) {
	uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i < b.n) discarded[i] = b.filter[i] == FILTER_multimappers ? 1 : 0;
}
// owner: the counters of an owned candidate lose the reads that became multi-mappers (reference :188-211); had_support marks what the reference looks at
__global__ void had_support_kernel(CandidateTable t, uint8_t* had_support) {
	uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c < t.n) had_support[c] = t.filter[c] == FILTER_none && t.split_reads1[c] + t.split_reads2[c] + t.discordant_mates[c] != 0;
}
__global__ void multimapper_finish_kernel(CandidateTable t, const int32_t* counters, const uint8_t* had_support, unsigned int* remaining) {
	__shared__ uint32_t block_sum;
	uint32_t kept = 0;
	for (uint32_t c = blockIdx.x * BLOCK + threadIdx.x; c < t.n; c += gridDim.x * BLOCK) {
		t.split_reads1[c] = (uint32_t) counters[c]; t.split_reads2[c] = (uint32_t) counters[(uint64_t) t.n + c]; t.discordant_mates[c] = (uint32_t) counters[2 * (uint64_t) t.n + c];
		if (t.filter[c] != FILTER_none) continue;
		if (had_support[c] && counters[c] + counters[(uint64_t) t.n + c] + counters[2 * (uint64_t) t.n + c] == 0) t.filter[c] = FILTER_multimappers; else ++kept;
	}
	block_tally(kept, remaining, &block_sum);
}

// shared by the single-context and the sharded form: bitmap of the multi-mapping reads / of the reads discarded as multi-mappers
int build_read_bitmap(agpu_ctx* ctx, DeviceBuffer& words, int what, const uint8_t* bytes, uint64_t n) {
	ALLOC(words, ((n + 31) / 32 + 2) * 4);
	if (n > 0) read_bitmap_kernel<<<grid_for(n), BLOCK, 0, ctx->stream>>>(ctx->batch, bytes, n, what, words.as<uint32_t>());
	return AGPU_OK;
}
// compacts the multi-mapping fragments of this context, scores them, resolves the groups; counters[0] += fragments discarded
int resolve_groups(agpu_ctx* ctx, const uint32_t* best_rank, unsigned int* counters) {
	hipStream_t s = ctx->stream;
	const uint64_t n = ctx->n;
	DeviceBuffer& selected = ctx->scratch("multimappers.selected"); DeviceBuffer& scores = ctx->scratch("multimappers.scores");
	ALLOC(selected, n * 4); ALLOC(scores, n * 4);
	HIP_CHECK(hipMemsetAsync(scores.ptr, 0, n * 4, s));
	unsigned int* n_selected = counters + 2;
	{ KernelTimer timer(ctx, "select_multimappers_kernel", n * 2); select_multimappers_kernel<<<(unsigned int) ((n + 1023) / 1024), 1024, 0, s>>>(ctx->batch, selected.as<uint32_t>(), n_selected); }
	{ KernelTimer timer(ctx, "multimapper_score_kernel", n * 12); multimapper_score_kernel<<<grid_for(3 * n), BLOCK, 0, s>>>(ctx->batch, ctx->annotation, ctx->genome, selected.as<uint32_t>(), n_selected, scores.as<int32_t>()); }
	{ KernelTimer timer(ctx, "multimapper_group_kernel", n * 1);
	  multimapper_group_kernel<<<tally_grid(n / 8 + 1, BLOCK), BLOCK, 0, s>>>(ctx->batch, ctx->annotation, ctx->genome, selected.as<uint32_t>(), n_selected, best_rank, scores.as<int32_t>(), counters); }
	return AGPU_OK;
}

}

extern "C" int agpu_filter_multimappers(agpu_ctx* ctx, uint64_t* remaining, uint64_t* discarded_reads) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	if (ctx->global_n != 0 && ctx->global_n != ctx->n) { set_last_error("this context holds one shard of the sample: use agpu_multimappers_begin ... agpu_multimappers_finish"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	const uint64_t n = ctx->n;
	DeviceBuffer& rank = ctx->scratch("multimappers.rank"); DeviceBuffer& best_rank = ctx->scratch("multimappers.best_rank"); DeviceBuffer& counters = ctx->scratch("multimappers.counters");
	DeviceBuffer& bits = ctx->scratch("multimappers.bits");
	const size_t C1 = std::max<uint32_t>(C, 1), n1 = std::max<uint64_t>(n, 1);
	ALLOC(rank, C1 * 4); ALLOC(best_rank, n1 * 4); ALLOC(counters, 16);
	HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t) best_rank.ptr, (int) NO_FUSION, n1, s));
	HIP_CHECK(hipMemsetAsync(counters.ptr, 0, 16, s));
	(void) hipEventRecord(ctx->event_start, s);
	if (n > 0 && ctx->params.filter_enabled[FILTER_multimappers]) { // (the groups of alignments are resolved even when there is no candidate at all, as in the reference)
		const CandidateTable& t = ctx->candidates;
		if (C > 0) {
			{ const int status = compute_support_rank(ctx, rank.as<uint32_t>()); if (status != AGPU_OK) return status; }
			{ const int status = build_read_bitmap(ctx, bits, BITS_MULTIMAPPER, nullptr, n); if (status != AGPU_OK) return status; }
			const int status = for_each_list_window(ctx, [&](const CandidateTable& window, uint32_t begin, uint32_t end) -> int {
				KernelTimer timer(ctx, "list_best_rank_kernel", (uint64_t) ctx->n_list_entries * 4 + (uint64_t) C * 8);
				list_best_rank_kernel<<<grid_for(end - begin), BLOCK, 0, s>>>(begin, end, window.list_offset, window.read_lists, nullptr, rank.as<uint32_t>(), bits.as<uint32_t>(), nullptr, best_rank.as<uint32_t>());
				return AGPU_OK;
			});
			if (status != AGPU_OK) return status;
		}
		{ const int status = resolve_groups(ctx, best_rank.as<uint32_t>(), counters.as<unsigned int>()); if (status != AGPU_OK) return status; }
		if (C > 0) {
			{ const int status = build_read_bitmap(ctx, bits, BITS_DISCARDED, nullptr, n); if (status != AGPU_OK) return status; }
			const int status = for_each_list_window(ctx, [&](const CandidateTable& window, uint32_t begin, uint32_t end) -> int {
				KernelTimer timer(ctx, "list_recount_kernel", (uint64_t) ctx->n_list_entries * 4 + (uint64_t) C * 26);
				list_recount_kernel<<<grid_for(end - begin), BLOCK, 0, s>>>(window, begin, end, window.list_offset, window.read_lists, nullptr, bits.as<uint32_t>(), true, counters.as<unsigned int>() + 1);
				return AGPU_OK;
			}, LISTS_OF_UNFILTERED, true); // (the counters of a filtered candidate are not touched: its lists need not be expanded, but the kernel strides over every entry of the window)
			if (status != AGPU_OK) return status;
		}
	}
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (uint64_t) C * 120 + (uint64_t) ctx->n_list_entries * 8 + n * 6;
	unsigned int host_counters[2] = { 0, 0 };
	HIP_CHECK(hipMemcpy(host_counters, counters.ptr, sizeof(host_counters), hipMemcpyDeviceToHost));
	if (remaining) *remaining = host_counters[1];
	if (discarded_reads) *discarded_reads = host_counters[0];
	return AGPU_OK;
}

// ---- sharded entry points: every rank calls them in this order, with the collectives named in between ------------------

extern "C" int agpu_set_owned_candidates(agpu_ctx* ctx, const uint32_t* global_index, uint64_t n_owned) {
	if (!ctx || !ctx->fusions_done || !ctx->candidates_imported) { set_last_error("agpu_import_candidates must run first"); return AGPU_ERR_INVALID; }
	if (n_owned != ctx->n_owned) { set_last_error("the number of owned candidates differs from the table this context built"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	ALLOC(ctx->owned_global_index, std::max<uint64_t>(n_owned, 1) * 4);
	if (n_owned) HIP_CHECK(hipMemcpyAsync(ctx->owned_global_index.ptr, global_index, n_owned * 4, hipMemcpyDefault, ctx->stream));
	HIP_CHECK(hipStreamSynchronize(ctx->stream));
	ctx->owned_index_set = true;
	return AGPU_OK;
}

extern "C" int agpu_copy_multimapper_flags(agpu_ctx* ctx, uint8_t* flags) {
	if (!ctx || !ctx->have_batch) { set_last_error("no batch"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	DeviceBuffer& local = ctx->scratch("multimappers.local_flags");
	ALLOC(local, std::max<uint64_t>(ctx->n, 1));
	if (ctx->n) {
		multimapper_flag_kernel<<<grid_for(ctx->n), BLOCK, 0, ctx->stream>>>(ctx->batch, local.as<uint8_t>());
		HIP_CHECK(hipMemcpyAsync(flags, local.ptr, ctx->n, hipMemcpyDefault, ctx->stream));
	}
	HIP_CHECK(hipStreamSynchronize(ctx->stream));
	return AGPU_OK;
}

// global_flags: the flags of all shards in rank order (after the all-gather).  Ranks the replicated candidates, numbers the multi-mapping reads.
extern "C" int agpu_multimappers_begin(agpu_ctx* ctx, const uint8_t* global_flags, uint64_t* n_multimappers) {
	if (!ctx || !ctx->fusions_done || !ctx->candidates_imported || !ctx->owned_index_set || ctx->global_n == 0) { set_last_error("agpu_import_candidates and agpu_set_owned_candidates must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint64_t N = ctx->global_n, n_words = (N + 31) / 32;
	DeviceBuffer& bytes = ctx->scratch("multimappers.global_flags"); DeviceBuffer& bits = ctx->scratch("multimappers.bits"); DeviceBuffer& popcounts = ctx->scratch("multimappers.popcounts");
	DeviceBuffer& word_prefix = ctx->scratch("multimappers.word_prefix"); DeviceBuffer& rank = ctx->scratch("multimappers.rank"); DeviceBuffer& scratch = ctx->scratch("multimappers.rocprim");
	ALLOC(bytes, N + 1); ALLOC(popcounts, (n_words + 1) * 4); ALLOC(word_prefix, (n_words + 1) * 4); ALLOC(rank, std::max<uint32_t>(ctx->n_candidates, 1) * 4);
	HIP_CHECK(hipMemcpyAsync(bytes.ptr, global_flags, N, hipMemcpyDefault, s));
	{ const int status = build_read_bitmap(ctx, bits, BITS_FROM_BYTES, bytes.as<uint8_t>(), N); if (status != AGPU_OK) return status; }
	HIP_CHECK(hipMemsetAsync(popcounts.as<uint32_t>() + n_words, 0, 4, s));
	if (n_words) bitmap_popcount_kernel<<<grid_for(n_words), BLOCK, 0, s>>>(bits.as<uint32_t>(), n_words, popcounts.as<uint32_t>());
	size_t temporary = 0;
	HIP_CHECK(rocprim::exclusive_scan(nullptr, temporary, popcounts.as<uint32_t>(), word_prefix.as<uint32_t>(), 0u, n_words + 1, rocprim::plus<uint32_t>(), s));
	if (temporary > scratch.capacity) ALLOC(scratch, temporary);
	HIP_CHECK(rocprim::exclusive_scan(scratch.ptr, temporary, popcounts.as<uint32_t>(), word_prefix.as<uint32_t>(), 0u, n_words + 1, rocprim::plus<uint32_t>(), s));
	uint32_t total = 0;
	HIP_CHECK(hipMemcpyAsync(&total, word_prefix.as<uint32_t>() + n_words, 4, hipMemcpyDeviceToHost, s));
	if (ctx->n_candidates > 0 && ctx->params.filter_enabled[FILTER_multimappers]) { const int status = compute_support_rank(ctx, rank.as<uint32_t>()); if (status != AGPU_OK) return status; }
	HIP_CHECK(hipStreamSynchronize(s));
	collect_kernel_samples(ctx);
	ctx->n_multimappers_global = total; ctx->multimappers_begun = true;
	if (n_multimappers) *n_multimappers = total;
	return AGPU_OK;
}

// best[n_multimappers] (int32, 0x7FFFFFFF = in no list): this rank's contribution; all-reduce MIN over the ranks follows
extern "C" int agpu_multimappers_partial_best(agpu_ctx* ctx, int32_t* best) {
	if (!ctx || !ctx->multimappers_begun) { set_last_error("agpu_multimappers_begin must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint64_t M = ctx->n_multimappers_global;
	DeviceBuffer& partial = ctx->scratch("multimappers.partial_best");
	ALLOC(partial, std::max<uint64_t>(M, 1) * 4);
	if (M == 0) return AGPU_OK;
	HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t) partial.ptr, (int) NO_FUSION, M, s));
	if (ctx->n_owned > 0 && ctx->params.filter_enabled[FILTER_multimappers]) {
		KernelTimer timer(ctx, "list_best_rank_kernel", (uint64_t) ctx->n_owned_list_entries * 4 + (uint64_t) ctx->n_owned * 12);
		list_best_rank_kernel<<<grid_for(ctx->n_owned), BLOCK, 0, s>>>(0, ctx->n_owned, ctx->owned_list_offset.as<uint64_t>(), ctx->owned_read_lists.as<uint32_t>(), ctx->owned_global_index.as<uint32_t>(),
			ctx->scratch("multimappers.rank").as<uint32_t>(), ctx->scratch("multimappers.bits").as<uint32_t>(), ctx->scratch("multimappers.word_prefix").as<uint32_t>(), partial.as<uint32_t>());
	}
	HIP_CHECK(hipMemcpyAsync(best, partial.ptr, M * 4, hipMemcpyDefault, s));
	HIP_CHECK(hipStreamSynchronize(s));
	collect_kernel_samples(ctx);
	return AGPU_OK;
}

// best: the all-reduced table.  Resolves the groups of this shard; discarded_flags[n] = 1 where the fragment now has the filter `multimappers`
// (all-gather over the ranks follows); *discarded = fragments newly filtered in this shard.
extern "C" int agpu_multimappers_resolve(agpu_ctx* ctx, const int32_t* best, uint8_t* discarded_flags, uint64_t* discarded) {
	if (!ctx || !ctx->multimappers_begun) { set_last_error("agpu_multimappers_begin must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint64_t M = ctx->n_multimappers_global, n = ctx->n;
	DeviceBuffer& reduced = ctx->scratch("multimappers.partial_best"); DeviceBuffer& best_rank = ctx->scratch("multimappers.best_rank"); DeviceBuffer& counters = ctx->scratch("multimappers.counters");
	DeviceBuffer& local = ctx->scratch("multimappers.local_flags");
	ALLOC(reduced, std::max<uint64_t>(M, 1) * 4); ALLOC(best_rank, std::max<uint64_t>(n, 1) * 4); ALLOC(counters, 16); ALLOC(local, std::max<uint64_t>(n, 1));
	if (M) HIP_CHECK(hipMemcpyAsync(reduced.ptr, best, M * 4, hipMemcpyDefault, s));
	HIP_CHECK(hipMemsetAsync(counters.ptr, 0, 16, s));
	if (n > 0) {
		if (ctx->params.filter_enabled[FILTER_multimappers]) {
			local_best_rank_kernel<<<grid_for(n), BLOCK, 0, s>>>(ctx->batch, ctx->scratch("multimappers.bits").as<uint32_t>(), ctx->scratch("multimappers.word_prefix").as<uint32_t>(), reduced.as<uint32_t>(), best_rank.as<uint32_t>());
			const int status = resolve_groups(ctx, best_rank.as<uint32_t>(), counters.as<unsigned int>()); if (status != AGPU_OK) return status;
		}
		multimapper_discarded_kernel<<<grid_for(n), BLOCK, 0, s>>>(ctx->batch, local.as<uint8_t>());
		HIP_CHECK(hipMemcpyAsync(discarded_flags, local.ptr, n, hipMemcpyDefault, s));
	}
	unsigned int host_counter = 0;
	HIP_CHECK(hipMemcpyAsync(&host_counter, counters.ptr, 4, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	collect_kernel_samples(ctx);
	if (discarded) *discarded = host_counter;
	return AGPU_OK;
}

// global_discarded: the flags of all shards in rank order.  The owners lower the counters of their candidates; counters[3 * C] (split_reads1,
// split_reads2, discordant_mates of the replicated table as int32) is this rank's view; all-reduce MIN over the ranks follows.
extern "C" int agpu_multimappers_recount(agpu_ctx* ctx, const uint8_t* global_discarded, int32_t* counters) {
	if (!ctx || !ctx->multimappers_begun) { set_last_error("agpu_multimappers_begin must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint64_t N = ctx->global_n; const uint32_t C = ctx->n_candidates;
	DeviceBuffer& bytes = ctx->scratch("multimappers.global_flags"); DeviceBuffer& bits = ctx->scratch("multimappers.bits"); DeviceBuffer& had_support = ctx->scratch("multimappers.had_support");
	ALLOC(bytes, N + 1); ALLOC(had_support, std::max<uint32_t>(C, 1));
	HIP_CHECK(hipMemcpyAsync(bytes.ptr, global_discarded, N, hipMemcpyDefault, s));
	{ const int status = build_read_bitmap(ctx, bits, BITS_FROM_BYTES, bytes.as<uint8_t>(), N); if (status != AGPU_OK) return status; }
	if (C > 0) {
		const CandidateTable& t = ctx->candidates;
		had_support_kernel<<<grid_for(C), BLOCK, 0, s>>>(t, had_support.as<uint8_t>());
		if (ctx->n_owned > 0 && ctx->params.filter_enabled[FILTER_multimappers]) {
			KernelTimer timer(ctx, "list_recount_kernel", (uint64_t) ctx->n_owned_list_entries * 4 + (uint64_t) ctx->n_owned * 30);
			list_recount_kernel<<<grid_for(ctx->n_owned), BLOCK, 0, s>>>(t, 0, ctx->n_owned, ctx->owned_list_offset.as<uint64_t>(), ctx->owned_read_lists.as<uint32_t>(), ctx->owned_global_index.as<uint32_t>(),
				bits.as<uint32_t>(), false, nullptr);
		}
		HIP_CHECK(hipMemcpyAsync(counters, t.split_reads1, (size_t) C * 4, hipMemcpyDefault, s));
		HIP_CHECK(hipMemcpyAsync(counters + C, t.split_reads2, (size_t) C * 4, hipMemcpyDefault, s));
		HIP_CHECK(hipMemcpyAsync(counters + 2 * (size_t) C, t.discordant_mates, (size_t) C * 4, hipMemcpyDefault, s));
	}
	HIP_CHECK(hipStreamSynchronize(s));
	collect_kernel_samples(ctx);
	return AGPU_OK;
}

// counters: the all-reduced columns.  Candidates that had supporting reads and lost all of them get the filter `multimappers`.
extern "C" int agpu_multimappers_finish(agpu_ctx* ctx, const int32_t* counters, uint64_t* remaining) {
	if (!ctx || !ctx->multimappers_begun) { set_last_error("agpu_multimappers_begin must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	DeviceBuffer& reduced = ctx->scratch("multimappers.reduced_counters"); DeviceBuffer& tally = ctx->scratch("multimappers.counters");
	ALLOC(reduced, std::max<uint32_t>(C, 1) * 12ull); ALLOC(tally, 16);
	HIP_CHECK(hipMemsetAsync(tally.ptr, 0, 16, s));
	if (C > 0) {
		HIP_CHECK(hipMemcpyAsync(reduced.ptr, counters, (size_t) C * 12, hipMemcpyDefault, s));
		multimapper_finish_kernel<<<tally_grid(C, BLOCK), BLOCK, 0, s>>>(ctx->candidates, reduced.as<int32_t>(), ctx->scratch("multimappers.had_support").as<uint8_t>(), tally.as<unsigned int>());
	}
	unsigned int kept = 0;
	HIP_CHECK(hipMemcpyAsync(&kept, tally.ptr, 4, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	ctx->multimappers_begun = false;
	if (remaining) *remaining = kept;
	return AGPU_OK;
}

// ---- the reads sharded over the ranks, every read list on every rank (include/arriba_gpu.h: agpu_shard_keep; agpu_sharded.hip) ------------------------------------------------
// The best candidate of every multi-mapping read of the sample is found from the lists without an exchange (they are all here); the alignment scores and the choice inside a group
// of alignments are made where the reads are (the parts are cut between read names: a group is never split); the reads that lost travel as states (agpu_read_state_export / _import),
// and the recount reads the replicated filters.

extern "C" int agpu_filter_multimappers_resolve(agpu_ctx* ctx, uint64_t* discarded_reads) {
	if (!ctx || !ctx->fusions_done || !ctx->read_sharded || !ctx->state_imported) { set_last_error("agpu_shard_keep, agpu_find_fusions_from_emissions and agpu_read_state_import must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	const uint64_t n = ctx->n, N = ctx->global_n, n_words = (N + 31) / 32;
	DeviceBuffer& rank = ctx->scratch("multimappers.rank"); DeviceBuffer& best_rank = ctx->scratch("multimappers.best_rank"); DeviceBuffer& counters = ctx->scratch("multimappers.counters");
	DeviceBuffer& bits = ctx->scratch("multimappers.bits"); DeviceBuffer& popcounts = ctx->scratch("multimappers.popcounts"); DeviceBuffer& word_prefix = ctx->scratch("multimappers.word_prefix");
	DeviceBuffer& best = ctx->scratch("multimappers.partial_best"); DeviceBuffer& scratch = ctx->scratch("multimappers.rocprim");
	const size_t C1 = std::max<uint32_t>(C, 1), n1 = std::max<uint64_t>(n, 1);
	ALLOC(rank, C1 * 4); ALLOC(best_rank, n1 * 4); ALLOC(counters, 16); ALLOC(bits, (n_words + 2) * 4); ALLOC(popcounts, (n_words + 1) * 4); ALLOC(word_prefix, (n_words + 1) * 4);
	HIP_CHECK(hipMemsetAsync(counters.ptr, 0, 16, s));
	(void) hipEventRecord(ctx->event_start, s);
	if (ctx->params.filter_enabled[FILTER_multimappers]) {
		// the multi-mapping reads of the sample, numbered: the table of the best candidates holds an entry for each of them (3 % of the fragments) instead of one per fragment
		if (N > 0) state_bitmap_kernel<<<grid_for(N), BLOCK, 0, s>>>(ctx->scratch("sharded.bits").as<uint8_t>(), N, WALK_MULTIMAPPER, WALK_MULTIMAPPER, bits.as<uint32_t>());
		HIP_CHECK(hipMemsetAsync(popcounts.as<uint32_t>() + n_words, 0, 4, s));
		if (n_words) bitmap_popcount_kernel<<<grid_for(n_words), BLOCK, 0, s>>>(bits.as<uint32_t>(), n_words, popcounts.as<uint32_t>());
		size_t temporary = 0;
		HIP_CHECK(rocprim::exclusive_scan(nullptr, temporary, popcounts.as<uint32_t>(), word_prefix.as<uint32_t>(), 0u, n_words + 1, rocprim::plus<uint32_t>(), s));
		if (temporary > scratch.capacity) ALLOC(scratch, temporary);
		HIP_CHECK(rocprim::exclusive_scan(scratch.ptr, temporary, popcounts.as<uint32_t>(), word_prefix.as<uint32_t>(), 0u, n_words + 1, rocprim::plus<uint32_t>(), s));
		uint32_t total = 0;
		HIP_CHECK(hipMemcpyAsync(&total, word_prefix.as<uint32_t>() + n_words, 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		ALLOC(best, std::max<uint32_t>(total, 1) * 4ull);
		HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t) best.ptr, (int) NO_FUSION, std::max<uint32_t>(total, 1), s));
		if (C > 0 && total > 0) {
			{ const int status = compute_support_rank(ctx, rank.as<uint32_t>()); if (status != AGPU_OK) return status; }
			const int status = for_each_list_window(ctx, [&](const CandidateTable& window, uint32_t begin, uint32_t end) -> int {
				KernelTimer timer(ctx, "list_best_rank_kernel", (uint64_t) ctx->n_list_entries * 4 + (uint64_t) C * 8);
				list_best_rank_kernel<<<grid_for(end - begin), BLOCK, 0, s>>>(begin, end, window.list_offset, window.read_lists, nullptr, rank.as<uint32_t>(), bits.as<uint32_t>(), word_prefix.as<uint32_t>(), best.as<uint32_t>());
				return AGPU_OK;
			});
			if (status != AGPU_OK) return status;
		}
		if (n > 0) { // the groups of alignments of the reads this context holds (their bit in the bitmap is their own multi-mapper flag)
			local_best_rank_kernel<<<grid_for(n), BLOCK, 0, s>>>(ctx->batch, bits.as<uint32_t>(), word_prefix.as<uint32_t>(), best.as<uint32_t>(), best_rank.as<uint32_t>());
			const int status = resolve_groups(ctx, best_rank.as<uint32_t>(), counters.as<unsigned int>()); if (status != AGPU_OK) return status;
		}
	}
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (uint64_t) C * 60 + (uint64_t) ctx->n_list_entries * 4 + n * 6;
	unsigned int host_counter = 0;
	HIP_CHECK(hipMemcpy(&host_counter, counters.ptr, 4, hipMemcpyDeviceToHost));
	ctx->state_imported = false; // (the filters of reads have changed where they live: the candidates are judged behind the next exchange of the states)
	if (discarded_reads) *discarded_reads = host_counter;
	return AGPU_OK;
}

extern "C" int agpu_filter_multimappers_recount(agpu_ctx* ctx, uint64_t* remaining) {
	if (!ctx || !ctx->fusions_done || !ctx->read_sharded || !ctx->state_imported) { set_last_error("agpu_filter_multimappers_resolve and agpu_read_state_import must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	const uint64_t N = ctx->global_n, n_words = (N + 31) / 32;
	DeviceBuffer& counters = ctx->scratch("multimappers.counters"); DeviceBuffer& bits = ctx->scratch("multimappers.bits");
	ALLOC(counters, 16); ALLOC(bits, (n_words + 2) * 4);
	HIP_CHECK(hipMemsetAsync(counters.ptr, 0, 16, s));
	(void) hipEventRecord(ctx->event_start, s);
	if (C > 0 && N > 0 && ctx->params.filter_enabled[FILTER_multimappers]) {
		state_bitmap_kernel<<<grid_for(N), BLOCK, 0, s>>>(ctx->scratch("sharded.filter").as<uint8_t>(), N, 0xFF, FILTER_multimappers, bits.as<uint32_t>());
		const int status = for_each_list_window(ctx, [&](const CandidateTable& window, uint32_t begin, uint32_t end) -> int {
			KernelTimer timer(ctx, "list_recount_kernel", (uint64_t) ctx->n_list_entries * 4 + (uint64_t) C * 26);
			list_recount_kernel<<<grid_for(end - begin), BLOCK, 0, s>>>(window, begin, end, window.list_offset, window.read_lists, nullptr, bits.as<uint32_t>(), true, counters.as<unsigned int>() + 1);
			return AGPU_OK;
		}, LISTS_OF_UNFILTERED, true);
		if (status != AGPU_OK) return status;
	}
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (uint64_t) C * 60 + (uint64_t) ctx->n_list_entries * 4;
	unsigned int host_counters[2] = { 0, 0 };
	HIP_CHECK(hipMemcpy(host_counters, counters.ptr, sizeof(host_counters), hipMemcpyDeviceToHost));
	if (remaining) *remaining = host_counters[1];
	return AGPU_OK;
}
