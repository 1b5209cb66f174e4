// arriba_amd/csrc/device/agpu_evalue.hip -- estimate_expected_fusions + filter_relative_support on the device
// (reference: source/filter_relative_support.cpp:17-224).
//
//   partner_insert/resolve_kernel   first event per (gene, breakpoint1, breakpoint2) in the reference's iteration order: lock-free
//                                   open addressing, the slot holds min(priority << 32 | event handle) of its key (64-bit atomicMin)
//   radix sort + partner_size/count distinct (gene, partner) pairs -> |partners(gene)| -> fusion_partner_count(gene)
//   evalue_globals_kernel           the eight sample-wide counters + per-gene marks (read-through fraction)
//   evalue_kernel                   per candidate: the float/double multiplication chain over host-tabulated pow() factors
//   relative_support_kernel         e-value cutoff
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>
#include <rocprim/rocprim.hpp>
#include "agpu_context.hpp"
#include "device_utils.hpp"
#include "evalue_host.hpp"

using namespace agpu;

namespace {

const int BLOCK = 256;
const unsigned long long EMPTY_SLOT64 = ~0ull;
inline unsigned int grid_for(uint64_t n) { return (unsigned int) ((n + BLOCK - 1) / BLOCK); }

#define HIP_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_last_error(std::string(#call) + ": " + hipGetErrorString(e_)); return AGPU_ERR_DEVICE; } } while (0)
#define ALLOC(buffer, bytes) do { if (!(buffer).allocate(bytes)) { set_last_error("hipMalloc failed (" #buffer ")"); return AGPU_ERR_NO_MEMORY; } } while (0)

__global__ void partner_insert_kernel(CandidateTable t, const uint32_t* iteration_rank, unsigned long long* slots, uint32_t mask) {
	uint32_t handle = blockIdx.x * BLOCK + threadIdx.x;
	if (handle >= 2 * t.n) return;
	uint32_t c = handle >> 1;
	if (!raises_partner_events(t, c)) return;
	PartnerKey key = partner_event_key(t, handle);
	unsigned long long mine = (unsigned long long) (2 * (uint64_t) iteration_rank[c] + (handle & 1)) << 32 | handle;
	uint32_t h = (uint32_t) hash_partner_key(key) & mask;
	while (true) {
		unsigned long long owner = __hip_atomic_load(&slots[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (owner == EMPTY_SLOT64) {
			owner = atomicCAS(&slots[h], EMPTY_SLOT64, mine);
			if (owner == EMPTY_SLOT64) return;
		}
		if (partner_keys_equal(partner_event_key(t, (uint32_t) owner), key)) { if (mine < owner) atomicMin(&slots[h], mine); return; } // a slot never changes its key
		h = (h + 1) & mask;
	}
}

// winners emit (gene << 32 | partner); one atomic per workgroup reserves the output range
__global__ void __launch_bounds__(BLOCK) partner_resolve_kernel(CandidateTable t, const unsigned long long* slots, uint32_t mask, uint64_t* pairs, uint32_t* pair_count) {
	__shared__ uint32_t wave_offset[BLOCK / 64];
	__shared__ uint32_t block_base;
	uint32_t handle = blockIdx.x * BLOCK + threadIdx.x;
	bool winner = false;
	uint64_t pair = 0;
	if (handle < 2 * t.n && raises_partner_events(t, handle >> 1)) {
		PartnerKey key = partner_event_key(t, handle);
		uint32_t h = (uint32_t) hash_partner_key(key) & mask;
		while (true) {
			unsigned long long owner = slots[h];
			if (partner_keys_equal(partner_event_key(t, (uint32_t) owner), key)) {
				winner = (uint32_t) owner == handle;
				pair = (uint64_t) key.gene << 32 | partner_event_partner(t, handle);
				break;
			}
			h = (h + 1) & mask;
		}
	}
	const uint32_t at = block_append<BLOCK>(winner ? 1u : 0u, pair_count, wave_offset, &block_base);
	if (winner) pairs[at] = pair;
}

__global__ void partner_size_kernel(const uint64_t* sorted_pairs, uint32_t n, int32_t* partner_set_size) {
	// the pairs are sorted by gene: the lanes of one gene are contiguous, the first of them adds the distinct pairs of the run in this wavefront
	uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63;
	const bool valid = k < n;
	const uint64_t pair = valid ? sorted_pairs[k] : ~0ull;
	const uint32_t gene = (uint32_t) (pair >> 32);
	const bool distinct = valid && (k == 0 || sorted_pairs[k - 1] != pair);
	const uint32_t previous_gene = __shfl_up(gene, 1);
	const bool head = lane == 0 || previous_gene != gene;
	const unsigned long long heads = __ballot(head), distincts = __ballot(distinct);
	if (head && valid) {
		const unsigned long long later_heads = lane == 63 ? 0ull : heads & ~((2ull << lane) - 1);
		const int run_end = later_heads ? __ffsll((unsigned long long) later_heads) - 1 : 64;
		const unsigned long long run = (run_end == 64 ? ~0ull : (1ull << run_end) - 1) & ~((1ull << lane) - 1);
		const int count = __popcll(distincts & run);
		if (count) atomicAdd(&partner_set_size[gene], count);
	}
}
// fusion_partner_count[gene] = number of partners whose own partner set is not larger (source/filter_relative_support.cpp:33-41)
__global__ void partner_count_kernel(const uint64_t* sorted_pairs, uint32_t n, const int32_t* partner_set_size, int32_t* partner_count) {
	uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
	if (k >= n) return;
	if (k != 0 && sorted_pairs[k - 1] == sorted_pairs[k]) return;
	uint32_t gene = (uint32_t) (sorted_pairs[k] >> 32), partner = (uint32_t) sorted_pairs[k];
	if (partner_set_size[gene] >= partner_set_size[partner]) atomicAdd(&partner_count[gene], 1);
}

__global__ void evalue_globals_kernel(AnnotationView ann, CandidateTable t, unsigned int* counters /* [EG_COUNT + 1]: classes, max supporting reads */, uint8_t* gene_marks) {
	__shared__ unsigned int block_counters[EG_COUNT];
	__shared__ unsigned int block_max;
	if (threadIdx.x < EG_COUNT) block_counters[threadIdx.x] = 0;
	if (threadIdx.x == 0) block_max = 0;
	__syncthreads();
	for (uint32_t c = blockIdx.x * BLOCK + threadIdx.x; c < t.n; c += gridDim.x * BLOCK) {
		EvalueContribution r = evalue_contribution(ann, t, c);
		if (r.breakpoint_class >= 0) atomicAdd(&block_counters[r.breakpoint_class], 1u);
		if (r.intragenic_class >= 0) atomicAdd(&block_counters[r.intragenic_class], 1u);
		if (r.spliced_class >= 0) atomicAdd(&block_counters[r.spliced_class], 1u);
		if (r.marks_genes) {
			uint8_t mark = r.marks_read_through ? 3 : 1;
			uint32_t gene1 = t.gene1[c], gene2 = t.gene2[c];
			if ((gene_marks[gene1] & mark) != mark) atomicOr((unsigned int*) (gene_marks + (gene1 & ~3u)), (unsigned int) mark << ((gene1 & 3u) * 8));
			if ((gene_marks[gene2] & mark) != mark) atomicOr((unsigned int*) (gene_marks + (gene2 & ~3u)), (unsigned int) mark << ((gene2 & 3u) * 8));
		}
		atomicMax(&block_max, t.split_reads1[c] + t.split_reads2[c] + t.discordant_mates[c]);
	}
	__syncthreads();
	if (threadIdx.x < EG_COUNT && block_counters[threadIdx.x]) atomicAdd(&counters[threadIdx.x], block_counters[threadIdx.x]);
	if (threadIdx.x == 0 && block_max) atomicMax(&counters[EG_COUNT], block_max);
}
__global__ void gene_mark_count_kernel(const uint8_t* gene_marks, uint32_t n_genes, unsigned int* counts /* [2]: genes with fusions, with read-through fusions */) {
	uint32_t g = blockIdx.x * BLOCK + threadIdx.x;
	uint8_t mark = g < n_genes ? gene_marks[g] : 0;
	unsigned long long with_fusions = __ballot(mark & 1), with_read_through = __ballot(mark & 2);
	if ((threadIdx.x & 63) == 0) {
		if (with_fusions) atomicAdd(&counts[0], (unsigned int) __popcll(with_fusions));
		if (with_read_through) atomicAdd(&counts[1], (unsigned int) __popcll(with_read_through));
	}
}

__global__ void evalue_kernel(AnnotationView ann, CandidateTable t, const int32_t* partner_count, EvalueGlobals g, EvalueTables tables, float* evalue) {
	uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c >= t.n) return;
	evalue[c] = candidate_evalue(ann, t, c, partner_count, g, tables);
}

__global__ void relative_support_kernel(AnnotationView ann, CandidateTable t, const float* evalue, float evalue_cutoff, unsigned int* remaining) {
	__shared__ uint32_t block_sum;
	uint32_t kept = 0;
	for (uint32_t c = blockIdx.x * BLOCK + threadIdx.x; c < t.n; c += gridDim.x * BLOCK) {
		if (t.filter[c] != FILTER_none) continue;
		if (fails_relative_support(ann, t, c, evalue[c], evalue_cutoff)) t.filter[c] = FILTER_relative_support; else ++kept;
	}
	block_tally(kept, remaining, &block_sum);
}

__global__ void candidate_predicates_kernel(AnnotationView ann, CandidateTable t, const uint8_t* enabled, float exonic_fraction, int32_t min_support, unsigned int* discarded /* [3] */) {
	__shared__ unsigned int block_discarded[3];
	if (threadIdx.x < 3) block_discarded[threadIdx.x] = 0;
	__syncthreads();
	for (uint32_t c = blockIdx.x * BLOCK + threadIdx.x; c < t.n; c += gridDim.x * BLOCK) {
		if (t.filter[c] != FILTER_none) continue;
		const int stage = candidate_predicate_stage(ann, t, c, enabled, exonic_fraction, min_support);
		if (stage < 3) {
			t.filter[c] = stage == 0 ? 14 : stage == 1 ? 15 : 17; // non_coding_neighbors, intragenic_exonic, min_support (source/common.hpp:29-67)
			atomicAdd(&block_discarded[stage], 1u);
		}
	}
	__syncthreads();
	if (threadIdx.x < 3 && block_discarded[threadIdx.x]) atomicAdd(&discarded[threadIdx.x], block_discarded[threadIdx.x]);
}

template <class T> int upload_table(DeviceBuffer& buffer, const std::vector<T>& host, hipStream_t stream) {
	if (!buffer.allocate(host.size() * sizeof(T))) { set_last_error("hipMalloc failed"); return AGPU_ERR_NO_MEMORY; }
	HIP_CHECK(hipMemcpyAsync(buffer.ptr, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice, stream));
	return AGPU_OK;
}

}

extern "C" int agpu_set_candidate_state(agpu_ctx* ctx, const uint8_t* filter, const uint32_t* split_reads1, const uint32_t* split_reads2, const uint32_t* discordant_mates) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	const size_t C = ctx->n_candidates;
	if (C == 0) return AGPU_OK;
	if (filter) HIP_CHECK(hipMemcpyAsync(ctx->cand_filter.ptr, filter, C, hipMemcpyHostToDevice, ctx->stream));
	if (split_reads1) HIP_CHECK(hipMemcpyAsync(ctx->cand_split_reads1.ptr, split_reads1, C * 4, hipMemcpyHostToDevice, ctx->stream));
	if (split_reads2) HIP_CHECK(hipMemcpyAsync(ctx->cand_split_reads2.ptr, split_reads2, C * 4, hipMemcpyHostToDevice, ctx->stream));
	if (discordant_mates) HIP_CHECK(hipMemcpyAsync(ctx->cand_discordant_mates.ptr, discordant_mates, C * 4, hipMemcpyHostToDevice, ctx->stream));
	HIP_CHECK(hipStreamSynchronize(ctx->stream));
	return AGPU_OK;
}

extern "C" int agpu_estimate_expected_fusions(agpu_ctx* ctx, uint64_t mapped_reads, const uint32_t* iteration_rank) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	if (!iteration_rank && !ctx->iteration_order_done) { set_last_error("pass the iteration order or call agpu_candidate_iteration_order first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	const uint32_t n_genes = ctx->n_genes + ctx->n_dummy;
	const CandidateTable& t = ctx->candidates;
	ALLOC(ctx->cand_evalue, (size_t) std::max<uint32_t>(C, 1) * 4);
	ctx->evalue_done = false;
	if (C == 0) { ctx->evalue_done = true; return AGPU_OK; }
	if (2ull * C >= 0xFFFFFFF0ull) { set_last_error("too many candidates for the partner table"); return AGPU_ERR_CAPACITY; }

	DeviceBuffer& rank = ctx->scratch("evalue.rank"); DeviceBuffer& slots = ctx->scratch("evalue.slots"); DeviceBuffer& pairs = ctx->scratch("evalue.pairs"); DeviceBuffer& sorted_pairs = ctx->scratch("evalue.sorted_pairs");
	DeviceBuffer& pair_count = ctx->scratch("evalue.pair_count"); DeviceBuffer& set_size = ctx->scratch("evalue.set_size"); DeviceBuffer& partner_count = ctx->scratch("evalue.partner_count");
	DeviceBuffer& counters = ctx->scratch("evalue.counters"); DeviceBuffer& gene_marks = ctx->scratch("evalue.gene_marks"); DeviceBuffer& scratch = ctx->scratch("evalue.rocprim");
	const uint32_t* device_rank = ctx->cand_iteration_rank.as<uint32_t>();
	if (iteration_rank) {
		ALLOC(rank, (size_t) C * 4);
		HIP_CHECK(hipMemcpyAsync(rank.ptr, iteration_rank, (size_t) C * 4, hipMemcpyHostToDevice, s));
		device_rank = rank.as<uint32_t>();
	}
	uint64_t n_slots = 1024;
	while (n_slots < 4ull * C) n_slots <<= 1; // two events per candidate, load factor <= 0.5
	const uint32_t mask = (uint32_t) (n_slots - 1);
	ALLOC(slots, n_slots * 8); ALLOC(pairs, 2 * (size_t) C * 8); ALLOC(sorted_pairs, 2 * (size_t) C * 8); ALLOC(pair_count, 16);
	ALLOC(set_size, (size_t) n_genes * 4); ALLOC(partner_count, (size_t) n_genes * 4); ALLOC(counters, 16 * 4); ALLOC(gene_marks, ((size_t) n_genes + 8) & ~(size_t) 3);
	HIP_CHECK(hipMemsetAsync(slots.ptr, 0xFF, n_slots * 8, s));
	HIP_CHECK(hipMemsetAsync(pair_count.ptr, 0, 16, s));
	HIP_CHECK(hipMemsetAsync(set_size.ptr, 0, (size_t) n_genes * 4, s));
	HIP_CHECK(hipMemsetAsync(partner_count.ptr, 0, (size_t) n_genes * 4, s));
	HIP_CHECK(hipMemsetAsync(counters.ptr, 0, 16 * 4, s));
	HIP_CHECK(hipMemsetAsync(gene_marks.ptr, 0, gene_marks.bytes, s));

	(void) hipEventRecord(ctx->event_start, s);
	{ KernelTimer timer(ctx, "partner_insert_kernel", (uint64_t) C * (4 + 1 + 16 + 2 * 8)); partner_insert_kernel<<<grid_for(2ull * C), BLOCK, 0, s>>>(t, device_rank, slots.as<unsigned long long>(), mask); }
	{ KernelTimer timer(ctx, "partner_resolve_kernel", (uint64_t) C * (1 + 16 + 2 * 8 + 2 * 8)); partner_resolve_kernel<<<grid_for(2ull * C), BLOCK, 0, s>>>(t, slots.as<unsigned long long>(), mask, pairs.as<uint64_t>(), pair_count.as<uint32_t>()); }
	{ KernelTimer timer(ctx, "evalue_globals_kernel", (uint64_t) C * 34); evalue_globals_kernel<<<tally_grid(C, BLOCK), BLOCK, 0, s>>>(ctx->annotation, t, counters.as<unsigned int>(), gene_marks.as<uint8_t>()); }
	gene_mark_count_kernel<<<grid_for(n_genes), BLOCK, 0, s>>>(gene_marks.as<uint8_t>(), n_genes, counters.as<unsigned int>() + 12);
	uint32_t n_pairs = 0;
	unsigned int host_counters[16];
	HIP_CHECK(hipMemcpyAsync(&n_pairs, pair_count.ptr, 4, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipMemcpyAsync(host_counters, counters.ptr, sizeof(host_counters), hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	if (n_pairs > 0) {
		size_t bytes = 0;
		HIP_CHECK(rocprim::radix_sort_keys(nullptr, bytes, pairs.as<uint64_t>(), sorted_pairs.as<uint64_t>(), n_pairs, 0, 64, s));
		if (bytes > scratch.capacity) ALLOC(scratch, bytes);
		{ KernelTimer timer(ctx, "rocprim::radix_sort_keys(partner pairs)", (uint64_t) n_pairs * 16);
		  HIP_CHECK(rocprim::radix_sort_keys(scratch.ptr, bytes, pairs.as<uint64_t>(), sorted_pairs.as<uint64_t>(), n_pairs, 0, 64, s)); }
		partner_size_kernel<<<grid_for(n_pairs), BLOCK, 0, s>>>(sorted_pairs.as<uint64_t>(), n_pairs, set_size.as<int32_t>());
		partner_count_kernel<<<grid_for(n_pairs), BLOCK, 0, s>>>(sorted_pairs.as<uint64_t>(), n_pairs, set_size.as<int32_t>(), partner_count.as<int32_t>());
	}

	// sample-wide covariates with the reference's fallbacks for small samples; pow() factors tabulated with the host's libm
	EvalueGlobals g = make_evalue_globals(host_counters, host_counters[12], host_counters[13]);
	ctx->evalue_globals = g;
	const unsigned int max_support = host_counters[EG_COUNT];
	EvalueHostTables host_tables;
	host_tables.build_support_tables(mapped_reads, max_support);
	if (upload_table(ctx->evalue_support_scale, host_tables.support_scale, s) != AGPU_OK || upload_table(ctx->evalue_intragenic_support, host_tables.intragenic_support, s) != AGPU_OK ||
	    upload_table(ctx->evalue_intergenic_support, host_tables.intergenic_support, s) != AGPU_OK) return AGPU_ERR_DEVICE;
	if (ctx->evalue_distance_tables.ptr == nullptr) { // independent of the sample: built once per context
		host_tables.build_distance_tables();
		if (upload_table(ctx->evalue_distance_tables, host_tables.distances, s) != AGPU_OK) return AGPU_ERR_DEVICE;
	}
	HIP_CHECK(hipStreamSynchronize(s)); // the host vectors go out of scope
	EvalueTables tables = evalue_table_view(ctx->evalue_support_scale.as<double>(), ctx->evalue_intragenic_support.as<double>(), ctx->evalue_intergenic_support.as<double>(), max_support, ctx->evalue_distance_tables.as<double>());

	{ KernelTimer timer(ctx, "evalue_kernel", (uint64_t) C * (34 + 4)); evalue_kernel<<<grid_for(C), BLOCK, 0, s>>>(ctx->annotation, t, partner_count.as<int32_t>(), g, tables, ctx->cand_evalue.as<float>()); }
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (uint64_t) C * (34 + 4 + 4 + 2 * 24);
	ctx->evalue_done = true;
	return AGPU_OK;
}

extern "C" int agpu_get_evalues(agpu_ctx* ctx, float* evalue) {
	if (!ctx || !ctx->evalue_done || !evalue) { set_last_error("agpu_estimate_expected_fusions must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	HIP_CHECK(hipStreamSynchronize(ctx->stream));
	if (ctx->n_candidates) HIP_CHECK(hipMemcpy(evalue, ctx->cand_evalue.ptr, (size_t) ctx->n_candidates * 4, hipMemcpyDeviceToHost));
	return AGPU_OK;
}

extern "C" int agpu_filter_relative_support(agpu_ctx* ctx, uint64_t* remaining) {
	if (!ctx || !ctx->evalue_done) { set_last_error("agpu_estimate_expected_fusions must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	DeviceBuffer& counter = ctx->scratch("evalue.remaining");
	ALLOC(counter, 16);
	HIP_CHECK(hipMemsetAsync(counter.ptr, 0, 16, s));
	(void) hipEventRecord(ctx->event_start, s);
	if (C > 0 && ctx->params.filter_enabled[FILTER_relative_support]) {
		KernelTimer timer(ctx, "relative_support_kernel", (uint64_t) C * (1 + 4 + 24 + 1));
		relative_support_kernel<<<tally_grid(C, BLOCK), BLOCK, 0, s>>>(ctx->annotation, ctx->candidates, ctx->cand_evalue.as<float>(), ctx->params.evalue_cutoff, counter.as<unsigned int>());
	}
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (uint64_t) C * 30;
	unsigned int kept = 0;
	HIP_CHECK(hipMemcpy(&kept, counter.ptr, 4, hipMemcpyDeviceToHost));
	if (remaining) *remaining = kept;
	return AGPU_OK;
}

extern "C" int agpu_filter_candidate_predicates(agpu_ctx* ctx, uint64_t* remaining) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	DeviceBuffer& counters = ctx->scratch("evalue.predicate_counters");
	ALLOC(counters, 8 * 4);
	HIP_CHECK(hipMemsetAsync(counters.ptr, 0, 8 * 4, s));
	(void) hipEventRecord(ctx->event_start, s);
	if (C > 0) {
		// unfiltered candidates before the stage: counted on the device to keep the call self-contained
		KernelTimer timer(ctx, "candidate_predicates_kernel", (uint64_t) C * 40);
		candidate_predicates_kernel<<<tally_grid(C, BLOCK), BLOCK, 0, s>>>(ctx->annotation, ctx->candidates, ctx->filter_enabled.as<uint8_t>(), ctx->params.exonic_fraction, (int32_t) ctx->params.min_support, counters.as<unsigned int>());
	}
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (uint64_t) C * 40;
	unsigned int discarded[3] = { 0, 0, 0 };
	HIP_CHECK(hipMemcpy(discarded, counters.ptr, sizeof(discarded), hipMemcpyDeviceToHost));
	if (remaining) { remaining[0] = discarded[0]; remaining[1] = discarded[1]; remaining[2] = discarded[2]; } // numbers discarded per stage; the caller knows how many were unfiltered before
	return AGPU_OK;
}
