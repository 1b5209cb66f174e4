// arriba_amd/csrc/device/agpu_homologs.hip -- filter_homologs (reference: source/filter_homologs.cpp:68-141, called at source/arriba.cpp:556-560).
// The homology verdicts (k-mer index + genome, homolog_core.hpp) are evaluated on the device, one thread per gene pair, for every pair of genes the
// elimination can ask about; the elimination itself is sequential and runs on the host over the few unfiltered candidates that are left at
// this point of the workflow (it is quadratic in their number in the reference too).
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "agpu_context.hpp"
#include "device_utils.hpp"
#include "homolog_host.hpp"

using namespace agpu;

namespace {

const int BLOCK = 256;
const uint32_t MAX_REMAINING = 4u << 20; // unfiltered candidates gathered to the host (40 bytes each)

#define HIP_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_last_error(std::string(#call) + ": " + hipGetErrorString(e_)); return AGPU_ERR_DEVICE; } } while (0)
#define ALLOC(buffer, bytes) do { if (!(buffer).allocate(bytes)) { set_last_error("hipMalloc failed (" #buffer ")"); return AGPU_ERR_NO_MEMORY; } } while (0)

__global__ void homolog_collect_kernel(CandidateTable t, const uint32_t* iteration_rank, const float* evalue, RemainingCandidate* out, uint32_t capacity, unsigned int* count) {
	const uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c >= t.n || t.filter[c] != FILTER_none) return;
	const uint32_t at = atomicAdd(count, 1u); // few candidates are left at this stage
	if (at >= capacity) return;
	RemainingCandidate r;
	r.candidate = c; r.iteration_rank = iteration_rank[c]; r.gene1 = t.gene1[c]; r.gene2 = t.gene2[c]; r.breakpoint1 = t.breakpoint1[c]; r.breakpoint2 = t.breakpoint2[c];
	r.split_reads1 = t.split_reads1[c]; r.split_reads2 = t.split_reads2[c]; r.discordant_mates = t.discordant_mates[c]; r.evalue = evalue[c];
	out[at] = r;
}
// The verdict of a gene pair (genes_are_homologs, homolog_core.hpp) by a wavefront per pair: the 64 lanes ask 64 positions of the smaller gene at once (the answers are independent
// of each other), then every lane walks the 64 answers in position order with the reference's two early exits (homolog_walk: a few comparisons per position, no memory).  A pair
// of long genes is 10^5 dependent look-ups for one thread -- a launch of a thread per pair waits for the longest pair -- and 10^5 / 64 rounds here (10^7 fragments: 24.4 -> 2.5 ms,
// 10^8: 52.5 -> 19.7 ms; profiles/r03h_output_side_and_ingest.txt).
__global__ void __launch_bounds__(64) homolog_verdict_wave_kernel(AnnotationView ann, GenomeView genome, KmerIndexView kmers, const uint64_t* pairs, uint32_t n_pairs, float max_identity_fraction, uint8_t* verdicts, uint32_t first_pair) {
	const uint32_t k = first_pair + blockIdx.x;
	if (k >= n_pairs) return;
	HomologPair p;
	if (!homolog_pair_setup(ann, genome, kmers, (uint32_t) (pairs[k] >> 32), (uint32_t) pairs[k], p)) { if (threadIdx.x == 0) verdicts[k] = 0; return; }
	uint32_t matching_kmers = 0;
	int verdict = 0;
	for (uint64_t base = 0; base + 2 * KMER_LENGTH < p.size && verdict == 0; base += 64ull * KMER_LENGTH) {
		const uint64_t pos = base + (uint64_t) threadIdx.x * KMER_LENGTH;
		const bool mine = pos + 2 * KMER_LENGTH < p.size && homolog_position_matches(p, kmers, pos);
		const unsigned long long answers = __ballot(mine);
		for (uint32_t lane = 0; lane < 64 && verdict == 0; ++lane) { // (every lane walks the same answers: the verdict is uniform)
			const uint64_t at = base + (uint64_t) lane * KMER_LENGTH;
			if (!(at + 2 * KMER_LENGTH < p.size)) break;
			verdict = homolog_walk(p, max_identity_fraction, at, (answers >> lane) & 1ull, matching_kmers);
		}
	}
	if (threadIdx.x == 0) verdicts[k] = verdict > 0 ? 1 : 0;
}
__global__ void homolog_apply_kernel(CandidateTable t, const uint32_t* candidates, const uint8_t* filters, uint32_t n) {
	const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
	if (k < n) t.filter[candidates[k]] = filters[k];
}

}

extern "C" int agpu_filter_homologs(agpu_ctx* ctx, float max_identity_fraction, uint64_t* remaining) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	if (!ctx->kmer_index_done) { set_last_error("agpu_make_kmer_index must run first"); return AGPU_ERR_INVALID; }
	if (!ctx->evalue_done) { set_last_error("agpu_estimate_expected_fusions must run first (the e-value breaks ties)"); return AGPU_ERR_INVALID; }
	if (!ctx->iteration_order_done) { const int status = agpu_candidate_iteration_order(ctx, nullptr); if (status != AGPU_OK) return status; } // hazard H2
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	DeviceBuffer& counter = ctx->scratch("homologs.counter"); DeviceBuffer& collected = ctx->scratch("homologs.collected");
	const uint32_t capacity = std::min<uint32_t>(std::max<uint32_t>(C, 1), MAX_REMAINING);
	ALLOC(counter, 16); ALLOC(collected, (size_t) capacity * sizeof(RemainingCandidate));
	HIP_CHECK(hipMemsetAsync(counter.ptr, 0, 16, s));
	(void) hipEventRecord(ctx->event_start, s);
	uint32_t n_remaining = 0;
	if (C > 0) {
		homolog_collect_kernel<<<(C + BLOCK - 1) / BLOCK, BLOCK, 0, s>>>(ctx->candidates, ctx->cand_iteration_rank.as<uint32_t>(), ctx->cand_evalue.as<float>(), collected.as<RemainingCandidate>(), capacity, counter.as<unsigned int>());
		HIP_CHECK(hipMemcpyAsync(&n_remaining, counter.ptr, 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
	}
	uint64_t kept = n_remaining;
	if (n_remaining > 0 && ctx->params.filter_enabled[FILTER_homologs]) {
		if (n_remaining > capacity) { set_last_error("filter_homologs: more unfiltered candidates than the elimination on the host is sized for (4 M)"); return AGPU_ERR_CAPACITY; }
		std::vector<RemainingCandidate> list(n_remaining);
		HIP_CHECK(hipMemcpy(list.data(), collected.ptr, (size_t) n_remaining * sizeof(RemainingCandidate), hipMemcpyDeviceToHost));
		HomologElimination elimination;
		elimination.prepare(list);
		const std::vector<uint64_t>& pairs = elimination.pairs;
		DeviceBuffer& device_pairs = ctx->scratch("homologs.pairs"); DeviceBuffer& device_verdicts = ctx->scratch("homologs.verdicts");
		ALLOC(device_pairs, pairs.size() * 8); ALLOC(device_verdicts, pairs.size());
		HIP_CHECK(hipMemcpyAsync(device_pairs.ptr, pairs.data(), pairs.size() * 8, hipMemcpyHostToDevice, s));
		KmerIndexView kmers;
		kmers.contig_table = ctx->kmer_contig_table.as<uint32_t>(); kmers.offsets = ctx->kmer_offsets.as<uint32_t>(); kmers.positions = ctx->kmer_positions.as<int32_t>(); kmers.n_contigs = ctx->genome.n_contigs;
		if (!pairs.empty()) {
			KernelTimer timer(ctx, "homolog_verdict_wave_kernel", pairs.size() * 64);
			for_each_wave_chunk(pairs.size(), [&](uint64_t first, uint64_t count) { homolog_verdict_wave_kernel<<<(unsigned int) count, 64, 0, s>>>(ctx->annotation, ctx->genome, kmers, device_pairs.as<uint64_t>(), (uint32_t) pairs.size(), max_identity_fraction, device_verdicts.as<uint8_t>(), (uint32_t) first); });
		}
		std::vector<uint8_t> host_verdicts(pairs.size());
		HIP_CHECK(hipMemcpyAsync(host_verdicts.data(), device_verdicts.ptr, pairs.size(), hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		kept = elimination.run(host_verdicts);
		const std::vector<uint32_t>& candidates = elimination.candidates;
		const std::vector<uint8_t>& filter = elimination.filter;
		DeviceBuffer& device_candidates = ctx->scratch("homologs.candidates"); DeviceBuffer& device_filters = ctx->scratch("homologs.filters");
		ALLOC(device_candidates, (size_t) n_remaining * 4); ALLOC(device_filters, n_remaining);
		HIP_CHECK(hipMemcpyAsync(device_candidates.ptr, candidates.data(), (size_t) n_remaining * 4, hipMemcpyHostToDevice, s));
		HIP_CHECK(hipMemcpyAsync(device_filters.ptr, filter.data(), n_remaining, hipMemcpyHostToDevice, s));
		homolog_apply_kernel<<<(n_remaining + BLOCK - 1) / BLOCK, BLOCK, 0, s>>>(ctx->candidates, device_candidates.as<uint32_t>(), device_filters.as<uint8_t>(), n_remaining);
		HIP_CHECK(hipStreamSynchronize(s));
	}
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (uint64_t) C * 2;
	if (remaining) *remaining = kept;
	return AGPU_OK;
}
