// arriba_amd/csrc/device/agpu_events.hip -- event-level predicates behind filter_relative_support on the device:
// filter_both_intronic, filter_short_anchor, filter_end_to_end_fusions, filter_no_coverage (event_core.hpp), and the upload of coverage_t.
// One thread per candidate; the tallies go through one atomic per workgroup (device_utils.hpp).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>
#include <rocprim/rocprim.hpp>
#include "agpu_context.hpp"
#include "event_core.hpp"
#include "device_utils.hpp"
#include "in_vitro_host.hpp"
#include "range_rule_host.hpp"
#include "genomic_support_core.hpp"

using namespace agpu;

namespace {

const int BLOCK = 256;
// A candidate whose read lists hold more than LONG_LIST entries is walked by a wavefront (event_core.hpp: ListLanes): the thread that meets it notes it, a second kernel takes the
// noted ones.  (Lists hold up to -U reads each: 300 by default, 32 767 in BASELINE.json's config 3, where in_vitro_kernel, both_spliced_reads_kernel and the both-intronic
// predicate were 14 s of a 48 s sample with one thread per candidate -- profiles/r05g.)
// (what a wavefront costs more than a thread is the part of the predicate around the walk, which 64 lanes then do for one candidate instead of for 64: with the lists of the default
//  threshold, -U 300, a wavefront per candidate of more than 192 entries made both-intronic and in vitro SLOWER at 10^8 fragments of config 2 -- 42 -> 99 ms, 74 -> 90 ms,
//  profiles/r05h_bench100m.json -- and both spliced faster, 84 -> 69 ms)
const uint32_t LONG_LIST = 1024, LONG_LIST_BOTH_SPLICED = 192;
// (ARRIBA_LONG_LIST = n, tests: lists of more than n entries count as long, so that the kernels of the wavefronts run on the toy samples of the test tier, too)
inline uint32_t long_list_entries(uint32_t by_default) { const char* knob = getenv("ARRIBA_LONG_LIST"); return knob != nullptr && atoi(knob) >= 0 && knob[0] != 0 ? (uint32_t) atoi(knob) : by_default; }
struct WaveLanes {
	uint32_t lane, lanes;
	__device__ WaveLanes() : lane(threadIdx.x & 63), lanes(64) {}
	__device__ uint32_t sum(uint32_t mine) const { for (int offset = 32; offset > 0; offset >>= 1) mine += __shfl_xor(mine, offset); return mine; }
	__device__ bool any(bool mine) const { return __ballot(mine) != 0; }
};
__device__ __forceinline__ uint64_t list_entries_of(const CandidateTable& t, uint32_t c, int first_list) { return t.list_offset[3 * (uint64_t) c + 3] - t.list_offset[3 * (uint64_t) c + first_list]; }

#define HIP_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_last_error(std::string(#call) + ": " + hipGetErrorString(e_)); return AGPU_ERR_DEVICE; } } while (0)
#define ALLOC(buffer, bytes) do { if (!(buffer).allocate(bytes)) { set_last_error("hipMalloc failed (" #buffer ")"); return AGPU_ERR_NO_MEMORY; } } while (0)

// (the candidates [first, end): all of them, or a window of them when the stage walks read lists and the discordant lists are implicit -- for_each_list_window)
__global__ void event_predicate_kernel(int stage, BatchView b, AnnotationView ann, GenomeView genome, CoverageView coverage, CandidateTable t, uint32_t min_anchor_length, unsigned int* remaining, uint32_t first = 0, uint32_t end = 0xFFFFFFFFu,
                                       uint32_t* long_list = nullptr, uint32_t* n_long = nullptr, uint32_t long_entries = LONG_LIST) {
	__shared__ uint32_t block_sum;
	uint32_t kept = 0;
	if (end > t.n) end = t.n;
	for (uint32_t c = first + blockIdx.x * BLOCK + threadIdx.x; c < end; c += gridDim.x * BLOCK) {
		if (t.filter[c] != FILTER_none) continue;
		if (long_list != nullptr && stage == EVENT_both_intronic && list_entries_of(t, c, 0) > long_entries) { long_list[atomicAdd(n_long, 1u)] = c; continue; } // (event_predicate_wave_kernel)
		const uint8_t verdict = event_predicate(stage, b, ann, genome, coverage, t, c, min_anchor_length);
		if (verdict == FILTER_none) ++kept; else if (verdict != EVENT_KEPT_UNCOUNTED) t.filter[c] = verdict;
	}
	block_tally(kept, remaining, &block_sum);
}

__global__ void __launch_bounds__(BLOCK) event_predicate_wave_kernel(int stage, BatchView b, AnnotationView ann, GenomeView genome, CoverageView coverage, CandidateTable t, uint32_t min_anchor_length, unsigned int* remaining,
                                                                     const uint32_t* long_list, const uint32_t* n_long, uint32_t first_wave) {
	const uint32_t wave = first_wave + ((blockIdx.x * BLOCK + threadIdx.x) >> 6);
	if (wave >= *n_long) return;
	const uint32_t c = long_list[wave];
	const WaveLanes lanes;
	const uint8_t verdict = event_predicate(stage, b, ann, genome, coverage, t, c, min_anchor_length, lanes); // (the same for every lane)
	if (lanes.lane != 0) return;
	if (verdict == FILTER_none) atomicAdd(remaining, 1u); else if (verdict != EVENT_KEPT_UNCOUNTED) t.filter[c] = verdict;
}

// select_most_supported_breakpoints: sort keys and the fold over the groups
__global__ void select_best_key_kernel(CandidateTable t, const uint32_t* order, const uint32_t* iteration_rank, int pass, uint64_t* keys) {
	uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
	if (j >= t.n) return;
	const uint32_t c = order ? order[j] : j;
	keys[j] = pass == 0 ? (uint64_t) iteration_rank[c] : select_best_group_key(t, c);
}
// A group of up to SELECT_BEST_SMALL candidates is folded by the thread that finds its head; a larger one -- a hot gene pair of a deep sample holds 20 000 candidates, and one thread
// walking them was the whole 57 ms of the kernel at 10^8 fragments -- is listed for a wavefront of its own (select_best_wave_kernel).
const uint32_t SELECT_BEST_SMALL = 48;
struct GroupRange { uint32_t begin, end; };
__global__ void select_best_group_kernel(CandidateTable t, const uint32_t* order, const uint64_t* group_keys, unsigned int* remaining, GroupRange* large_groups, uint32_t* n_large, uint32_t small) {
	__shared__ uint32_t block_sum;
	uint32_t heads = 0;
	for (uint32_t j = blockIdx.x * BLOCK + threadIdx.x; j < t.n; j += gridDim.x * BLOCK) {
		if (group_keys[j] == ~0ull || (j > 0 && group_keys[j - 1] == group_keys[j])) continue; // filtered, or not the first of its group
		uint32_t end = j + 1;
		while (end < t.n && end - j <= small && group_keys[end] == group_keys[j]) ++end;
		if (end - j > small) { // its end is found by the wavefront that takes it
			GroupRange range; range.begin = j; range.end = 0;
			large_groups[atomicAdd(n_large, 1u)] = range;
		} else if (end - j > 1) select_best_in_group(t, order, j, end);
		++heads; // one candidate per group stays
	}
	block_tally(heads, remaining, &block_sum);
}
// The fold of the reference (source/select_best.cpp:33-60) replaces the best so far by a candidate of a higher (rank, supporting reads) pair whatever else holds, never by one of a lower pair,
// and decides only between equal pairs by the rules that depend on the order.  So the candidate that stays is the fold, in iteration order, over the candidates that have the HIGHEST pair of the
// group, starting with the first of them: whatever stood before it is replaced by it, and nothing of a lower pair replaces what follows.  The wavefront takes the maximum of the pairs over the
// group (strided, one reduction), then folds the few candidates that have it in the order of their positions (ballots over chunks of 64), then marks all others.
__global__ void __launch_bounds__(BLOCK) select_best_wave_kernel(CandidateTable t, const uint32_t* order, const uint64_t* group_keys, const GroupRange* large_groups, const uint32_t* n_large, uint32_t first_wave) {
	const uint32_t wave = first_wave + ((blockIdx.x * BLOCK + threadIdx.x) >> 6), lane = threadIdx.x & 63;
	if (wave >= *n_large) return;
	const uint32_t begin = large_groups[wave].begin;
	const uint64_t group = group_keys[begin];
	uint32_t end = begin + 1; // the end of the group: 64 keys at a time
	while (true) {
		const uint32_t j = end + lane;
		const unsigned long long same = __ballot(j < t.n && group_keys[j] == group);
		if (same == ~0ull) { end += 64; continue; }
		end += (uint32_t) __ffsll((unsigned long long) ~same) - 1;
		break;
	}
	unsigned long long highest = 0;
	for (uint32_t j = begin + lane; j < end; j += 64) {
		const uint32_t c = order[j];
		const unsigned long long pair = (unsigned long long) select_best_rank(t, c) << 32 | (t.split_reads1[c] + t.split_reads2[c] + t.discordant_mates[c]);
		if (pair > highest) highest = pair;
	}
	for (int offset = 32; offset > 0; offset >>= 1) { const unsigned long long other = __shfl_xor(highest, offset); if (other > highest) highest = other; }
	uint32_t best = 0xFFFFFFFFu;
	for (uint32_t base = begin; base < end; base += 64) {
		const uint32_t j = base + lane;
		uint32_t c = 0; bool candidate = false;
		if (j < end) { c = order[j]; candidate = ((unsigned long long) select_best_rank(t, c) << 32 | (t.split_reads1[c] + t.split_reads2[c] + t.discordant_mates[c])) == highest; }
		unsigned long long holders = __ballot(candidate);
		while (holders) { // in the order of their positions
			const int l = __ffsll((unsigned long long) holders) - 1;
			holders &= holders - 1;
			const uint32_t next = (uint32_t) __shfl((int) c, l);
			if (best == 0xFFFFFFFFu || select_best_replaces(t, next, best)) best = next; // (every lane folds the same values)
		}
	}
	for (uint32_t j = begin + lane; j < end; j += 64) if (order[j] != best) t.filter[order[j]] = FILTER_select_best;
}

// recover_many_spliced: sort keys and the per-pair pass
__global__ void many_spliced_key_kernel(AnnotationView ann, CandidateTable t, const uint32_t* order, int pass, uint64_t* keys) {
	uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
	if (j >= t.n) return;
	const uint32_t c = order ? order[j] : j;
	keys[j] = pass == 0 ? many_spliced_bin_key(t, c) : many_spliced_pair_key(ann, t, c);
}
__global__ void many_spliced_pair_kernel(CandidateTable t, const uint32_t* order, const uint64_t* pair_keys, uint32_t min_spliced_events) {
	uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
	if (j >= t.n || pair_keys[j] == ~0ull || (j > 0 && pair_keys[j - 1] == pair_keys[j])) return; // not the first of a gene pair
	uint32_t end = j + 1;
	while (end < t.n && pair_keys[end] == pair_keys[j]) ++end;
	recover_many_spliced_in_pair(t, order, j, end, min_spliced_events);
}

// recover_isoforms: sort keys (pass 0: iteration rank, pass 1: gene pair + directions of the unfiltered candidates) and the verdicts
__global__ void isoform_key_kernel(CandidateTable t, const uint32_t* order, const uint32_t* iteration_rank, int pass, uint64_t* keys) {
	uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
	if (j >= t.n) return;
	const uint32_t c = order ? order[j] : j;
	keys[j] = pass == 0 ? (uint64_t) iteration_rank[c] : isoform_pair_key(t, c);
}
__global__ void isoform_verdict_kernel(CandidateTable t, const uint64_t* member_keys, const uint32_t* members, uint8_t* recovered) {
	const uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c < t.n) recovered[c] = isoform_may_be_recovered(t, c) && isoform_is_recovered(t, c, member_keys, members, t.n);
}
__global__ void isoform_recover_kernel(CandidateTable t, const uint8_t* recovered) { // a kernel of its own: the verdicts read the filters of all candidates
	const uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c < t.n && recovered[c]) t.filter[c] = FILTER_none;
}

// filter_blacklisted_ranges (mode 0) / recover_known_fusions (mode 1)
__global__ void range_rule_kernel(AnnotationView ann, CoverageView coverage, CandidateTable t, const float* evalues, RangeRuleIndex index, int mode, int32_t max_mate_gap, float evalue_cutoff, GenomicSupport wgs) {
	const uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c >= t.n || !(mode == 0 ? blacklist_considers(t, c, wgs) : known_fusions_considers(t, c))) return;
	if (candidate_matches_any_rule(ann, coverage, t, evalues, c, index, mode, max_mate_gap, evalue_cutoff)) t.filter[c] = mode == 0 ? FILTER_blacklist : FILTER_none;
}

// assign_confidence: sort keys (pass 0: gene pair, pass 1: gene2) and the verdicts
__global__ void confidence_key_kernel(CandidateTable t, int pass, uint64_t* keys) {
	const uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c < t.n) keys[c] = confidence_sort_key(t, c, pass);
}
__global__ void confidence_kernel(AnnotationView ann, CoverageView coverage, CandidateTable t, const float* evalues, ConfidenceTables tables, GenomicSupport wgs, uint8_t* confidence) {
	const uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c < t.n) confidence[c] = candidate_confidence(ann, coverage, t, evalues, tables, c, wgs);
}

// structural variants from WGS: closest genomic breakpoints per candidate; the two filters that use them (mode 0: filter_no_genomic_support, 1: recover_genomic_support)
__global__ void genomic_support_kernel(AnnotationView ann, CandidateTable t, GenomicBreakpoints variants, int32_t max_distance, uint32_t max_itd_length, int32_t* closest1, int32_t* closest2, unsigned int* marked) {
	__shared__ uint32_t block_sum;
	uint32_t mine = 0;
	for (uint32_t c = blockIdx.x * BLOCK + threadIdx.x; c < t.n; c += gridDim.x * BLOCK) {
		closest_genomic_breakpoints(ann, t, variants, c, max_distance, max_itd_length, closest1[c], closest2[c]);
		mine += closest1[c] >= 0;
	}
	block_tally(mine, marked, &block_sum);
}
__global__ void genomic_support_filter_kernel(GenomeView genome, CandidateTable t, GenomicSupport wgs, const uint8_t* confidence, int mode) {
	const uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c >= t.n) return;
	if (mode == 0) { if (t.filter[c] == FILTER_none && lacks_genomic_support(genome, t, wgs, confidence, c)) t.filter[c] = FILTER_no_genomic_support; }
	else if (t.filter[c] != FILTER_none && recovered_by_genomic_support(t, wgs, c)) t.filter[c] = FILTER_none;
}

// filter_in_vitro: expression proxy, gene-pair table, verdicts
// Fragments per gene (reference: the loop over the alignments of filter_in_vitro, source/filter_in_vitro.cpp:91-104).  A few genes hold most of the fragments of a sample, and an
// atomic per fragment on their counters serialises at the L2 (134 ms for 10^8 fragments); every workgroup counts in LDS instead -- a table of (gene, count) slots, a gene owns the
// slot gene % slots once it has claimed it, the rare gene that finds its slot taken by another counts in HBM -- and adds its table to the counters of HBM at the end.
const uint32_t GENE_COUNT_SLOTS = 8192;
__device__ __forceinline__ void count_gene(uint32_t gene, uint32_t* slot_gene, uint32_t* slot_count, uint32_t* gene_read_count) {
	const uint32_t slot = gene & (GENE_COUNT_SLOTS - 1);
	uint32_t owner = slot_gene[slot];
	if (owner == 0xFFFFFFFFu) { owner = atomicCAS(&slot_gene[slot], 0xFFFFFFFFu, gene); if (owner == 0xFFFFFFFFu) owner = gene; }
	if (owner == gene) atomicAdd(&slot_count[slot], 1u); else atomicAdd(&gene_read_count[gene], 1u);
}
__global__ void __launch_bounds__(BLOCK) gene_read_count_kernel(BatchView b, uint32_t* gene_read_count) {
	__shared__ uint32_t slot_gene[GENE_COUNT_SLOTS];
	__shared__ uint32_t slot_count[GENE_COUNT_SLOTS];
	for (uint32_t k = threadIdx.x; k < GENE_COUNT_SLOTS; k += BLOCK) { slot_gene[k] = 0xFFFFFFFFu; slot_count[k] = 0; }
	__syncthreads();
	for (uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x; i < b.n; i += gridDim.x * (uint64_t) BLOCK) {
		AGPU_IDSET(genes);
		load_genes(b, MATE1, i, genes);
		for (uint32_t g = 0; g < genes.n; ++g) count_gene(genes.get(g), slot_gene, slot_count, gene_read_count);
		load_genes(b, in_vitro_second_slot(b, i), i, genes);
		for (uint32_t g = 0; g < genes.n; ++g) count_gene(genes.get(g), slot_gene, slot_count, gene_read_count);
	}
	__syncthreads();
	for (uint32_t k = threadIdx.x; k < GENE_COUNT_SLOTS; k += BLOCK) if (slot_count[k] > 0) atomicAdd(&gene_read_count[slot_gene[k]], slot_count[k]);
}
__global__ void in_vitro_pair_key_kernel(CandidateTable t, uint64_t* keys) { // two keys per candidate: (gene1, gene2) and (gene2, gene1); ~0 = does not count
	const uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c >= t.n) return;
	const bool counts = counts_as_exonic_breakpoint(t, c);
	keys[2 * (uint64_t) c] = counts ? (uint64_t) t.gene1[c] << 32 | t.gene2[c] : ~0ull;
	keys[2 * (uint64_t) c + 1] = counts ? (uint64_t) t.gene2[c] << 32 | t.gene1[c] : ~0ull;
}
__global__ void clip_summary_kernel(BatchView b, ClipSummary* summaries, uint8_t* clip_any) {
	const uint64_t k = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (k >= 3 * b.n) return;
	const ClipSummary summary = clip_summary_of(b, k / 3, (int) (k % 3));
	summaries[CLIP_SUMMARIES_PER_READ * (k / 3) + k % 3] = summary;
	if (summary.clipped) clip_any[k / 3] = 1; // (cleared by the caller; the three threads of a read write the same value)
}
__global__ void in_vitro_kernel(BatchView b, AnnotationView ann, CoverageView coverage, InVitroTables tables, CandidateTable t, uint32_t first, uint32_t end, uint32_t* long_list, uint32_t* n_long, uint32_t long_entries) {
	const uint32_t c = first + blockIdx.x * BLOCK + threadIdx.x;
	if (c >= end) return;
	if (list_entries_of(t, c, 2) > long_entries && in_vitro_looks_at(t, c)) { long_list[atomicAdd(n_long, 1u)] = c; return; } // (the verdict walks the discordant mates: in_vitro_wave_kernel)
	if (is_in_vitro_artifact(b, ann, coverage, tables, t, c)) t.filter[c] = FILTER_in_vitro;
}
// the reads sharded over the ranks: the clipped discordant mates of every candidate the stage looks at, over the reads THIS context holds (event_core.hpp: in_vitro_clipped_mates skips
// the others); a wavefront per 64 candidates would idle on the long lists: one wavefront per candidate, the lanes stride over its list
__global__ void __launch_bounds__(BLOCK) in_vitro_partial_kernel(BatchView b, InVitroTables tables, CandidateTable t, uint32_t first, uint32_t end, uint32_t* clipped) {
	const WaveLanes lanes;
	for (uint64_t c = first + ((blockIdx.x * (uint64_t) BLOCK + threadIdx.x) >> 6); c < end; c += (uint64_t) gridDim.x * (BLOCK / 64)) { // (a bounded grid: a sample has tens of millions of candidates)
		if (!in_vitro_looks_at(t, (uint32_t) c) || list_entries_of(t, (uint32_t) c, 2) == 0) continue;
		uint32_t clipped1 = 0, clipped2 = 0;
		in_vitro_clipped_mates(b, tables, t, (uint32_t) c, clipped1, clipped2, lanes);
		if (lanes.lane == 0) { clipped[2 * c] = clipped1; clipped[2 * c + 1] = clipped2; }
	}
}
__global__ void clipped_entry_flag_kernel(const uint32_t* clipped, uint32_t n, uint8_t* flags) {
	const uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c < n) flags[c] = (clipped[2 * (uint64_t) c] | clipped[2 * (uint64_t) c + 1]) != 0;
}
__global__ void clipped_entry_write_kernel(const uint32_t* clipped, const uint32_t* selected, uint32_t n_selected, uint32_t* entries) {
	const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
	if (k >= n_selected) return;
	const uint32_t c = selected[k];
	entries[3 * (uint64_t) k] = c; entries[3 * (uint64_t) k + 1] = clipped[2 * (uint64_t) c]; entries[3 * (uint64_t) k + 2] = clipped[2 * (uint64_t) c + 1];
}
__global__ void clipped_entry_add_kernel(const uint32_t* entries, uint64_t n_entries, uint32_t n_candidates, uint32_t* clipped, unsigned int* error) {
	const uint64_t k = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (k >= n_entries) return;
	const uint32_t c = entries[3 * k];
	if (c >= n_candidates) { atomicOr(error, 1u); return; }
	atomicAdd(&clipped[2 * (uint64_t) c], entries[3 * k + 1]); atomicAdd(&clipped[2 * (uint64_t) c + 1], entries[3 * k + 2]);
}
__global__ void in_vitro_verdict_kernel(AnnotationView ann, CoverageView coverage, InVitroTables tables, CandidateTable t, const uint32_t* clipped) {
	const uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c >= t.n || !in_vitro_looks_at(t, c)) return;
	if (in_vitro_verdict(ann, coverage, tables, t, c, clipped[2 * (uint64_t) c], clipped[2 * (uint64_t) c + 1])) t.filter[c] = FILTER_in_vitro;
}
__global__ void __launch_bounds__(BLOCK) in_vitro_wave_kernel(BatchView b, AnnotationView ann, CoverageView coverage, InVitroTables tables, CandidateTable t, const uint32_t* long_list, const uint32_t* n_long, uint32_t first_wave) {
	const uint32_t wave = first_wave + ((blockIdx.x * BLOCK + threadIdx.x) >> 6);
	if (wave >= *n_long) return;
	const uint32_t c = long_list[wave];
	const WaveLanes lanes;
	const bool artifact = is_in_vitro_artifact(b, ann, coverage, tables, t, c, lanes);
	if (artifact && lanes.lane == 0) t.filter[c] = FILTER_in_vitro;
}

// recover_both_spliced
__global__ void both_spliced_reads_kernel(BatchView b, AnnotationView ann, CoverageView coverage, const uint32_t* gene_read_count, uint32_t threshold, CandidateTable t, int32_t max_exon_size, uint32_t max_coverage,
                                          uint32_t* reads, uint64_t* keys, uint32_t first, uint32_t end, uint32_t* long_list, uint32_t* n_long, uint32_t long_entries) {
	const uint32_t c = first + blockIdx.x * BLOCK + threadIdx.x;
	if (c >= end) return;
	const bool member = both_spliced_is_member(ann, t, c);
	if (member && list_entries_of(t, c, 0) > long_entries) { long_list[atomicAdd(n_long, 1u)] = c; return; } // (both_spliced_reads_wave_kernel)
	const uint32_t count = member ? both_spliced_supporting_reads(b, ann, coverage, gene_read_count, threshold, t, c, max_exon_size, max_coverage) : 0;
	reads[c] = count;
	keys[c] = count > 0 ? both_spliced_group_key(t, c, false) : ~0ull;
}
__global__ void __launch_bounds__(BLOCK) both_spliced_reads_wave_kernel(BatchView b, AnnotationView ann, CoverageView coverage, const uint32_t* gene_read_count, uint32_t threshold, CandidateTable t, int32_t max_exon_size, uint32_t max_coverage,
                                                                        uint32_t* reads, uint64_t* keys, const uint32_t* long_list, const uint32_t* n_long, uint32_t first_wave) {
	const uint32_t wave = first_wave + ((blockIdx.x * BLOCK + threadIdx.x) >> 6);
	if (wave >= *n_long) return;
	const uint32_t c = long_list[wave];
	const WaveLanes lanes;
	const uint32_t count = both_spliced_supporting_reads(b, ann, coverage, gene_read_count, threshold, t, c, max_exon_size, max_coverage, lanes);
	if (lanes.lane != 0) return;
	reads[c] = count;
	keys[c] = count > 0 ? both_spliced_group_key(t, c, false) : ~0ull;
}
__global__ void both_spliced_gather_kernel(const uint32_t* members, const uint32_t* reads, uint32_t n, uint32_t* member_reads) {
	const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
	if (j < n) member_reads[j] = reads[members[j]];
}
__global__ void both_spliced_eligible_kernel(AnnotationView ann, CandidateTable t, const uint64_t* member_keys, const uint32_t* members, const uint32_t* member_reads, uint8_t* eligible, uint32_t* histogram) {
	const uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c >= t.n) return;
	bool yes = false;
	if (both_spliced_is_recoverable(ann, t, c)) yes = both_spliced_pair_support(ann, t, c, member_keys, members, member_reads, t.n) >= 2; // at least two reads, or the false positive rate sky-rockets
	eligible[c] = yes;
	if (yes) { const uint32_t supporting = t.split_reads1[c] + t.split_reads2[c] + t.discordant_mates[c]; atomicAdd(&histogram[supporting < BOTH_SPLICED_HISTOGRAM_BINS ? supporting : BOTH_SPLICED_HISTOGRAM_BINS - 1], 1u); }
}
__global__ void both_spliced_recover_kernel(CandidateTable t, const uint8_t* eligible, uint32_t min_supporting_reads) {
	const uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c < t.n && eligible[c] && t.split_reads1[c] + t.split_reads2[c] + t.discordant_mates[c] >= min_supporting_reads + both_spliced_proximal_bonus(t, c)) t.filter[c] = FILTER_none;
}

// recover_internal_tandem_duplication
__global__ void count_duplicates_kernel(BatchView b, unsigned int* duplicates) {
	__shared__ uint32_t block_sum;
	uint32_t mine = 0;
	for (uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x; i < b.n; i += gridDim.x * (uint64_t) BLOCK) mine += b.filter[i] == FILTER_duplicates;
	block_tally(mine, duplicates, &block_sum);
}
__global__ void itd_verdict_kernel(BatchView b, AnnotationView ann, CoverageView coverage, CandidateTable t, uint32_t max_itd_length, uint32_t min_supporting_reads, float min_fraction_of_coverage,
                                   uint32_t subsampling_threshold, float duplication_rate, const uint32_t* iteration_rank, uint8_t* verdict, uint32_t* owner, unsigned int* error) {
	const uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c >= t.n) return;
	const int result = itd_verdict(b, ann, coverage, t, c, max_itd_length, min_supporting_reads, min_fraction_of_coverage, subsampling_threshold, duplication_rate);
	if (result == 2) atomicOr(error, 1u);
	verdict[c] = result == 1;
	if (result == 1) // claim the reads this candidate would clear: the first recovered candidate in iteration order counts them
		for (uint64_t k = t.list_offset[3 * (uint64_t) c]; k < t.list_offset[3 * (uint64_t) c + 2]; ++k) {
			const uint32_t read = split_list_entry(t, c, k);
			if (itd_read_is_cleared(b.filter[read])) atomicMin(&owner[read], iteration_rank[c]);
		}
}
__global__ void itd_recover_kernel(BatchView b, CandidateTable t, const uint32_t* iteration_rank, const uint8_t* verdict, uint32_t* owner) {
	const uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c >= t.n || !verdict[c]) return;
	itd_count_cleared_reads(b, t, c, iteration_rank[c], owner);
	t.filter[c] = FILTER_none;
}
__global__ void itd_clear_reads_kernel(BatchView b, const uint32_t* owner) {
	const uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i < b.n && owner[i] != ITD_READ_UNCLAIMED) b.filter[i] = FILTER_none;
}

// chimeric fragments per gene on the device (left in scratch "events.gene_read_count") and the quantile of the non-zero counts
int expression_proxy(agpu_ctx* ctx, float high_expression_quantile, uint32_t& threshold) {
	hipStream_t s = ctx->stream;
	const uint64_t n = ctx->n;
	const size_t n_genes = (size_t) ctx->n_genes + ctx->n_dummy;
	DeviceBuffer& gene_read_count = ctx->scratch("events.gene_read_count");
	// (the counts depend on the gene sets of the fragments alone: filter_in_vitro and recover_both_spliced ask for them one after the other, counted once per annotation)
	if (ctx->read_sharded) { // the reads sharded over the ranks: the counts of the SAMPLE, brought by agpu_set_gene_read_counts (this context's own reads: agpu_gene_read_counts)
		if (!ctx->sample_gene_read_counts_set || ctx->host_gene_read_counts.size() != n_genes) { set_last_error("the reads of the sample are sharded: agpu_set_gene_read_counts must run first"); return AGPU_ERR_INVALID; }
	} else
	if (ctx->gene_read_counts_of_annotation != ctx->annotation_serial || ctx->host_gene_read_counts.size() != n_genes) {
		ALLOC(gene_read_count, std::max<size_t>(n_genes, 1) * 4);
		HIP_CHECK(hipMemsetAsync(gene_read_count.ptr, 0, std::max<size_t>(n_genes, 1) * 4, s));
		if (n > 0) { KernelTimer timer(ctx, "gene_read_count_kernel", n * 22); gene_read_count_kernel<<<(unsigned int) std::min<uint64_t>((n + BLOCK - 1) / BLOCK, 2048), BLOCK, 0, s>>>(ctx->batch, gene_read_count.as<uint32_t>()); }
		ctx->host_gene_read_counts.assign(n_genes, 0);
		if (n_genes > 0) HIP_CHECK(hipMemcpyAsync(ctx->host_gene_read_counts.data(), gene_read_count.ptr, n_genes * 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		ctx->gene_read_counts_of_annotation = ctx->annotation_serial;
	}
	threshold = high_expression_threshold(ctx->host_gene_read_counts, high_expression_quantile);
	return AGPU_OK;
}

// the batch a stage sees that judges candidates by their read lists, with its walk bytes as of now (agpu_sharded.hip: candidate_walk_batch -- the own batch, or, the reads sharded
// over the ranks, the replicated states of the fragments of the sample)
int batch_with_walk_bytes(agpu_ctx* ctx, BatchView& batch) { return candidate_walk_batch(ctx, batch, true); }

// the list of the candidates a thread kernel notes for the wavefronts, and their number read back (the launch of the second kernel is sized by it)
struct LongLists {
	uint32_t* list = nullptr; uint32_t* count = nullptr;
	int prepare(agpu_ctx* ctx, uint32_t candidates) {
		DeviceBuffer& buffer = ctx->scratch("events.long_list"); DeviceBuffer& counter = ctx->scratch("events.long_count");
		ALLOC(buffer, (size_t) std::max<uint32_t>(candidates, 1) * 4); ALLOC(counter, 4);
		HIP_CHECK(hipMemsetAsync(counter.ptr, 0, 4, ctx->stream));
		list = buffer.as<uint32_t>(); count = counter.as<uint32_t>();
		return AGPU_OK;
	}
	int noted(agpu_ctx* ctx, uint32_t& n) { HIP_CHECK(hipMemcpyAsync(&n, count, 4, hipMemcpyDeviceToHost, ctx->stream)); HIP_CHECK(hipStreamSynchronize(ctx->stream)); return AGPU_OK; }
};

int run_event_stage(agpu_ctx* ctx, int stage, uint8_t filter_id, const char* kernel_name, uint32_t min_anchor_length, uint64_t* remaining) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	if (stage == EVENT_both_intronic && ctx->candidates_imported) { set_last_error("filter_both_intronic reads the read lists: not available on an imported (replicated) candidate table"); return AGPU_ERR_INVALID; }
	if ((stage == EVENT_no_coverage || stage == EVENT_marginal_read_through) && !ctx->have_coverage) { set_last_error("agpu_upload_coverage must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	DeviceBuffer& counter = ctx->scratch("events.counter");
	ALLOC(counter, 16);
	HIP_CHECK(hipMemsetAsync(counter.ptr, 0, 16, s));
	(void) hipEventRecord(ctx->event_start, s);
	if (C > 0) {
		const int effective_stage = ctx->params.filter_enabled[filter_id] ? stage : EVENT_count_only; // a stage switched off with -f only counts
		if (effective_stage == EVENT_both_intronic) { // (walks the read lists)
			BatchView batch;
			{ const int status = batch_with_walk_bytes(ctx, batch); if (status != AGPU_OK) return status; }
			const int status = for_each_list_window(ctx, [&](const CandidateTable& window, uint32_t begin, uint32_t end) -> int {
				LongLists lists; uint32_t n_long = 0;
				{ const int status = lists.prepare(ctx, end - begin); if (status != AGPU_OK) return status; }
				{ KernelTimer timer(ctx, kernel_name, (uint64_t) C * 60 + (uint64_t) ctx->n_list_entries * 8);
				  event_predicate_kernel<<<tally_grid(end - begin, BLOCK) * 4, BLOCK, 0, s>>>(effective_stage, batch, ctx->annotation, ctx->genome, ctx->coverage, window, min_anchor_length, counter.as<unsigned int>(), begin, end, lists.list, lists.count, long_list_entries(LONG_LIST)); }
				{ const int status = lists.noted(ctx, n_long); if (status != AGPU_OK) return status; }
				if (n_long > 0) { KernelTimer timer(ctx, "event_predicate_wave_kernel(both_intronic)", (uint64_t) n_long * 60);
				  for_each_wave_chunk(n_long, [&](uint64_t first, uint64_t count) { event_predicate_wave_kernel<<<(unsigned int) ((count * 64 + BLOCK - 1) / BLOCK), BLOCK, 0, s>>>(effective_stage, batch, ctx->annotation, ctx->genome, ctx->coverage, window, min_anchor_length, counter.as<unsigned int>(), lists.list, lists.count, (uint32_t) first); }); }
				return AGPU_OK;
			}, LISTS_OF_UNFILTERED);
			if (status != AGPU_OK) return status;
		} else {
			KernelTimer timer(ctx, kernel_name, (uint64_t) C * 60);
			event_predicate_kernel<<<tally_grid(C, BLOCK) * 4, BLOCK, 0, s>>>(effective_stage, ctx->batch, ctx->annotation, ctx->genome, ctx->coverage, ctx->candidates, min_anchor_length, counter.as<unsigned int>());
		}
	}
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	unsigned int kept = 0;
	HIP_CHECK(hipMemcpy(&kept, counter.ptr, 4, hipMemcpyDeviceToHost));
	// algorithmic bytes: a candidate that an earlier filter has dropped costs its filter byte, one that is still alive its row (60 bytes); the number alive is known only now
	// (the ones this stage drops are not counted: the figure is a lower bound).  Round 2 priced every candidate with 60 bytes: 19 TB/s "achieved" for the late stages.
	ctx->last_bytes = (uint64_t) C + (uint64_t) kept * 59 + (stage == EVENT_both_intronic ? (uint64_t) ctx->n_list_entries * 8 : 0);
	{ std::lock_guard<std::mutex> lock(ctx->profile_mutex); if (!ctx->samples_done.empty() && ctx->samples_done.back().name == kernel_name) ctx->samples_done.back().bytes = ctx->last_bytes; }
	if (remaining) *remaining = kept;
	return AGPU_OK;
}

}

extern "C" int agpu_upload_coverage(agpu_ctx* ctx, const agpu_coverage_view* in) {
	if (!ctx || !in) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint64_t windows = in->n_contigs ? in->window_offset[in->n_contigs] : 0;
	ALLOC(ctx->coverage_window_offset, ((size_t) in->n_contigs + 1) * 8); ALLOC(ctx->coverage_windows, std::max<uint64_t>(windows, 1) * 2);
	ALLOC(ctx->coverage_fragment_starts, std::max<uint64_t>(windows, 1)); ALLOC(ctx->coverage_fragment_ends, std::max<uint64_t>(windows, 1));
	HIP_CHECK(hipMemcpyAsync(ctx->coverage_window_offset.ptr, in->window_offset, ((size_t) in->n_contigs + 1) * 8, hipMemcpyHostToDevice, s));
	if (windows > 0) {
		HIP_CHECK(hipMemcpyAsync(ctx->coverage_windows.ptr, in->coverage, windows * 2, hipMemcpyHostToDevice, s));
		HIP_CHECK(hipMemcpyAsync(ctx->coverage_fragment_starts.ptr, in->fragment_starts, windows, hipMemcpyHostToDevice, s));
		HIP_CHECK(hipMemcpyAsync(ctx->coverage_fragment_ends.ptr, in->fragment_ends, windows, hipMemcpyHostToDevice, s));
	}
	HIP_CHECK(hipStreamSynchronize(s));
	ctx->coverage.n_contigs = in->n_contigs; ctx->coverage.window_offset = ctx->coverage_window_offset.as<uint64_t>(); ctx->coverage.coverage = ctx->coverage_windows.as<uint16_t>();
	ctx->coverage.fragment_starts = ctx->coverage_fragment_starts.as<uint8_t>(); ctx->coverage.fragment_ends = ctx->coverage_fragment_ends.as<uint8_t>();
	ctx->have_coverage = true;
	return AGPU_OK;
}

extern "C" int agpu_filter_both_intronic(agpu_ctx* ctx, uint64_t* remaining) { return run_event_stage(ctx, EVENT_both_intronic, FILTER_intronic, "event_predicate_kernel(both_intronic)", 0, remaining); }
extern "C" int agpu_filter_short_anchor(agpu_ctx* ctx, uint32_t min_length, uint64_t* remaining) { return run_event_stage(ctx, EVENT_short_anchor, FILTER_short_anchor, "event_predicate_kernel(short_anchor)", min_length, remaining); }
extern "C" int agpu_filter_end_to_end(agpu_ctx* ctx, uint64_t* remaining) { return run_event_stage(ctx, EVENT_end_to_end, FILTER_end_to_end, "event_predicate_kernel(end_to_end)", 0, remaining); }
extern "C" int agpu_filter_no_coverage(agpu_ctx* ctx, uint64_t* remaining) { return run_event_stage(ctx, EVENT_no_coverage, FILTER_no_coverage, "event_predicate_kernel(no_coverage)", 0, remaining); }
extern "C" int agpu_filter_marginal_read_through(agpu_ctx* ctx, uint64_t* remaining) { return run_event_stage(ctx, EVENT_marginal_read_through, FILTER_marginal_read_through, "event_predicate_kernel(marginal_read_through)", 0, remaining); }

extern "C" int agpu_select_most_supported_breakpoints(agpu_ctx* ctx, uint64_t* remaining) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	if (!ctx->iteration_order_done) { const int status = agpu_candidate_iteration_order(ctx, nullptr); if (status != AGPU_OK) return status; } // hazard H2
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	DeviceBuffer& counter = ctx->scratch("events.counter"); DeviceBuffer& keys_in = ctx->scratch("events.keys_in"); DeviceBuffer& keys_out = ctx->scratch("events.keys_out");
	DeviceBuffer& order_a = ctx->scratch("events.order_a"); DeviceBuffer& order_b = ctx->scratch("events.order_b"); DeviceBuffer& scratch = ctx->scratch("events.rocprim");
	const size_t C1 = std::max<uint32_t>(C, 1);
	ALLOC(counter, 16); ALLOC(keys_in, C1 * 8); ALLOC(keys_out, C1 * 8); ALLOC(order_a, C1 * 4); ALLOC(order_b, C1 * 4);
	HIP_CHECK(hipMemsetAsync(counter.ptr, 0, 16, s));
	(void) hipEventRecord(ctx->event_start, s);
	if (C > 0 && ctx->params.filter_enabled[FILTER_select_best]) {
		const CandidateTable& t = ctx->candidates;
		const unsigned int grid = (unsigned int) ((C + BLOCK - 1) / BLOCK);
		size_t bytes = 0;
		// stable sorts: by iteration rank, then by group
		select_best_key_kernel<<<grid, BLOCK, 0, s>>>(t, nullptr, ctx->cand_iteration_rank.as<uint32_t>(), 0, keys_in.as<uint64_t>());
		HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), rocprim::counting_iterator<uint32_t>(0), order_a.as<uint32_t>(), C, 0, 32, s));
		if (bytes > scratch.capacity) ALLOC(scratch, bytes);
		HIP_CHECK(rocprim::radix_sort_pairs(scratch.ptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), rocprim::counting_iterator<uint32_t>(0), order_a.as<uint32_t>(), C, 0, 32, s));
		select_best_key_kernel<<<grid, BLOCK, 0, s>>>(t, order_a.as<uint32_t>(), ctx->cand_iteration_rank.as<uint32_t>(), 1, keys_in.as<uint64_t>());
		HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), order_a.as<uint32_t>(), order_b.as<uint32_t>(), C, 0, 64, s));
		if (bytes > scratch.capacity) ALLOC(scratch, bytes);
		HIP_CHECK(rocprim::radix_sort_pairs(scratch.ptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), order_a.as<uint32_t>(), order_b.as<uint32_t>(), C, 0, 64, s));
		DeviceBuffer& large_groups = ctx->scratch("events.large_groups");
		const char* knob = getenv("ARRIBA_SELECT_BEST_SMALL"); // (tests: 1 sends every group of two and more through the wavefront's fold)
		const uint32_t small = knob != nullptr && atoi(knob) >= 1 ? (uint32_t) atoi(knob) : SELECT_BEST_SMALL;
		ALLOC(large_groups, (C1 / small + 1) * sizeof(GroupRange));
		{ KernelTimer timer(ctx, "select_best_group_kernel", (uint64_t) C * 40);
		  select_best_group_kernel<<<tally_grid(C, BLOCK) * 4, BLOCK, 0, s>>>(t, order_b.as<uint32_t>(), keys_out.as<uint64_t>(), counter.as<unsigned int>(), large_groups.as<GroupRange>(), counter.as<uint32_t>() + 1, small); }
		uint32_t n_large = 0;
		HIP_CHECK(hipMemcpyAsync(&n_large, counter.as<uint32_t>() + 1, 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		if (n_large > 0) { KernelTimer timer(ctx, "select_best_wave_kernel", (uint64_t) n_large * small * 40);
		  for_each_wave_chunk(n_large, [&](uint64_t first, uint64_t count) { select_best_wave_kernel<<<(unsigned int) ((count * 64 + BLOCK - 1) / BLOCK), BLOCK, 0, s>>>(t, order_b.as<uint32_t>(), keys_out.as<uint64_t>(), large_groups.as<GroupRange>(), counter.as<uint32_t>() + 1, (uint32_t) first); }); }
	} else if (C > 0) {
		event_predicate_kernel<<<tally_grid(C, BLOCK) * 4, BLOCK, 0, s>>>(EVENT_count_only, ctx->batch, ctx->annotation, ctx->genome, ctx->coverage, ctx->candidates, 0u, counter.as<unsigned int>());
	}
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (uint64_t) C * 100;
	unsigned int kept = 0;
	HIP_CHECK(hipMemcpy(&kept, counter.ptr, 4, hipMemcpyDeviceToHost));
	if (remaining) *remaining = kept;
	return AGPU_OK;
}

extern "C" int agpu_recover_many_spliced(agpu_ctx* ctx, uint32_t min_spliced_events, uint64_t* remaining) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	DeviceBuffer& counter = ctx->scratch("events.counter"); DeviceBuffer& keys_in = ctx->scratch("events.keys_in"); DeviceBuffer& keys_out = ctx->scratch("events.keys_out");
	DeviceBuffer& order_a = ctx->scratch("events.order_a"); DeviceBuffer& order_b = ctx->scratch("events.order_b"); DeviceBuffer& scratch = ctx->scratch("events.rocprim");
	const size_t C1 = std::max<uint32_t>(C, 1);
	ALLOC(counter, 16); ALLOC(keys_in, C1 * 8); ALLOC(keys_out, C1 * 8); ALLOC(order_a, C1 * 4); ALLOC(order_b, C1 * 4);
	HIP_CHECK(hipMemsetAsync(counter.ptr, 0, 16, s));
	(void) hipEventRecord(ctx->event_start, s);
	if (C > 0) {
		const CandidateTable& t = ctx->candidates;
		const unsigned int grid = (unsigned int) ((C + BLOCK - 1) / BLOCK);
		if (ctx->params.filter_enabled[28 /* many_spliced */]) {
			size_t bytes = 0;
			many_spliced_key_kernel<<<grid, BLOCK, 0, s>>>(ctx->annotation, t, nullptr, 0, keys_in.as<uint64_t>());
			HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), rocprim::counting_iterator<uint32_t>(0), order_a.as<uint32_t>(), C, 0, 64, s));
			if (bytes > scratch.capacity) ALLOC(scratch, bytes);
			HIP_CHECK(rocprim::radix_sort_pairs(scratch.ptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), rocprim::counting_iterator<uint32_t>(0), order_a.as<uint32_t>(), C, 0, 64, s));
			many_spliced_key_kernel<<<grid, BLOCK, 0, s>>>(ctx->annotation, t, order_a.as<uint32_t>(), 1, keys_in.as<uint64_t>());
			HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), order_a.as<uint32_t>(), order_b.as<uint32_t>(), C, 0, 64, s));
			if (bytes > scratch.capacity) ALLOC(scratch, bytes);
			HIP_CHECK(rocprim::radix_sort_pairs(scratch.ptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), order_a.as<uint32_t>(), order_b.as<uint32_t>(), C, 0, 64, s));
			KernelTimer timer(ctx, "many_spliced_pair_kernel", (uint64_t) C * 40);
			many_spliced_pair_kernel<<<grid, BLOCK, 0, s>>>(t, order_b.as<uint32_t>(), keys_out.as<uint64_t>(), min_spliced_events);
		}
		event_predicate_kernel<<<tally_grid(C, BLOCK) * 4, BLOCK, 0, s>>>(EVENT_count_only, ctx->batch, ctx->annotation, ctx->genome, ctx->coverage, ctx->candidates, 0u, counter.as<unsigned int>());
	}
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (uint64_t) C * 100;
	unsigned int kept = 0;
	HIP_CHECK(hipMemcpy(&kept, counter.ptr, 4, hipMemcpyDeviceToHost));
	if (remaining) *remaining = kept;
	return AGPU_OK;
}

namespace {
int run_range_rules(agpu_ctx* ctx, const agpu_range_rule* rules, uint32_t n_rules, int mode, uint8_t filter_id, int32_t max_mate_gap, float evalue_cutoff, uint64_t* remaining) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	if (n_rules > 0 && !rules) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	if (!ctx->evalue_done) { set_last_error("agpu_estimate_expected_fusions must run first"); return AGPU_ERR_INVALID; }
	if (mode == 1 && !ctx->have_coverage) { set_last_error("agpu_upload_coverage must run first"); return AGPU_ERR_INVALID; }
	for (uint32_t r = 0; r < n_rules; ++r) {
		const agpu_range_item* items[2] = { &rules[r].first, &rules[r].second };
		for (int k = 0; k < 2; ++k) {
			if (items[k]->type > AGPU_RULE_NOT_BOTH_SPLICED) { set_last_error("range rule with an unknown item type"); return AGPU_ERR_INVALID; }
			if (items[k]->type == AGPU_RULE_GENE && items[k]->gene >= ctx->n_genes + ctx->n_dummy) { set_last_error("range rule names a gene outside the annotation"); return AGPU_ERR_INVALID; }
		}
	}
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	DeviceBuffer& counter = ctx->scratch("events.counter"); DeviceBuffer& device_rules = ctx->scratch("rules.rules"); DeviceBuffer& bin_keys = ctx->scratch("rules.bin_keys");
	DeviceBuffer& bin_offset = ctx->scratch("rules.bin_offset"); DeviceBuffer& bin_rules = ctx->scratch("rules.bin_rules");
	ALLOC(counter, 16);
	HIP_CHECK(hipMemsetAsync(counter.ptr, 0, 16, s));
	(void) hipEventRecord(ctx->event_start, s);
	if (C > 0 && n_rules > 0 && ctx->params.filter_enabled[filter_id]) {
		RangeRuleBins bins;
		build_range_rule_bins(rules, n_rules, mode == 0 ? max_mate_gap : 0, bins);
		if (!bins.bin_keys.empty()) { // a file of keywords only has no bins: nothing can match
			ALLOC(device_rules, (size_t) n_rules * sizeof(agpu_range_rule)); ALLOC(bin_keys, bins.bin_keys.size() * 8); ALLOC(bin_offset, bins.bin_offset.size() * 4); ALLOC(bin_rules, bins.bin_rules.size() * 4);
			HIP_CHECK(hipMemcpyAsync(device_rules.ptr, rules, (size_t) n_rules * sizeof(agpu_range_rule), hipMemcpyHostToDevice, s));
			HIP_CHECK(hipMemcpyAsync(bin_keys.ptr, bins.bin_keys.data(), bins.bin_keys.size() * 8, hipMemcpyHostToDevice, s));
			HIP_CHECK(hipMemcpyAsync(bin_offset.ptr, bins.bin_offset.data(), bins.bin_offset.size() * 4, hipMemcpyHostToDevice, s));
			HIP_CHECK(hipMemcpyAsync(bin_rules.ptr, bins.bin_rules.data(), bins.bin_rules.size() * 4, hipMemcpyHostToDevice, s));
			RangeRuleIndex index;
			index.rules = device_rules.as<agpu_range_rule>(); index.n_rules = n_rules; index.bin_keys = bin_keys.as<uint64_t>(); index.bin_offset = bin_offset.as<uint32_t>(); index.bin_rules = bin_rules.as<uint32_t>();
			index.n_bins = (uint32_t) bins.bin_keys.size();
			KernelTimer timer(ctx, mode == 0 ? "range_rule_kernel(blacklist)" : "range_rule_kernel(known_fusions)", (uint64_t) C * 40);
			range_rule_kernel<<<(unsigned int) ((C + BLOCK - 1) / BLOCK), BLOCK, 0, s>>>(ctx->annotation, ctx->coverage, ctx->candidates, ctx->cand_evalue.as<float>(), index, mode, max_mate_gap, evalue_cutoff, ctx->genomic_support());
			HIP_CHECK(hipStreamSynchronize(s)); // the host vectors of the index are read by the copies above
		}
	}
	if (C > 0) event_predicate_kernel<<<tally_grid(C, BLOCK) * 4, BLOCK, 0, s>>>(EVENT_count_only, ctx->batch, ctx->annotation, ctx->genome, ctx->coverage, ctx->candidates, 0u, counter.as<unsigned int>());
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (uint64_t) C * 40;
	unsigned int kept = 0;
	HIP_CHECK(hipMemcpy(&kept, counter.ptr, 4, hipMemcpyDeviceToHost));
	if (remaining) *remaining = kept;
	return AGPU_OK;
}
}
extern "C" int agpu_filter_blacklisted_ranges(agpu_ctx* ctx, const agpu_range_rule* rules, uint32_t n_rules, float evalue_cutoff, int32_t max_mate_gap, uint64_t* remaining) {
	return run_range_rules(ctx, rules, n_rules, 0, FILTER_blacklist, max_mate_gap, evalue_cutoff, remaining);
}
extern "C" int agpu_recover_known_fusions(agpu_ctx* ctx, const agpu_range_rule* rules, uint32_t n_rules, int32_t max_mate_gap, uint64_t* remaining) {
	return run_range_rules(ctx, rules, n_rules, 1, FILTER_known_fusions, max_mate_gap, 0, remaining);
}

extern "C" int agpu_mark_genomic_support(agpu_ctx* ctx, const agpu_genomic_breakpoint* variants, uint32_t n_variants, int32_t max_distance, uint64_t* marked) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	if (n_variants > 0 && !variants) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	// one array sorted by (contigs + directions, first position, line of the file): the reference's nested index flattened
	std::vector<uint32_t> order(n_variants);
	for (uint32_t k = 0; k < n_variants; ++k) order[k] = k;
	std::vector<uint64_t> keys(n_variants);
	for (uint32_t k = 0; k < n_variants; ++k) keys[k] = genomic_breakpoint_key(variants[k].contig1, variants[k].contig2, variants[k].upstream1 != 0, variants[k].upstream2 != 0);
	std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return keys[x] != keys[y] ? keys[x] < keys[y] : variants[x].position1 < variants[y].position1; });
	std::vector<uint64_t> sorted_keys(n_variants); std::vector<int32_t> position1(n_variants), position2(n_variants);
	for (uint32_t k = 0; k < n_variants; ++k) { sorted_keys[k] = keys[order[k]]; position1[k] = variants[order[k]].position1; position2[k] = variants[order[k]].position2; }
	DeviceBuffer& counter = ctx->scratch("events.counter"); DeviceBuffer& device_keys = ctx->scratch("wgs.keys"); DeviceBuffer& device_position1 = ctx->scratch("wgs.position1"); DeviceBuffer& device_position2 = ctx->scratch("wgs.position2");
	const size_t C1 = std::max<uint32_t>(C, 1), V1 = std::max<uint32_t>(n_variants, 1);
	ALLOC(counter, 16); ALLOC(device_keys, V1 * 8); ALLOC(device_position1, V1 * 4); ALLOC(device_position2, V1 * 4); ALLOC(ctx->cand_closest1, C1 * 4); ALLOC(ctx->cand_closest2, C1 * 4);
	HIP_CHECK(hipMemsetAsync(counter.ptr, 0, 16, s));
	HIP_CHECK(hipMemsetAsync(ctx->cand_closest1.ptr, 0xFF, C1 * 4, s));
	HIP_CHECK(hipMemsetAsync(ctx->cand_closest2.ptr, 0xFF, C1 * 4, s));
	(void) hipEventRecord(ctx->event_start, s);
	if (C > 0 && n_variants > 0) {
		HIP_CHECK(hipMemcpyAsync(device_keys.ptr, sorted_keys.data(), (size_t) n_variants * 8, hipMemcpyHostToDevice, s));
		HIP_CHECK(hipMemcpyAsync(device_position1.ptr, position1.data(), (size_t) n_variants * 4, hipMemcpyHostToDevice, s));
		HIP_CHECK(hipMemcpyAsync(device_position2.ptr, position2.data(), (size_t) n_variants * 4, hipMemcpyHostToDevice, s));
		GenomicBreakpoints index = { device_keys.as<uint64_t>(), device_position1.as<int32_t>(), device_position2.as<int32_t>(), n_variants };
		KernelTimer timer(ctx, "genomic_support_kernel", (uint64_t) C * 40);
		genomic_support_kernel<<<tally_grid(C, BLOCK), BLOCK, 0, s>>>(ctx->annotation, ctx->candidates, index, max_distance, ctx->params.max_itd_length, ctx->cand_closest1.as<int32_t>(), ctx->cand_closest2.as<int32_t>(), counter.as<unsigned int>());
	}
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop)); // the host arrays of the index are read by the copies above
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (uint64_t) C * 40;
	ctx->genomic_support_marked = true;
	unsigned int count = 0;
	HIP_CHECK(hipMemcpy(&count, counter.ptr, 4, hipMemcpyDeviceToHost));
	if (marked) *marked = count;
	return AGPU_OK;
}
extern "C" int agpu_get_genomic_support(agpu_ctx* ctx, int32_t* closest1, int32_t* closest2) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	const uint32_t C = ctx->n_candidates;
	if (!ctx->genomic_support_marked) { for (uint32_t c = 0; c < C; ++c) { if (closest1) closest1[c] = -1; if (closest2) closest2[c] = -1; } return AGPU_OK; }
	if (closest1 && C > 0) HIP_CHECK(hipMemcpy(closest1, ctx->cand_closest1.ptr, (size_t) C * 4, hipMemcpyDeviceToHost));
	if (closest2 && C > 0) HIP_CHECK(hipMemcpy(closest2, ctx->cand_closest2.ptr, (size_t) C * 4, hipMemcpyDeviceToHost));
	return AGPU_OK;
}

// ---- the candidates of one output file, picked on the device (agpu_select_candidates / agpu_get_selected_candidates; agpu_get_filters_of) ------------------------------
namespace {
__global__ void candidate_written_kernel(const uint8_t* filter, uint32_t n, bool discarded, uint8_t* flags) {
	const uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c < n) flags[c] = ((filter[c] == FILTER_none) != discarded) ? 1 : 0; // source/output_fusions.cpp:1083-1089
}
template <class T> __global__ void gather_column_kernel(const T* column, const uint32_t* ids, uint64_t n, T fill, T* out) {
	const uint64_t k = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (k < n) out[k] = column != nullptr ? column[ids[k]] : fill;
}
template <class T> int gather_column(agpu_ctx* ctx, const char* scratch_name, const T* column, T fill, const uint32_t* ids, uint64_t n, T* host) {
	if (host == nullptr || n == 0) return AGPU_OK;
	DeviceBuffer& staged = ctx->scratch(scratch_name);
	ALLOC(staged, n * sizeof(T));
	gather_column_kernel<T><<<(unsigned int) ((n + BLOCK - 1) / BLOCK), BLOCK, 0, ctx->stream>>>(column, ids, n, fill, staged.as<T>());
	HIP_CHECK(hipMemcpyAsync(host, staged.ptr, n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
	return AGPU_OK;
}
}
extern "C" int agpu_select_candidates(agpu_ctx* ctx, int discarded, uint64_t* n) {
	if (!ctx || !ctx->fusions_done || !n) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	if (!ctx->failed_launch.empty()) { set_last_error("a kernel of this sample was not launched (" + ctx->failed_launch + "): its results are not to be written"); return AGPU_ERR_DEVICE; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	DeviceBuffer& flags = ctx->scratch("select.flags"); DeviceBuffer& ids = ctx->scratch("select.ids"); DeviceBuffer& count = ctx->scratch("select.count"); DeviceBuffer& scratch = ctx->scratch("select.rocprim");
	ALLOC(flags, std::max<uint32_t>(C, 1)); ALLOC(ids, (size_t) std::max<uint32_t>(C, 1) * 4); ALLOC(count, 4);
	HIP_CHECK(hipMemsetAsync(count.ptr, 0, 4, s));
	uint32_t selected = 0;
	if (C > 0) {
		candidate_written_kernel<<<(C + BLOCK - 1) / BLOCK, BLOCK, 0, s>>>(ctx->cand_filter.as<uint8_t>(), C, discarded != 0, flags.as<uint8_t>());
		size_t bytes = 0;
		HIP_CHECK(rocprim::select(nullptr, bytes, rocprim::counting_iterator<uint32_t>(0), flags.as<uint8_t>(), ids.as<uint32_t>(), count.as<uint32_t>(), C, s));
		if (bytes > scratch.capacity) ALLOC(scratch, bytes);
		HIP_CHECK(rocprim::select(scratch.ptr, bytes, rocprim::counting_iterator<uint32_t>(0), flags.as<uint8_t>(), ids.as<uint32_t>(), count.as<uint32_t>(), C, s));
		HIP_CHECK(hipMemcpyAsync(&selected, count.ptr, 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
	}
	ctx->n_selected = selected; ctx->selected_of_candidates = C;
	*n = selected;
	return AGPU_OK;
}
extern "C" int agpu_get_selected_candidates(agpu_ctx* ctx, const agpu_selected_candidates* out) {
	if (!ctx || !ctx->fusions_done || !out || ctx->selected_of_candidates != ctx->n_candidates) { set_last_error("agpu_select_candidates must run first"); return AGPU_ERR_INVALID; }
	if (out->confidence && ctx->confidence_candidates != ctx->n_candidates) { set_last_error("agpu_assign_confidence must run first"); return AGPU_ERR_INVALID; }
	if (out->iteration_rank && !ctx->iteration_order_done) { set_last_error("agpu_candidate_iteration_order must run first"); return AGPU_ERR_INVALID; }
	if (out->evalue && !ctx->evalue_done) { set_last_error("agpu_estimate_expected_fusions must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	const uint64_t n = ctx->n_selected;
	if (n == 0) return AGPU_OK;
	const uint32_t* ids = ctx->scratch("select.ids").as<uint32_t>();
	if (out->candidate) HIP_CHECK(hipMemcpyAsync(out->candidate, ids, n * 4, hipMemcpyDeviceToHost, ctx->stream));
	#define GATHER(field, buffer, type, fill) do { int status_ = gather_column<type>(ctx, "select." #field, (const type*) (buffer), (type) (fill), ids, n, out->field); if (status_ != AGPU_OK) return status_; } while (0)
	GATHER(gene1, ctx->cand_gene1.ptr, uint32_t, 0); GATHER(gene2, ctx->cand_gene2.ptr, uint32_t, 0); GATHER(contigs, ctx->cand_contigs.ptr, uint32_t, 0);
	GATHER(breakpoint1, ctx->cand_breakpoint1.ptr, int32_t, 0); GATHER(breakpoint2, ctx->cand_breakpoint2.ptr, int32_t, 0); GATHER(flags, ctx->cand_flags.ptr, uint32_t, 0); GATHER(filter, ctx->cand_filter.ptr, uint8_t, 0);
	GATHER(split_reads1, ctx->cand_split_reads1.ptr, uint32_t, 0); GATHER(split_reads2, ctx->cand_split_reads2.ptr, uint32_t, 0); GATHER(discordant_mates, ctx->cand_discordant_mates.ptr, uint32_t, 0);
	GATHER(evalue, ctx->cand_evalue.ptr, float, 0); GATHER(confidence, ctx->scratch("events.confidence").ptr, uint8_t, 0); GATHER(iteration_rank, ctx->cand_iteration_rank.ptr, uint32_t, 0);
	GATHER(closest_genomic_breakpoint1, ctx->genomic_support_marked ? ctx->cand_closest1.ptr : nullptr, int32_t, -1); GATHER(closest_genomic_breakpoint2, ctx->genomic_support_marked ? ctx->cand_closest2.ptr : nullptr, int32_t, -1);
	#undef GATHER
	HIP_CHECK(hipStreamSynchronize(ctx->stream));
	return AGPU_OK;
}
extern "C" int agpu_get_filters_of(agpu_ctx* ctx, const uint32_t* fragments, uint64_t n, uint8_t* filter) {
	if (!ctx || !ctx->have_batch || (n > 0 && (!fragments || !filter))) { set_last_error("no batch uploaded"); return AGPU_ERR_INVALID; }
	// (the reads sharded over the ranks: the fragments are global name ranks, their filters the replicated ones)
	if (ctx->read_sharded && !ctx->state_imported) { set_last_error("the reads of the sample are sharded: agpu_read_state_import must run first"); return AGPU_ERR_INVALID; }
	const uint64_t known = ctx->read_sharded ? ctx->global_n : ctx->n;
	for (uint64_t k = 0; k < n; ++k) if (fragments[k] >= known) { set_last_error("fragment index out of range"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	if (n == 0) return AGPU_OK;
	DeviceBuffer& ids = ctx->scratch("filters_of.ids");
	ALLOC(ids, n * 4);
	HIP_CHECK(hipMemcpyAsync(ids.ptr, fragments, n * 4, hipMemcpyHostToDevice, ctx->stream));
	const int status = gather_column<uint8_t>(ctx, "filters_of.out", ctx->read_sharded ? ctx->scratch("sharded.filter").as<uint8_t>() : ctx->filter.as<uint8_t>(), 0, ids.as<uint32_t>(), n, filter);
	if (status != AGPU_OK) return status;
	HIP_CHECK(hipStreamSynchronize(ctx->stream));
	return AGPU_OK;
}

namespace {
int run_genomic_support_filter(agpu_ctx* ctx, int mode, uint8_t filter_id, uint64_t* remaining) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	if (!ctx->genomic_support_marked) { set_last_error("agpu_mark_genomic_support must run first"); return AGPU_ERR_INVALID; }
	if (mode == 0 && ctx->confidence_candidates != ctx->n_candidates) { set_last_error("agpu_assign_confidence must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	DeviceBuffer& counter = ctx->scratch("events.counter");
	ALLOC(counter, 16);
	HIP_CHECK(hipMemsetAsync(counter.ptr, 0, 16, s));
	(void) hipEventRecord(ctx->event_start, s);
	if (C > 0) {
		if (ctx->params.filter_enabled[filter_id])
			genomic_support_filter_kernel<<<(unsigned int) ((C + BLOCK - 1) / BLOCK), BLOCK, 0, s>>>(ctx->genome, ctx->candidates, ctx->genomic_support(), ctx->scratch("events.confidence").as<uint8_t>(), mode);
		event_predicate_kernel<<<tally_grid(C, BLOCK) * 4, BLOCK, 0, s>>>(EVENT_count_only, ctx->batch, ctx->annotation, ctx->genome, ctx->coverage, ctx->candidates, 0u, counter.as<unsigned int>());
	}
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (uint64_t) C * 10;
	unsigned int kept = 0;
	HIP_CHECK(hipMemcpy(&kept, counter.ptr, 4, hipMemcpyDeviceToHost));
	if (remaining) *remaining = kept;
	return AGPU_OK;
}
}
extern "C" int agpu_filter_no_genomic_support(agpu_ctx* ctx, uint64_t* remaining) { return run_genomic_support_filter(ctx, 0, FILTER_no_genomic_support, remaining); }
extern "C" int agpu_recover_genomic_support(agpu_ctx* ctx, uint64_t* remaining) { return run_genomic_support_filter(ctx, 1, FILTER_genomic_support, remaining); }

extern "C" int agpu_assign_confidence(agpu_ctx* ctx, uint8_t* confidence) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	if (!ctx->evalue_done) { set_last_error("agpu_estimate_expected_fusions must run first"); return AGPU_ERR_INVALID; }
	if (!ctx->have_coverage) { set_last_error("agpu_upload_coverage must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	DeviceBuffer& keys_in = ctx->scratch("events.keys_in"); DeviceBuffer& pair_keys = ctx->scratch("events.keys_out"); DeviceBuffer& gene2_keys = ctx->scratch("events.gene2_keys");
	DeviceBuffer& pair_members = ctx->scratch("events.order_a"); DeviceBuffer& gene2_members = ctx->scratch("events.order_b"); DeviceBuffer& scratch = ctx->scratch("events.rocprim");
	DeviceBuffer& result = ctx->scratch("events.confidence");
	const size_t C1 = std::max<uint32_t>(C, 1);
	ALLOC(keys_in, C1 * 8); ALLOC(pair_keys, C1 * 8); ALLOC(gene2_keys, C1 * 8); ALLOC(pair_members, C1 * 4); ALLOC(gene2_members, C1 * 4); ALLOC(result, C1);
	(void) hipEventRecord(ctx->event_start, s);
	if (C > 0) {
		const CandidateTable& t = ctx->candidates;
		const unsigned int grid = (unsigned int) ((C + BLOCK - 1) / BLOCK);
		size_t bytes = 0;
		confidence_key_kernel<<<grid, BLOCK, 0, s>>>(t, 0, keys_in.as<uint64_t>());
		HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys_in.as<uint64_t>(), pair_keys.as<uint64_t>(), rocprim::counting_iterator<uint32_t>(0), pair_members.as<uint32_t>(), C, 0, 64, s));
		if (bytes > scratch.capacity) ALLOC(scratch, bytes);
		HIP_CHECK(rocprim::radix_sort_pairs(scratch.ptr, bytes, keys_in.as<uint64_t>(), pair_keys.as<uint64_t>(), rocprim::counting_iterator<uint32_t>(0), pair_members.as<uint32_t>(), C, 0, 64, s));
		confidence_key_kernel<<<grid, BLOCK, 0, s>>>(t, 1, keys_in.as<uint64_t>());
		HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys_in.as<uint64_t>(), gene2_keys.as<uint64_t>(), rocprim::counting_iterator<uint32_t>(0), gene2_members.as<uint32_t>(), C, 0, 32, s));
		if (bytes > scratch.capacity) ALLOC(scratch, bytes);
		HIP_CHECK(rocprim::radix_sort_pairs(scratch.ptr, bytes, keys_in.as<uint64_t>(), gene2_keys.as<uint64_t>(), rocprim::counting_iterator<uint32_t>(0), gene2_members.as<uint32_t>(), C, 0, 32, s));
		ConfidenceTables tables;
		tables.pair_keys = pair_keys.as<uint64_t>(); tables.pair_members = pair_members.as<uint32_t>(); tables.gene2_keys = gene2_keys.as<uint64_t>(); tables.gene2_members = gene2_members.as<uint32_t>(); tables.n = C;
		KernelTimer timer(ctx, "confidence_kernel", (uint64_t) C * 60);
		confidence_kernel<<<grid, BLOCK, 0, s>>>(ctx->annotation, ctx->coverage, t, ctx->cand_evalue.as<float>(), tables, ctx->genomic_support(), result.as<uint8_t>());
		ctx->confidence_candidates = C;
	}
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (uint64_t) C * 100;
	if (confidence && C > 0) HIP_CHECK(hipMemcpy(confidence, result.ptr, C, hipMemcpyDeviceToHost));
	return AGPU_OK;
}

extern "C" int agpu_recover_isoforms(agpu_ctx* ctx, uint64_t* remaining) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	if (!ctx->iteration_order_done) { const int status = agpu_candidate_iteration_order(ctx, nullptr); if (status != AGPU_OK) return status; } // hazard H2
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	DeviceBuffer& counter = ctx->scratch("events.counter"); DeviceBuffer& keys_in = ctx->scratch("events.keys_in"); DeviceBuffer& keys_out = ctx->scratch("events.keys_out");
	DeviceBuffer& order_a = ctx->scratch("events.order_a"); DeviceBuffer& order_b = ctx->scratch("events.order_b"); DeviceBuffer& scratch = ctx->scratch("events.rocprim");
	DeviceBuffer& recovered = ctx->scratch("events.eligible");
	const size_t C1 = std::max<uint32_t>(C, 1);
	ALLOC(counter, 16); ALLOC(keys_in, C1 * 8); ALLOC(keys_out, C1 * 8); ALLOC(order_a, C1 * 4); ALLOC(order_b, C1 * 4); ALLOC(recovered, C1);
	HIP_CHECK(hipMemsetAsync(counter.ptr, 0, 16, s));
	(void) hipEventRecord(ctx->event_start, s);
	if (C > 0 && ctx->params.filter_enabled[FILTER_isoforms]) {
		const CandidateTable& t = ctx->candidates;
		const unsigned int grid = (unsigned int) ((C + BLOCK - 1) / BLOCK);
		size_t bytes = 0;
		// stable sorts: by iteration rank, then by gene pair + directions (candidates that did not pass all filters go to the end)
		isoform_key_kernel<<<grid, BLOCK, 0, s>>>(t, nullptr, ctx->cand_iteration_rank.as<uint32_t>(), 0, keys_in.as<uint64_t>());
		HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), rocprim::counting_iterator<uint32_t>(0), order_a.as<uint32_t>(), C, 0, 32, s));
		if (bytes > scratch.capacity) ALLOC(scratch, bytes);
		HIP_CHECK(rocprim::radix_sort_pairs(scratch.ptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), rocprim::counting_iterator<uint32_t>(0), order_a.as<uint32_t>(), C, 0, 32, s));
		isoform_key_kernel<<<grid, BLOCK, 0, s>>>(t, order_a.as<uint32_t>(), ctx->cand_iteration_rank.as<uint32_t>(), 1, keys_in.as<uint64_t>());
		HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), order_a.as<uint32_t>(), order_b.as<uint32_t>(), C, 0, 64, s));
		if (bytes > scratch.capacity) ALLOC(scratch, bytes);
		HIP_CHECK(rocprim::radix_sort_pairs(scratch.ptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), order_a.as<uint32_t>(), order_b.as<uint32_t>(), C, 0, 64, s));
		{ KernelTimer timer(ctx, "isoform_verdict_kernel", (uint64_t) C * 40);
		  isoform_verdict_kernel<<<grid, BLOCK, 0, s>>>(t, keys_out.as<uint64_t>(), order_b.as<uint32_t>(), recovered.as<uint8_t>()); }
		isoform_recover_kernel<<<grid, BLOCK, 0, s>>>(t, recovered.as<uint8_t>());
	}
	if (C > 0) event_predicate_kernel<<<tally_grid(C, BLOCK) * 4, BLOCK, 0, s>>>(EVENT_count_only, ctx->batch, ctx->annotation, ctx->genome, ctx->coverage, ctx->candidates, 0u, counter.as<unsigned int>());
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (uint64_t) C * 100;
	unsigned int kept = 0;
	HIP_CHECK(hipMemcpy(&kept, counter.ptr, 4, hipMemcpyDeviceToHost));
	if (remaining) *remaining = kept;
	return AGPU_OK;
}

namespace {
// clipped: null = the verdicts walk the discordant lists themselves (this context holds every read); or the clipped discordant mates of every candidate [2 * C], counted where
// the reads are (agpu_filter_in_vitro_sharded)
int filter_in_vitro_stage(agpu_ctx* ctx, float high_expression_quantile, const uint32_t* clipped, uint64_t* remaining) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	if (clipped == nullptr && (ctx->candidates_imported || (ctx->global_n != 0 && ctx->global_n != ctx->n))) { set_last_error("filter_in_vitro needs the read lists and all fragments in one context (the reads sharded over the ranks: agpu_in_vitro_clipped_mates / agpu_filter_in_vitro_sharded)"); return AGPU_ERR_INVALID; }
	if (!ctx->have_coverage) { set_last_error("agpu_upload_coverage must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	const uint64_t n = ctx->n;
	const size_t n_genes = (size_t) ctx->n_genes + ctx->n_dummy;
	DeviceBuffer& counter = ctx->scratch("events.counter"); DeviceBuffer& gene_read_count = ctx->scratch("events.gene_read_count"); DeviceBuffer& keys_in = ctx->scratch("events.pair_keys_in");
	(void) n_genes;
	DeviceBuffer& keys_sorted = ctx->scratch("events.pair_keys_sorted"); DeviceBuffer& unique_keys = ctx->scratch("events.pair_unique"); DeviceBuffer& unique_counts = ctx->scratch("events.pair_counts");
	DeviceBuffer& n_runs = ctx->scratch("events.pair_runs"); DeviceBuffer& scratch = ctx->scratch("events.rocprim");
	const size_t C1 = std::max<uint32_t>(C, 1);
	ALLOC(counter, 16); ALLOC(keys_in, C1 * 16); ALLOC(keys_sorted, C1 * 16); ALLOC(unique_keys, C1 * 16); ALLOC(unique_counts, C1 * 8); ALLOC(n_runs, 16);
	HIP_CHECK(hipMemsetAsync(counter.ptr, 0, 16, s));
	(void) hipEventRecord(ctx->event_start, s);
	if (C > 0 && ctx->params.filter_enabled[FILTER_in_vitro]) {
		const CandidateTable& t = ctx->candidates;
		// (1) chimeric fragments per gene and the quantile of the non-zero counts (the quantile is taken on the host: a few 10^4 numbers)
		uint32_t threshold = 0;
		{ const int status = expression_proxy(ctx, high_expression_quantile, threshold); if (status != AGPU_OK) return status; }
		// (2) breakpoints inside exons per ordered gene pair: sort the keys, run lengths
		const unsigned int grid = (unsigned int) ((C + BLOCK - 1) / BLOCK);
		in_vitro_pair_key_kernel<<<grid, BLOCK, 0, s>>>(t, keys_in.as<uint64_t>());
		size_t bytes = 0;
		HIP_CHECK(rocprim::radix_sort_keys(nullptr, bytes, keys_in.as<uint64_t>(), keys_sorted.as<uint64_t>(), 2 * (size_t) C, 0, 64, s));
		if (bytes > scratch.capacity) ALLOC(scratch, bytes);
		HIP_CHECK(rocprim::radix_sort_keys(scratch.ptr, bytes, keys_in.as<uint64_t>(), keys_sorted.as<uint64_t>(), 2 * (size_t) C, 0, 64, s));
		HIP_CHECK(rocprim::run_length_encode(nullptr, bytes, keys_sorted.as<uint64_t>(), 2 * (size_t) C, unique_keys.as<uint64_t>(), unique_counts.as<uint32_t>(), n_runs.as<uint32_t>(), s));
		if (bytes > scratch.capacity) ALLOC(scratch, bytes);
		HIP_CHECK(rocprim::run_length_encode(scratch.ptr, bytes, keys_sorted.as<uint64_t>(), 2 * (size_t) C, unique_keys.as<uint64_t>(), unique_counts.as<uint32_t>(), n_runs.as<uint32_t>(), s));
		uint32_t runs = 0;
		HIP_CHECK(hipMemcpyAsync(&runs, n_runs.ptr, 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		InVitroTables tables;
		tables.gene_read_count = gene_read_count.as<uint32_t>();
		tables.high_expression_threshold = threshold;
		tables.pair_keys = unique_keys.as<uint64_t>(); tables.pair_counts = unique_counts.as<uint32_t>(); tables.n_pairs = runs; // (the run of ~0 keys at the end is never looked up)
		tables.clip_summaries = nullptr; tables.clip_any = nullptr;
		if (clipped != nullptr) { // (3') the verdicts from the counts
			KernelTimer timer(ctx, "in_vitro_verdict_kernel", (uint64_t) C * 88);
			in_vitro_verdict_kernel<<<grid, BLOCK, 0, s>>>(ctx->annotation, ctx->coverage, tables, t, clipped);
		} else {
		if (ctx->n > 0) { // the clipped ends of the alignments, summarised once in 8 bytes each: the verdicts walk the read lists and would otherwise gather CIGAR ends, strand, contig and
			// position of up to three alignments per list entry (10^8 fragments: in_vitro_kernel 219 -> 66 + 3 ms, profiles/r03h_output_side_and_ingest.txt)
			DeviceBuffer& summaries = ctx->scratch("events.clip_summaries"); DeviceBuffer& clip_any = ctx->scratch("events.clip_any");
			ALLOC(summaries, CLIP_SUMMARIES_PER_READ * ctx->n * sizeof(ClipSummary)); ALLOC(clip_any, ctx->n);
			HIP_CHECK(hipMemsetAsync(clip_any.ptr, 0, ctx->n, s));
			{ KernelTimer summary_timer(ctx, "clip_summary_kernel", ctx->n * (3 * 8 + 3 * 20 + 1));
			  clip_summary_kernel<<<(unsigned int) ((3 * ctx->n + BLOCK - 1) / BLOCK), BLOCK, 0, s>>>(ctx->batch, summaries.as<ClipSummary>(), clip_any.as<uint8_t>()); }
			tables.clip_summaries = summaries.as<ClipSummary>(); tables.clip_any = clip_any.as<uint8_t>();
		}
		// (3) the verdicts
		// (the verdict of a candidate looks at the tables made above -- complete -- and at its own discordant list; the filter it sets is read by no other candidate's verdict)
		const int status = for_each_list_window(ctx, [&](const CandidateTable& window, uint32_t begin, uint32_t end) -> int {
			LongLists lists; uint32_t n_long = 0;
			{ const int status = lists.prepare(ctx, end - begin); if (status != AGPU_OK) return status; }
			{ KernelTimer timer(ctx, "in_vitro_kernel", (uint64_t) C * 80 + (uint64_t) ctx->n_list_entries * 4);
			  in_vitro_kernel<<<(unsigned int) ((end - begin + BLOCK - 1) / BLOCK), BLOCK, 0, s>>>(ctx->batch, ctx->annotation, ctx->coverage, tables, window, begin, end, lists.list, lists.count, long_list_entries(LONG_LIST)); }
			{ const int status = lists.noted(ctx, n_long); if (status != AGPU_OK) return status; }
			if (n_long > 0) { KernelTimer timer(ctx, "in_vitro_wave_kernel", (uint64_t) n_long * 80);
			  for_each_wave_chunk(n_long, [&](uint64_t first, uint64_t count) { in_vitro_wave_kernel<<<(unsigned int) ((count * 64 + BLOCK - 1) / BLOCK), BLOCK, 0, s>>>(ctx->batch, ctx->annotation, ctx->coverage, tables, window, lists.list, lists.count, (uint32_t) first); }); }
			return AGPU_OK;
		}, LISTS_OF_IN_VITRO);
		if (status != AGPU_OK) return status;
		}
	}
	if (C > 0) event_predicate_kernel<<<tally_grid(C, BLOCK) * 4, BLOCK, 0, s>>>(EVENT_count_only, ctx->batch, ctx->annotation, ctx->genome, ctx->coverage, ctx->candidates, 0u, counter.as<unsigned int>());
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (uint64_t) C * 120 + n * 22;
	unsigned int kept = 0;
	HIP_CHECK(hipMemcpy(&kept, counter.ptr, 4, hipMemcpyDeviceToHost));
	if (remaining) *remaining = kept;
	return AGPU_OK;
}
}
extern "C" int agpu_filter_in_vitro(agpu_ctx* ctx, float high_expression_quantile, uint64_t* remaining) { return filter_in_vitro_stage(ctx, high_expression_quantile, nullptr, remaining); }

// ---- filter_in_vitro and the expression proxy with the reads sharded over the ranks (include/arriba_gpu.h) ---------------------------------------------------------------
extern "C" int agpu_gene_read_counts(agpu_ctx* ctx, uint32_t* counts) {
	if (!ctx || !ctx->annotated || !counts) { set_last_error("agpu_annotate must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const size_t n_genes = (size_t) ctx->n_genes + ctx->n_dummy;
	DeviceBuffer& mine = ctx->scratch("sharded.gene_read_count");
	ALLOC(mine, std::max<size_t>(n_genes, 1) * 4);
	HIP_CHECK(hipMemsetAsync(mine.ptr, 0, std::max<size_t>(n_genes, 1) * 4, s));
	if (ctx->n > 0) { KernelTimer timer(ctx, "gene_read_count_kernel", ctx->n * 22); gene_read_count_kernel<<<(unsigned int) std::min<uint64_t>((ctx->n + BLOCK - 1) / BLOCK, 2048), BLOCK, 0, s>>>(ctx->batch, mine.as<uint32_t>()); }
	if (n_genes > 0) HIP_CHECK(hipMemcpyAsync(counts, mine.ptr, n_genes * 4, hipMemcpyDefault, s));
	HIP_CHECK(hipStreamSynchronize(s));
	collect_kernel_samples(ctx);
	return AGPU_OK;
}
extern "C" int agpu_set_gene_read_counts(agpu_ctx* ctx, const uint32_t* counts) {
	if (!ctx || !ctx->annotated || !counts) { set_last_error("agpu_annotate must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	const size_t n_genes = (size_t) ctx->n_genes + ctx->n_dummy;
	DeviceBuffer& gene_read_count = ctx->scratch("events.gene_read_count");
	ALLOC(gene_read_count, std::max<size_t>(n_genes, 1) * 4);
	ctx->host_gene_read_counts.assign(n_genes, 0);
	if (n_genes > 0) {
		HIP_CHECK(hipMemcpy(ctx->host_gene_read_counts.data(), counts, n_genes * 4, hipMemcpyDefault));
		HIP_CHECK(hipMemcpy(gene_read_count.ptr, ctx->host_gene_read_counts.data(), n_genes * 4, hipMemcpyHostToDevice));
	}
	ctx->sample_gene_read_counts_set = true;
	return AGPU_OK;
}
extern "C" int agpu_in_vitro_clipped_mates(agpu_ctx* ctx, uint64_t* n_entries) {
	if (!ctx || !ctx->fusions_done || !ctx->read_sharded) { set_last_error("agpu_shard_keep and agpu_find_fusions_from_emissions must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	ctx->n_clipped_entries = 0;
	if (n_entries) *n_entries = 0;
	if (C == 0 || !ctx->params.filter_enabled[FILTER_in_vitro]) return AGPU_OK;
	DeviceBuffer& clipped = ctx->scratch("sharded.clipped"); DeviceBuffer& flags = ctx->scratch("sharded.clipped_flags"); DeviceBuffer& selected = ctx->scratch("sharded.clipped_selected");
	DeviceBuffer& count = ctx->scratch("sharded.counter"); DeviceBuffer& scratch = ctx->scratch("events.rocprim"); DeviceBuffer& entries = ctx->scratch("sharded.clipped_entries");
	ALLOC(clipped, (size_t) C * 8); ALLOC(flags, C); ALLOC(selected, (size_t) C * 4); ALLOC(count, 16);
	HIP_CHECK(hipMemsetAsync(clipped.ptr, 0, (size_t) C * 8, s));
	HIP_CHECK(hipMemsetAsync(count.ptr, 0, 16, s));
	InVitroTables tables; memset(&tables, 0, sizeof(tables));
	if (ctx->n > 0) {
		DeviceBuffer& summaries = ctx->scratch("events.clip_summaries"); DeviceBuffer& clip_any = ctx->scratch("events.clip_any");
		ALLOC(summaries, CLIP_SUMMARIES_PER_READ * ctx->n * sizeof(ClipSummary)); ALLOC(clip_any, ctx->n);
		HIP_CHECK(hipMemsetAsync(clip_any.ptr, 0, ctx->n, s));
		{ KernelTimer summary_timer(ctx, "clip_summary_kernel", ctx->n * (3 * 8 + 3 * 20 + 1));
		  clip_summary_kernel<<<(unsigned int) ((3 * ctx->n + BLOCK - 1) / BLOCK), BLOCK, 0, s>>>(ctx->batch, summaries.as<ClipSummary>(), clip_any.as<uint8_t>()); }
		tables.clip_summaries = summaries.as<ClipSummary>(); tables.clip_any = clip_any.as<uint8_t>();
		const int status = for_each_list_window(ctx, [&](const CandidateTable& window, uint32_t begin, uint32_t end) -> int {
			KernelTimer timer(ctx, "in_vitro_partial_kernel", (uint64_t) (end - begin) * 20 + (uint64_t) ctx->n_list_entries * 4);
			in_vitro_partial_kernel<<<(unsigned int) std::min<uint64_t>(((uint64_t) (end - begin) * 64 + BLOCK - 1) / BLOCK, 1u << 18), BLOCK, 0, s>>>(ctx->batch, tables, window, begin, end, clipped.as<uint32_t>());
			return AGPU_OK;
		}, LISTS_OF_IN_VITRO);
		if (status != AGPU_OK) return status;
	}
	clipped_entry_flag_kernel<<<(C + BLOCK - 1) / BLOCK, BLOCK, 0, s>>>(clipped.as<uint32_t>(), C, flags.as<uint8_t>());
	size_t bytes = 0;
	HIP_CHECK(rocprim::select(nullptr, bytes, rocprim::counting_iterator<uint32_t>(0), flags.as<uint8_t>(), selected.as<uint32_t>(), count.as<uint32_t>(), C, s));
	if (bytes > scratch.capacity) ALLOC(scratch, bytes);
	HIP_CHECK(rocprim::select(scratch.ptr, bytes, rocprim::counting_iterator<uint32_t>(0), flags.as<uint8_t>(), selected.as<uint32_t>(), count.as<uint32_t>(), C, s));
	uint32_t n_selected = 0;
	HIP_CHECK(hipMemcpyAsync(&n_selected, count.ptr, 4, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	ALLOC(entries, std::max<size_t>(n_selected, 1) * AGPU_CLIPPED_MATES_ENTRY_BYTES);
	if (n_selected > 0) clipped_entry_write_kernel<<<(n_selected + BLOCK - 1) / BLOCK, BLOCK, 0, s>>>(clipped.as<uint32_t>(), selected.as<uint32_t>(), n_selected, entries.as<uint32_t>());
	HIP_CHECK(hipStreamSynchronize(s));
	collect_kernel_samples(ctx);
	ctx->n_clipped_entries = n_selected;
	if (n_entries) *n_entries = n_selected;
	return AGPU_OK;
}
extern "C" int agpu_copy_in_vitro_clipped_mates(agpu_ctx* ctx, void* destination) {
	if (!ctx || !destination) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	if (ctx->n_clipped_entries) HIP_CHECK(hipMemcpy(destination, ctx->scratch("sharded.clipped_entries").ptr, (size_t) ctx->n_clipped_entries * AGPU_CLIPPED_MATES_ENTRY_BYTES, hipMemcpyDefault));
	return AGPU_OK;
}
extern "C" int agpu_filter_in_vitro_sharded(agpu_ctx* ctx, float high_expression_quantile, const void* entries, uint64_t n_entries, uint64_t* remaining) {
	if (!ctx || !ctx->fusions_done || !ctx->read_sharded || (n_entries > 0 && !entries)) { set_last_error("agpu_shard_keep and agpu_find_fusions_from_emissions must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	DeviceBuffer& clipped = ctx->scratch("sharded.clipped"); DeviceBuffer& staged = ctx->scratch("sharded.clipped_all"); DeviceBuffer& error = ctx->scratch("sharded.counter");
	ALLOC(clipped, std::max<size_t>(C, 1) * 8); ALLOC(error, 16);
	HIP_CHECK(hipMemsetAsync(clipped.ptr, 0, std::max<size_t>(C, 1) * 8, s));
	HIP_CHECK(hipMemsetAsync(error.ptr, 0, 16, s));
	if (n_entries > 0) {
		ALLOC(staged, n_entries * AGPU_CLIPPED_MATES_ENTRY_BYTES);
		HIP_CHECK(hipMemcpyAsync(staged.ptr, entries, n_entries * AGPU_CLIPPED_MATES_ENTRY_BYTES, hipMemcpyDefault, s));
		clipped_entry_add_kernel<<<(unsigned int) ((n_entries + BLOCK - 1) / BLOCK), BLOCK, 0, s>>>(staged.as<uint32_t>(), n_entries, C, clipped.as<uint32_t>(), error.as<unsigned int>());
		unsigned int bad = 0;
		HIP_CHECK(hipMemcpyAsync(&bad, error.ptr, 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		if (bad) { set_last_error("an entry of the clipped discordant mates names a candidate outside the table"); return AGPU_ERR_INVALID; }
	}
	return filter_in_vitro_stage(ctx, high_expression_quantile, clipped.as<uint32_t>(), remaining);
}

extern "C" int agpu_recover_both_spliced(agpu_ctx* ctx, uint32_t max_fusions_to_recover, float high_expression_quantile, int32_t max_exon_size, uint32_t max_coverage, uint64_t* remaining) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	if (ctx->candidates_imported || (ctx->global_n != 0 && ctx->global_n != ctx->n && !ctx->read_sharded)) { set_last_error("recover_both_spliced needs the read lists and all fragments in one context"); return AGPU_ERR_INVALID; }
	if (!ctx->have_coverage) { set_last_error("agpu_upload_coverage must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	DeviceBuffer& counter = ctx->scratch("events.counter"); DeviceBuffer& reads = ctx->scratch("events.reads"); DeviceBuffer& keys_in = ctx->scratch("events.keys_in"); DeviceBuffer& keys_out = ctx->scratch("events.keys_out");
	DeviceBuffer& members = ctx->scratch("events.order_a"); DeviceBuffer& member_reads = ctx->scratch("events.member_reads"); DeviceBuffer& eligible = ctx->scratch("events.eligible");
	DeviceBuffer& histogram = ctx->scratch("events.histogram"); DeviceBuffer& scratch = ctx->scratch("events.rocprim");
	const size_t C1 = std::max<uint32_t>(C, 1);
	ALLOC(counter, 16); ALLOC(reads, C1 * 4); ALLOC(keys_in, C1 * 8); ALLOC(keys_out, C1 * 8); ALLOC(members, C1 * 4); ALLOC(member_reads, C1 * 4); ALLOC(eligible, C1); ALLOC(histogram, (size_t) BOTH_SPLICED_HISTOGRAM_BINS * 4);
	HIP_CHECK(hipMemsetAsync(counter.ptr, 0, 16, s));
	(void) hipEventRecord(ctx->event_start, s);
	if (C > 0 && ctx->params.filter_enabled[19 /* spliced */]) {
		const CandidateTable& t = ctx->candidates;
		const unsigned int grid = (unsigned int) ((C + BLOCK - 1) / BLOCK);
		uint32_t threshold = 0;
		{ const int status = expression_proxy(ctx, high_expression_quantile, threshold); if (status != AGPU_OK) return status; }
		HIP_CHECK(hipMemsetAsync(histogram.ptr, 0, (size_t) BOTH_SPLICED_HISTOGRAM_BINS * 4, s));
		{ const uint32_t* gene_read_count = ctx->scratch("events.gene_read_count").as<uint32_t>();
		  BatchView batch;
		  { const int status = batch_with_walk_bytes(ctx, batch); if (status != AGPU_OK) return status; }
		  const int status = for_each_list_window(ctx, [&](const CandidateTable& window, uint32_t begin, uint32_t end) -> int {
			LongLists lists; uint32_t n_long = 0;
			{ const int status = lists.prepare(ctx, end - begin); if (status != AGPU_OK) return status; }
			{ KernelTimer timer(ctx, "both_spliced_reads_kernel", (uint64_t) C * 70 + (uint64_t) ctx->n_list_entries * 6);
			  both_spliced_reads_kernel<<<(unsigned int) ((end - begin + BLOCK - 1) / BLOCK), BLOCK, 0, s>>>(batch, ctx->annotation, ctx->coverage, gene_read_count, threshold, window, max_exon_size, max_coverage, reads.as<uint32_t>(), keys_in.as<uint64_t>(), begin, end, lists.list, lists.count, long_list_entries(LONG_LIST_BOTH_SPLICED)); }
			{ const int status = lists.noted(ctx, n_long); if (status != AGPU_OK) return status; }
			if (n_long > 0) { KernelTimer timer(ctx, "both_spliced_reads_wave_kernel", (uint64_t) n_long * 70);
			  for_each_wave_chunk(n_long, [&](uint64_t first, uint64_t count) { both_spliced_reads_wave_kernel<<<(unsigned int) ((count * 64 + BLOCK - 1) / BLOCK), BLOCK, 0, s>>>(batch, ctx->annotation, ctx->coverage, gene_read_count, threshold, window, max_exon_size, max_coverage, reads.as<uint32_t>(), keys_in.as<uint64_t>(), lists.list, lists.count, (uint32_t) first); }); }
			return AGPU_OK;
		  }, LISTS_OF_BOTH_SPLICED);
		  if (status != AGPU_OK) return status; }
		size_t bytes = 0;
		HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), rocprim::counting_iterator<uint32_t>(0), members.as<uint32_t>(), C, 0, 64, s));
		if (bytes > scratch.capacity) ALLOC(scratch, bytes);
		HIP_CHECK(rocprim::radix_sort_pairs(scratch.ptr, bytes, keys_in.as<uint64_t>(), keys_out.as<uint64_t>(), rocprim::counting_iterator<uint32_t>(0), members.as<uint32_t>(), C, 0, 64, s));
		both_spliced_gather_kernel<<<grid, BLOCK, 0, s>>>(members.as<uint32_t>(), reads.as<uint32_t>(), C, member_reads.as<uint32_t>());
		{ KernelTimer timer(ctx, "both_spliced_eligible_kernel", (uint64_t) C * 60);
		  both_spliced_eligible_kernel<<<grid, BLOCK, 0, s>>>(ctx->annotation, t, keys_out.as<uint64_t>(), members.as<uint32_t>(), member_reads.as<uint32_t>(), eligible.as<uint8_t>(), histogram.as<uint32_t>()); }
		std::vector<uint32_t> host_histogram(BOTH_SPLICED_HISTOGRAM_BINS);
		HIP_CHECK(hipMemcpyAsync(host_histogram.data(), histogram.ptr, (size_t) BOTH_SPLICED_HISTOGRAM_BINS * 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		const uint32_t min_supporting_reads = both_spliced_min_supporting_reads(host_histogram, max_fusions_to_recover);
		both_spliced_recover_kernel<<<grid, BLOCK, 0, s>>>(t, eligible.as<uint8_t>(), min_supporting_reads);
	}
	if (C > 0) event_predicate_kernel<<<tally_grid(C, BLOCK) * 4, BLOCK, 0, s>>>(EVENT_count_only, ctx->batch, ctx->annotation, ctx->genome, ctx->coverage, ctx->candidates, 0u, counter.as<unsigned int>());
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (uint64_t) C * 200;
	unsigned int kept = 0;
	HIP_CHECK(hipMemcpy(&kept, counter.ptr, 4, hipMemcpyDeviceToHost));
	if (remaining) *remaining = kept;
	return AGPU_OK;
}

extern "C" int agpu_recover_internal_tandem_duplication(agpu_ctx* ctx, uint32_t min_supporting_reads, float min_fraction_of_coverage, uint64_t* remaining) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	if (ctx->candidates_imported || (ctx->global_n != 0 && ctx->global_n != ctx->n && !ctx->read_sharded)) { set_last_error("recover_internal_tandem_duplication needs the read lists and all fragments in one context"); return AGPU_ERR_INVALID; }
	if (!ctx->have_coverage) { set_last_error("agpu_upload_coverage must run first"); return AGPU_ERR_INVALID; }
	if (!ctx->iteration_order_done) { const int status = agpu_candidate_iteration_order(ctx, nullptr); if (status != AGPU_OK) return status; } // hazard H2
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	// (the reads sharded over the ranks: the stage runs on every rank over the replicated filters of ALL fragments of the sample -- the same verdicts, the same claims, the same
	//  reads cleared everywhere -- and every rank takes the filters of its own reads from there: agpu_sharded.hip)
	BatchView reads;
	{ const int status = candidate_walk_batch(ctx, reads, false); if (status != AGPU_OK) return status; }
	const uint64_t n = reads.n;
	DeviceBuffer& counter = ctx->scratch("events.counter"); DeviceBuffer& verdict = ctx->scratch("events.itd_verdict"); DeviceBuffer& owner = ctx->scratch("events.itd_owner");
	const size_t C1 = std::max<uint32_t>(C, 1), n1 = std::max<uint64_t>(n, 1);
	ALLOC(counter, 16); ALLOC(verdict, C1); ALLOC(owner, n1 * 4);
	HIP_CHECK(hipMemsetAsync(counter.ptr, 0, 16, s));
	(void) hipEventRecord(ctx->event_start, s);
	if (C > 0 && n > 0 && ctx->params.filter_enabled[FILTER_internal_tandem_duplication]) {
		const CandidateTable& t = ctx->candidates;
		const unsigned int grid = (unsigned int) ((C + BLOCK - 1) / BLOCK);
		// the duplication rate, because the coverage includes duplicates (:15-20)
		count_duplicates_kernel<<<tally_grid(n, BLOCK), BLOCK, 0, s>>>(reads, counter.as<unsigned int>() + 1);
		unsigned int duplicates = 0;
		HIP_CHECK(hipMemcpyAsync(&duplicates, counter.as<unsigned int>() + 1, 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		const float duplication_rate = 1.0 * duplicates / n;
		HIP_CHECK(hipMemsetAsync(owner.ptr, 0xFF, n1 * 4, s));
		{ KernelTimer timer(ctx, "itd_verdict_kernel", (uint64_t) C * 40);
		  itd_verdict_kernel<<<grid, BLOCK, 0, s>>>(reads, ctx->annotation, ctx->coverage, t, ctx->params.max_itd_length, min_supporting_reads, min_fraction_of_coverage, ctx->params.subsampling_threshold, duplication_rate,
			ctx->cand_iteration_rank.as<uint32_t>(), verdict.as<uint8_t>(), owner.as<uint32_t>(), counter.as<unsigned int>() + 2); }
		itd_recover_kernel<<<grid, BLOCK, 0, s>>>(reads, t, ctx->cand_iteration_rank.as<uint32_t>(), verdict.as<uint8_t>(), owner.as<uint32_t>());
		itd_clear_reads_kernel<<<(unsigned int) ((n + BLOCK - 1) / BLOCK), BLOCK, 0, s>>>(reads, owner.as<uint32_t>());
		{ const int status = pull_filters_of_own_reads(ctx); if (status != AGPU_OK) return status; }
	}
	if (C > 0) event_predicate_kernel<<<tally_grid(C, BLOCK) * 4, BLOCK, 0, s>>>(EVENT_count_only, ctx->batch, ctx->annotation, ctx->genome, ctx->coverage, ctx->candidates, 0u, counter.as<unsigned int>());
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = (uint64_t) C * 60 + n * 5;
	unsigned int host_counters[3] = { 0, 0, 0 };
	HIP_CHECK(hipMemcpy(host_counters, counter.ptr, sizeof(host_counters), hipMemcpyDeviceToHost));
	if (host_counters[2]) { set_last_error("recover_internal_tandem_duplication: more exons at a breakpoint than a device set holds"); return AGPU_ERR_CAPACITY; }
	if (remaining) *remaining = host_counters[0];
	return AGPU_OK;
}
