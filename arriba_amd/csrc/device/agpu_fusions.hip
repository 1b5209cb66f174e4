// arriba_amd/csrc/device/agpu_fusions.hip -- find_fusions on the device (reference: source/fusions.cpp:203-473).
//
// Pipeline (all arrays resident in HBM; M = emissions, C = candidates):
//   emission_count_kernel + exclusive scan      reads -> M
//   emission_write_kernel                       one record per read x gene1 x gene2, in name order
//   candidate_insert/resolve_kernel             lock-free hash table: representative = first emission with the same key
//   radix sort of (representative << 32 | e)    groups a candidate's emissions, name order inside, candidates in first-occurrence order
//   gather_sorted_kernel                        physical reorder + head flags; inclusive scan -> candidate index
//   rank scan (RankCombine)                     per-side read / unfiltered-read prefix counts -> who joins the read lists
//   fold scan (CandidateCombine)                filter / counts / anchors / exonic per candidate at the segment tails
//   discordant buckets                          discordant emissions sorted by (gene1, gene2, directions), stable in name order
//   attach_discordant_kernel (count + fill)     one thread per unfiltered candidate walks its bucket (early exit at the subsampling threshold)
//   split_list_fill_kernel, finish_kernel       read lists, strands, splice sites, transcript start
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>
#include <rocprim/rocprim.hpp>
#include "agpu_context.hpp"
#include "device_utils.hpp"
#include "fusion_core.hpp"

using namespace agpu;

namespace {

const int BLOCK = 256;
const uint32_t EMPTY_SLOT = 0xFFFFFFFFu;
inline unsigned int grid_for(uint64_t n) { return (unsigned int) ((n + BLOCK - 1) / BLOCK); }

#define HIP_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_last_error(std::string(#call) + ": " + hipGetErrorString(e_)); return AGPU_ERR_DEVICE; } } while (0)
#define ALLOC(buffer, bytes) do { if (!(buffer).allocate(bytes)) { set_last_error("hipMalloc failed (" #buffer ")"); return AGPU_ERR_NO_MEMORY; } } while (0)

__global__ void emission_count_kernel(BatchView b, uint32_t* counts) {
	uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i >= b.n) return;
	counts[i] = emission_count(b, i);
}

__global__ void emission_write_kernel(BatchView b, const uint32_t* offsets, const uint32_t* counts, FusionEmission* emissions) {
	uint64_t i = blockIdx.x * (uint64_t) BLOCK + threadIdx.x;
	if (i >= b.n || counts[i] == 0) return;
	write_emissions(b, i, emissions + offsets[i]);
}

__global__ void candidate_insert_kernel(uint32_t M, const FusionEmission* emissions, uint32_t* slots, uint32_t mask) {
	uint32_t e = blockIdx.x * BLOCK + threadIdx.x;
	if (e >= M) return;
	FusionEmission mine = emissions[e];
	uint32_t h = (uint32_t) hash_candidate(mine) & mask;
	while (true) {
		uint32_t owner = __hip_atomic_load(&slots[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (owner == EMPTY_SLOT) {
			owner = atomicCAS(&slots[h], EMPTY_SLOT, e);
			if (owner == EMPTY_SLOT) return;
		}
		if (same_candidate(emissions[owner], mine)) { if (e < owner) atomicMin(&slots[h], e); return; } // the slot only ever decreases
		h = (h + 1) & mask;
	}
}

__global__ void candidate_resolve_kernel(uint32_t M, const FusionEmission* emissions, const uint32_t* slots, uint32_t mask, uint64_t* keys) {
	uint32_t e = blockIdx.x * BLOCK + threadIdx.x;
	if (e >= M) return;
	FusionEmission mine = emissions[e];
	uint32_t h = (uint32_t) hash_candidate(mine) & mask;
	while (true) {
		uint32_t owner = slots[h];
		if (same_candidate(emissions[owner], mine)) { keys[e] = (uint64_t) owner << 32 | e; return; }
		h = (h + 1) & mask;
	}
}

__global__ void gather_sorted_kernel(uint32_t M, const uint64_t* sorted_keys, const FusionEmission* emissions, FusionEmission* sorted, uint32_t* heads) {
	uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
	if (j >= M) return;
	uint64_t key = sorted_keys[j];
	sorted[j] = emissions[(uint32_t) key];
	heads[j] = (j == 0 || (sorted_keys[j - 1] >> 32) != (key >> 32)) ? 1u : 0u;
}

__global__ void rank_input_kernel(uint32_t M, const FusionEmission* sorted, const uint32_t* heads, RankState* out) {
	uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
	if (j >= M) return;
	out[j] = rank_input(sorted[j], heads[j]);
}

__global__ void fold_input_kernel(uint32_t M, const FusionEmission* sorted, const uint32_t* heads, const RankState* ranks, uint32_t threshold, CandidateFold* out) {
	uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
	if (j >= M) return;
	FusionEmission e = sorted[j];
	bool joins = (e.info & EINFO_SPLIT) && joins_split_read_list(e, ranks[j], threshold);
	out[j] = candidate_input(e, heads[j], joins);
}

__global__ void candidate_write_kernel(uint32_t M, const FusionEmission* sorted, const uint32_t* heads, const uint32_t* candidate_of, const CandidateFold* folds, CandidateTable t, uint32_t* list_size) {
	uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
	if (j >= M) return;
	if (!(j + 1 == M || heads[j + 1])) return; // only segment tails carry the aggregate
	uint32_t c = candidate_of[j] - 1;
	FusionEmission e = sorted[j];
	CandidateFold f = folds[j];
	t.gene1[c] = e.gene1; t.gene2[c] = e.gene2; t.contigs[c] = e.contigs; t.breakpoint1[c] = e.breakpoint1; t.breakpoint2[c] = e.breakpoint2;
	t.flags[c] = ((e.info & EINFO_UPSTREAM1) ? CFLAG_UPSTREAM1 : 0) | ((e.info & EINFO_UPSTREAM2) ? CFLAG_UPSTREAM2 : 0) | ((f.info_or & EINFO_EXONIC1) ? CFLAG_EXONIC1 : 0) | ((f.info_or & EINFO_EXONIC2) ? CFLAG_EXONIC2 : 0);
	t.filter[c] = candidate_filter(f);
	t.split_reads1[c] = f.split_reads[0]; t.split_reads2[c] = f.split_reads[1]; t.discordant_mates[c] = 0;
	t.anchor1[c] = anchor_apply(0, f.anchor1, e.info & EINFO_UPSTREAM1);
	t.anchor2[c] = anchor_apply(0, f.anchor2, e.info & EINFO_UPSTREAM2);
	list_size[3 * (uint64_t) c] = f.list_size[0]; list_size[3 * (uint64_t) c + 1] = f.list_size[1]; list_size[3 * (uint64_t) c + 2] = 0;
}

// sharded samples: emissions travel to the rank that owns their gene pair (all candidates and discordant mates of a gene pair meet there)
AGPU_HD uint32_t emission_owner(const FusionEmission& e, uint32_t n_partitions) {
	uint64_t h = ((uint64_t) e.gene1 << 32 | e.gene2) * 0x9E3779B97F4A7C15ULL;
	h ^= h >> 31; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
	return (uint32_t) (h % n_partitions);
}
__global__ void emission_owner_kernel(uint32_t M, const FusionEmission* emissions, uint32_t n_partitions, uint32_t* owners) {
	uint32_t e = blockIdx.x * BLOCK + threadIdx.x;
	if (e < M) owners[e] = emission_owner(emissions[e], n_partitions);
}
__global__ void emission_permute_kernel(uint32_t M, const uint32_t* order, const FusionEmission* emissions, FusionEmission* out) {
	uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
	if (k < M) out[k] = emissions[order[k]];
}
__global__ void partition_offsets_kernel(const uint32_t* sorted_owners, uint32_t M, uint32_t n_partitions, uint32_t* offsets) {
	uint32_t p = blockIdx.x * BLOCK + threadIdx.x;
	if (p > n_partitions) return;
	uint32_t lo = 0, hi = M;
	while (lo < hi) { uint32_t mid = lo + ((hi - lo) >> 1); if (sorted_owners[mid] < p) lo = mid + 1; else hi = mid; }
	offsets[p] = lo;
}
// first occurrence of a candidate in the name order of the whole sample: name rank of the read << 8 | position among the read's emissions
__global__ void candidate_first_occurrence_kernel(uint32_t M, const FusionEmission* sorted, const uint32_t* heads, const uint32_t* candidate_of, uint64_t* first_occurrence) {
	uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
	if (j >= M || !heads[j]) return;
	first_occurrence[candidate_of[j] - 1] = (uint64_t) sorted[j].read << 8 | (sorted[j].info >> EINFO_ORDINAL_SHIFT & 255u);
}

__global__ void discordant_flag_kernel(uint32_t M, const FusionEmission* emissions, uint8_t* flags) {
	uint32_t e = blockIdx.x * BLOCK + threadIdx.x;
	if (e >= M) return;
	flags[e] = !(emissions[e].info & EINFO_SPLIT);
}
AGPU_HD uint64_t gene_pair_key(uint32_t gene1, uint32_t gene2, uint32_t direction_bits) { return (uint64_t) gene1 << 33 | (uint64_t) gene2 << 2 | (direction_bits & 3u); }
__global__ void discordant_key_kernel(uint32_t Md, const uint32_t* indices, const FusionEmission* emissions, uint64_t* keys) {
	uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
	if (k >= Md) return;
	const FusionEmission& e = emissions[indices[k]];
	keys[k] = gene_pair_key(e.gene1, e.gene2, e.info);
}
struct DiscordantBucketColumns { int32_t* breakpoint1; int32_t* breakpoint2; uint32_t* info; uint32_t* read; int32_t* anchor1; int32_t* anchor2; };
__global__ void discordant_gather_kernel(uint32_t Md, const uint32_t* sorted_indices, const FusionEmission* emissions, DiscordantBucketColumns out) {
	uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
	if (k >= Md) return;
	FusionEmission e = emissions[sorted_indices[k]];
	out.breakpoint1[k] = e.breakpoint1; out.breakpoint2[k] = e.breakpoint2; out.info[k] = e.info; out.read[k] = e.read; out.anchor1[k] = e.anchor1; out.anchor2[k] = e.anchor2;
}

AGPU_HD uint32_t lower_bound_key(const uint64_t* keys, uint32_t n, uint64_t value) {
	uint32_t lo = 0, hi = n;
	while (lo < hi) { uint32_t mid = lo + ((hi - lo) >> 1); if (keys[mid] < value) lo = mid + 1; else hi = mid; }
	return lo;
}

const uint32_t SMALL_BUCKET = 24;  // buckets up to this size are walked by the candidate's own thread, larger ones by a whole wave
const uint32_t SMALL_LIST = 32;

struct BucketRef { uint32_t candidate, begin, end; };

// What a candidate keeps of its discordant-mate list when the lists are implicit (fusion_core.hpp: CandidateTable::discordant_before): the range of its bucket and whether it had
// split reads when find_fusions looked (merge_adjacent_fusions appends split reads later; the predicate of the list must not change with them).  Kept for explicit lists too.
struct BucketRanges { uint32_t* begin; uint32_t* end; uint8_t* had_split_reads; };

// does the stage an expansion is made for read the lists of candidate c? (for_each_list_window)
__device__ __forceinline__ bool lists_wanted(const AnnotationView& ann, const CandidateTable& t, uint32_t c, int lists_of) {
	if (lists_of == LISTS_OF_UNFILTERED) return t.filter[c] == FILTER_none;
	if (lists_of == LISTS_OF_IN_VITRO) return in_vitro_looks_at(t, c);
	if (lists_of == LISTS_OF_BOTH_SPLICED) return both_spliced_is_member(ann, t, c);
	return true;
}
// one thread per candidate of [c_begin, c_end): small buckets are handled inline, large ones are queued for the wave kernel.
// mode: ATTACH_COUNT the list sizes (and the bucket ranges noted), ATTACH_FILL / ATTACH_FOLD the pass behind it (lists written / not written; anchors, votes, counters),
// ATTACH_EXPAND the lists of a window of candidates written again (t = the table of the window)
__global__ void __launch_bounds__(BLOCK) attach_discordant_kernel(AnnotationView ann, CandidateTable t, uint32_t c_begin, uint32_t c_end, const uint64_t* bucket_keys, DiscordantBuckets buckets, uint32_t Md, BucketRanges ranges,
                                         int32_t max_mate_gap, uint32_t threshold, uint32_t* list_size, uint8_t* discordant_swapped, BucketRef* worklist, uint32_t* worklist_size, int mode, int lists_of) {
	__shared__ uint32_t wave_offset[BLOCK / 64];
	__shared__ uint32_t block_base;
	const uint32_t c = c_begin + blockIdx.x * BLOCK + threadIdx.x;
	uint32_t begin = 0, end = 0;
	bool has_split_reads = false;
	if (c < c_end) {
		if (mode == ATTACH_COUNT) {
			if (t.filter[c] == FILTER_none) {
				const uint32_t flags = t.flags[c];
				const uint64_t key = gene_pair_key(t.gene1[c], t.gene2[c], ((flags & CFLAG_UPSTREAM1) ? 1u : 0u) | ((flags & CFLAG_UPSTREAM2) ? 2u : 0u));
				begin = lower_bound_key(bucket_keys, Md, key); end = lower_bound_key(bucket_keys, Md, key + 1);
			}
			has_split_reads = list_size[3 * (uint64_t) c] + list_size[3 * (uint64_t) c + 1] > 0;
			ranges.begin[c] = begin; ranges.end[c] = end; ranges.had_split_reads[c] = has_split_reads;
		} else if (mode != ATTACH_EXPAND || (t.list_offset[3 * (uint64_t) c + 3] > t.list_offset[3 * (uint64_t) c + 2] && lists_wanted(ann, t, c, lists_of))) { // (a list that was empty stays empty)
			begin = ranges.begin[c]; end = ranges.end[c]; has_split_reads = ranges.had_split_reads[c];
		}
	}
	const bool queue = end - begin > SMALL_BUCKET;
	const uint32_t at = block_append<BLOCK>(queue ? 1u : 0u, worklist_size, wave_offset, &block_base); // one atomic per workgroup
	if (queue) {
		BucketRef ref; ref.candidate = c; ref.begin = begin; ref.end = end;
		worklist[at] = ref;
		return;
	}
	if (begin == end) return;
	uint32_t* out_list = mode == ATTACH_FILL || mode == ATTACH_EXPAND ? t.read_lists + t.list_offset[3 * (uint64_t) c + 2] : nullptr;
	uint32_t size = attach_discordant_mates(ann, t, c, buckets, begin, end - begin, max_mate_gap, threshold, has_split_reads, out_list, discordant_swapped, mode);
	if (mode == ATTACH_COUNT) list_size[3 * (uint64_t) c + 2] = size;
}

__device__ __forceinline__ AnchorFold wave_fold_in_lane_order(AnchorFold mine, bool upstream) {
	for (int offset = 1; offset < 64; offset <<= 1) {
		AnchorFold other;
		other.has_zero = __shfl_down(mine.has_zero, offset);
		other.value = __shfl_down(mine.value, offset);
		mine = anchor_combine(mine, other, upstream);
	}
	return mine; // lane 0 holds the fold over lanes 0..63 in order
}
// order-free accumulation of anchors that are all non-zero: 0 = unset
__device__ __forceinline__ int32_t anchor_merge(int32_t a, int32_t b, bool upstream) {
	if (a == 0) return b;
	if (b == 0) return a;
	return upstream ? (a > b ? a : b) : (a < b ? a : b);
}

// one wave per queued candidate: 64 discordant mates are tested at once; the order-dependent subsampling rule of the reference
// (source/fusions.cpp:398-407) becomes prefix popcounts over the ballots.  MODE 0: count the list entries; MODE 1: write the list
// and accumulate the anchors per lane (valid while no downstream anchor is 0; WRITE_LIST false: implicit lists, everything but the list); MODE 2: anchors only, folded in name
// order (the rare case of an anchor at position 0, which resets a running minimum); MODE 3: the list only (an implicit list expanded for a window of candidates).
template <int MODE, bool WRITE_LIST = true> __device__ __forceinline__ void scan_bucket(const AnnotationView& ann, const DiscordantBuckets& buckets, const BucketRef& ref, uint32_t gene1, uint32_t gene2,
		int32_t breakpoint1, int32_t breakpoint2, bool upstream1, bool upstream2, bool has_split_reads, int32_t max_mate_gap, uint32_t threshold, uint32_t lane,
		uint32_t* out_list, uint8_t* discordant_swapped, uint32_t& unfiltered, uint32_t& appended, int32_t& lane_anchor1, int32_t& lane_anchor2, uint32_t& lane_votes, bool& zero_seen, AnchorFold& fold1, AnchorFold& fold2) {
	// SCAN_UNROLL x 64 rows of the bucket in flight: a wavefront alone with its bucket waits for three dependent loads per 64 rows (the breakpoints of the mates, the info word of
	// those that pass, the read of those that join) -- with -U 32767 a candidate of a hot gene pair walks tens of thousands of rows (10^8 fragments of config 3: 92.7 G rows per
	// pass over the candidates, profiles/r05f).  The loads of four chunks are issued together; what depends on the order of the mates (the positions among those that pass, the
	// cut-off at the threshold) is then settled chunk by chunk, as before.
	const int SCAN_UNROLL = 4;
	const unsigned long long lanes_before = (1ull << lane) - 1;
	DiscordantMatePredicate predicate;
	predicate.set(ann, gene1, gene2, breakpoint1, breakpoint2, upstream1, upstream2, has_split_reads, max_mate_gap);
	uint32_t passing = 0;
	unfiltered = 0; appended = 0;
	for (uint32_t base = ref.begin; base < ref.end; base += 64 * SCAN_UNROLL) {
		int32_t mate1[SCAN_UNROLL], mate2[SCAN_UNROLL]; uint32_t info[SCAN_UNROLL], read[SCAN_UNROLL]; bool pass[SCAN_UNROLL];
		AGPU_UNROLL for (int u = 0; u < SCAN_UNROLL; ++u) { const uint32_t k = base + 64 * u + lane; const bool inside = k < ref.end; mate1[u] = inside ? buckets.breakpoint1[k] : 0; mate2[u] = inside ? buckets.breakpoint2[k] : 0; pass[u] = inside; }
		AGPU_UNROLL for (int u = 0; u < SCAN_UNROLL; ++u) pass[u] = pass[u] && predicate.supports(mate1[u], mate2[u]);
		AGPU_UNROLL for (int u = 0; u < SCAN_UNROLL; ++u) { const uint32_t k = base + 64 * u + lane; info[u] = pass[u] ? buckets.info[k] : 0; read[u] = (MODE == 1 || MODE == 3) && pass[u] ? buckets.read[k] : 0; }
		bool done = false;
		AGPU_UNROLL for (int u = 0; u < SCAN_UNROLL; ++u) {
			if (done || base + 64 * u >= ref.end) break; // (the same for every lane)
			const uint32_t k = base + 64 * u + lane;
			const bool is_unfiltered = pass[u] && (info[u] >> EINFO_FILTER_SHIFT & 255) == FILTER_none;
			const unsigned long long ballot_pass = __ballot(pass[u]), ballot_unfiltered = __ballot(is_unfiltered);
			const uint32_t position = passing + __popcll(ballot_pass & lanes_before);
			const uint32_t unfiltered_before = unfiltered + __popcll(ballot_unfiltered & lanes_before);
			const bool joins = pass[u] && (position < threshold || (is_unfiltered && unfiltered_before < threshold));
			const unsigned long long ballot_joins = __ballot(joins);
			if (MODE == 3 && joins) out_list[appended + __popcll(ballot_joins & lanes_before)] = read[u]; // (an implicit list written again: the entries, nothing else)
			if (MODE == 1 && joins) {
				if (WRITE_LIST) out_list[appended + __popcll(ballot_joins & lanes_before)] = read[u];
				if ((info[u] & EINFO_MATES_SWAPPED) && !discordant_swapped[read[u]]) discordant_swapped[read[u]] = 1;
				const int32_t anchor1 = buckets.anchor1[k], anchor2 = buckets.anchor2[k];
				if ((!upstream1 && anchor1 == 0) || (!upstream2 && anchor2 == 0)) zero_seen = true;
				lane_anchor1 = anchor_merge(lane_anchor1, anchor1, upstream1);
				lane_anchor2 = anchor_merge(lane_anchor2, anchor2, upstream2);
				const int vote = discordant_mate_vote(info[u], upstream1, upstream2, breakpoint1, breakpoint2, mate1[u], mate2[u]);
				lane_votes += (vote == 1) ? 1u : (vote == 2) ? 0x10000u : 0u; // forward in the low half, reverse in the high half (<= 64 K entries per lane)
			}
			if (MODE == 2 && ballot_joins != 0) {
				AnchorFold chunk1 = wave_fold_in_lane_order(joins ? anchor_single(buckets.anchor1[k], upstream1) : anchor_identity(), upstream1);
				AnchorFold chunk2 = wave_fold_in_lane_order(joins ? anchor_single(buckets.anchor2[k], upstream2) : anchor_identity(), upstream2);
				fold1 = anchor_combine(fold1, chunk1, upstream1);
				fold2 = anchor_combine(fold2, chunk2, upstream2);
			}
			passing += __popcll(ballot_pass);
			unfiltered += __popcll(ballot_unfiltered);
			appended += __popcll(ballot_joins);
			if (unfiltered >= threshold) done = true; // the reference stops at the next unfiltered mate; filtered ones no longer fit either
		}
		if (done) break;
	}
}

__global__ void attach_discordant_wave_kernel(AnnotationView ann, CandidateTable t, DiscordantBuckets buckets, BucketRanges ranges, int32_t max_mate_gap, uint32_t threshold,
                                              uint32_t* list_size, uint8_t* discordant_swapped, const BucketRef* worklist, const uint32_t* worklist_size, int mode, uint32_t first_wave = 0) {
	const uint32_t wave = first_wave + ((blockIdx.x * BLOCK + threadIdx.x) >> 6), lane = threadIdx.x & 63;
	if (wave >= *worklist_size) return;
	const BucketRef ref = worklist[wave];
	const uint32_t c = ref.candidate;
	const uint32_t flags = t.flags[c];
	const bool upstream1 = flags & CFLAG_UPSTREAM1, upstream2 = flags & CFLAG_UPSTREAM2;
	const uint32_t gene1 = t.gene1[c], gene2 = t.gene2[c];
	const int32_t breakpoint1 = t.breakpoint1[c], breakpoint2 = t.breakpoint2[c];
	uint32_t unfiltered = 0, appended = 0;
	int32_t lane_anchor1 = 0, lane_anchor2 = 0;
	uint32_t lane_votes = 0;
	bool zero_seen = false;
	AnchorFold fold1 = anchor_identity(), fold2 = anchor_identity();
	const bool has_split_reads = ranges.had_split_reads[c];
	if (mode == ATTACH_COUNT) {
		scan_bucket<0>(ann, buckets, ref, gene1, gene2, breakpoint1, breakpoint2, upstream1, upstream2, has_split_reads, max_mate_gap, threshold, lane, nullptr, nullptr, unfiltered, appended, lane_anchor1, lane_anchor2, lane_votes, zero_seen, fold1, fold2);
		if (lane == 0) list_size[3 * (uint64_t) c + 2] = appended;
		return;
	}
	uint32_t* out_list = mode == ATTACH_FOLD ? nullptr : t.read_lists + t.list_offset[3 * (uint64_t) c + 2];
	if (mode == ATTACH_EXPAND) {
		scan_bucket<3>(ann, buckets, ref, gene1, gene2, breakpoint1, breakpoint2, upstream1, upstream2, has_split_reads, max_mate_gap, threshold, lane, out_list, nullptr, unfiltered, appended, lane_anchor1, lane_anchor2, lane_votes, zero_seen, fold1, fold2);
		return;
	}
	if (mode == ATTACH_FILL) scan_bucket<1, true>(ann, buckets, ref, gene1, gene2, breakpoint1, breakpoint2, upstream1, upstream2, has_split_reads, max_mate_gap, threshold, lane, out_list, discordant_swapped, unfiltered, appended, lane_anchor1, lane_anchor2, lane_votes, zero_seen, fold1, fold2);
	else scan_bucket<1, false>(ann, buckets, ref, gene1, gene2, breakpoint1, breakpoint2, upstream1, upstream2, has_split_reads, max_mate_gap, threshold, lane, nullptr, discordant_swapped, unfiltered, appended, lane_anchor1, lane_anchor2, lane_votes, zero_seen, fold1, fold2);
	if (__ballot(zero_seen) != 0) {
		uint32_t unfiltered_again, appended_again, lane_votes_again = 0;
		scan_bucket<2>(ann, buckets, ref, gene1, gene2, breakpoint1, breakpoint2, upstream1, upstream2, has_split_reads, max_mate_gap, threshold, lane, nullptr, nullptr, unfiltered_again, appended_again, lane_anchor1, lane_anchor2, lane_votes_again, zero_seen, fold1, fold2);
	} else {
		for (int offset = 32; offset > 0; offset >>= 1) {
			lane_anchor1 = anchor_merge(lane_anchor1, __shfl_down(lane_anchor1, offset), upstream1);
			lane_anchor2 = anchor_merge(lane_anchor2, __shfl_down(lane_anchor2, offset), upstream2);
		}
		fold1.value = lane_anchor1; fold2.value = lane_anchor2;
	}
	uint32_t forward_votes = lane_votes & 0xFFFFu, reverse_votes = lane_votes >> 16;
	for (int offset = 32; offset > 0; offset >>= 1) { forward_votes += __shfl_down(forward_votes, offset); reverse_votes += __shfl_down(reverse_votes, offset); }
	if (lane == 0) {
		t.votes[2 * (uint64_t) c] += forward_votes; t.votes[2 * (uint64_t) c + 1] += reverse_votes;
		t.discordant_mates[c] = unfiltered < threshold ? unfiltered : threshold;
		t.anchor1[c] = anchor_apply(t.anchor1[c], fold1, upstream1);
		t.anchor2[c] = anchor_apply(t.anchor2[c], fold2, upstream2);
	}
}

// the split-read lists of the candidates of a window into the buffer of the window (one wavefront per candidate)
__global__ void window_split_copy_kernel(AnnotationView ann, CandidateTable sample, CandidateTable window, uint32_t c_begin, uint32_t c_end, int lists_of) { // (c_begin: of this launch's chunk of the window)
	const uint32_t c = c_begin + ((blockIdx.x * BLOCK + threadIdx.x) >> 6), lane = threadIdx.x & 63;
	if (c >= c_end || !lists_wanted(ann, sample, c, lists_of)) return;
	const uint64_t begin = sample.list_offset[3 * (uint64_t) c], end = sample.list_offset[3 * (uint64_t) c + 2];
	for (uint64_t k = begin + lane; k < end; k += 64) window.read_lists[k] = split_list_entry(sample, c, k);
}
// where the windows are cut: as many candidates as fit `entries` list entries (a candidate alone may hold up to 3 x the threshold); cuts[0] = number of cuts, then the cuts
__global__ void window_cut_kernel(const uint64_t* list_offset, uint32_t n_candidates, uint64_t entries, uint32_t capacity, uint32_t* cuts) {
	if (blockIdx.x != 0 || threadIdx.x != 0) return;
	uint32_t n = 0, c = 0;
	while (c < n_candidates && n + 2 < capacity) {
		const uint64_t limit = list_offset[3 * (uint64_t) c] + entries;
		uint32_t lo = c + 1, hi = n_candidates; // the last candidate whose lists end at or in front of `limit` (at least one candidate per window)
		while (lo < hi) { const uint32_t mid = lo + ((hi - lo + 1) >> 1); if (list_offset[3 * (uint64_t) mid] <= limit) lo = mid; else hi = mid - 1; }
		c = lo;
		cuts[1 + n++] = c;
	}
	if (c < n_candidates) cuts[1 + n++] = n_candidates; // (more windows than there is room to note: the last one takes the rest and its allocation says so if it does not fit)
	cuts[0] = n;
}
__global__ void discordant_size_kernel(const uint32_t* list_size, uint32_t n_candidates, uint64_t* sizes) {
	const uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c <= n_candidates) sizes[c] = c < n_candidates ? list_size[3 * (uint64_t) c + 2] : 0;
}

__global__ void split_list_fill_kernel(uint32_t M, const FusionEmission* sorted, const uint32_t* candidate_of, const RankState* ranks, const CandidateFold* folds, uint32_t threshold, CandidateTable t) {
	uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
	if (j >= M) return;
	FusionEmission e = sorted[j];
	if (!(e.info & EINFO_SPLIT) || !joins_split_read_list(e, ranks[j], threshold)) return;
	int side = (e.info & EINFO_SWAPPED) ? 1 : 0;
	uint32_t c = candidate_of[j] - 1;
	t.read_lists[t.list_offset[3 * (uint64_t) c + side] + folds[j].list_size[side] - 1 - (t.discordant_before != nullptr ? t.discordant_before[c] : 0)] = e.read;
	int vote = split_read_vote(e.info);
	if (vote != 0) atomicAdd(&t.votes[2 * (uint64_t) c + (vote - 1)], 1u);
}

// strands / splice sites / transcript start from the strand votes collected while the read lists were filled
__global__ void finish_kernel(AnnotationView ann, CandidateTable t) {
	uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
	if (c >= t.n) return;
	finalize_candidate(ann, t, c, t.votes[2 * (uint64_t) c], t.votes[2 * (uint64_t) c + 1]);
}

struct Scratch { // grows on demand; reused by every rocprim call
	DeviceBuffer& buffer;
	explicit Scratch(DeviceBuffer& pooled) : buffer(pooled) {}
	int ensure(size_t bytes) { if (bytes > buffer.capacity) { if (!buffer.allocate(bytes + (bytes >> 2))) { set_last_error("hipMalloc failed (scratch)"); return AGPU_ERR_NO_MEMORY; } } return AGPU_OK; }
};

}

namespace {

// ---- implicit discordant lists: the windows (fusion_core.hpp: CandidateTable::discordant_before) ------------------------------------------------------------------------------

// how many list entries a window holds: ARRIBA_LIST_WINDOW_ENTRIES (tests: a few hundred, so that a toy sample has dozens of windows), else a quarter of what the device has free,
// at most 8 GB of entries
uint64_t list_window_entries(uint32_t threshold) {
	const char* knob = getenv("ARRIBA_LIST_WINDOW_ENTRIES");
	uint64_t entries = knob != nullptr && atoll(knob) > 0 ? (uint64_t) atoll(knob) : 0;
	if (entries == 0) {
		size_t free_bytes = 0, total_bytes = 0;
		(void) hipMemGetInfo(&free_bytes, &total_bytes);
		entries = std::min<uint64_t>(8ull << 30, free_bytes / 4) / 4;
	}
	return std::max<uint64_t>(entries, 1024);
}
int cut_list_windows(agpu_ctx* ctx) {
	hipStream_t s = ctx->stream;
	const uint32_t C = ctx->n_candidates;
	const uint32_t capacity = 1u << 16;
	DeviceBuffer& cuts = ctx->scratch("lists.cuts");
	ALLOC(cuts, (size_t) capacity * 4);
	ctx->list_window_entries = list_window_entries(ctx->params.subsampling_threshold);
	window_cut_kernel<<<1, 64, 0, s>>>(ctx->candidates.list_offset, C, ctx->list_window_entries, capacity, cuts.as<uint32_t>());
	uint32_t n = 0;
	HIP_CHECK(hipMemcpyAsync(&n, cuts.ptr, 4, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	ctx->list_window_cuts.assign((size_t) n + 1, 0);
	if (n > 0) HIP_CHECK(hipMemcpy(ctx->list_window_cuts.data() + 1, cuts.as<uint32_t>() + 1, (size_t) n * 4, hipMemcpyDeviceToHost));
	return AGPU_OK;
}
// all three lists of the candidates [c_begin, c_end) at their positions in a buffer: `window` = the table of the sample with read_lists pointing at that buffer (rebased) and no
// discordant_before -- the kernels of the stages walk it as they walk explicit lists
int expand_list_window(agpu_ctx* ctx, uint32_t c_begin, uint32_t c_end, CandidateTable& window, int lists_of, bool zero_fill) {
	hipStream_t s = ctx->stream;
	const CandidateTable& t = ctx->candidates;
	uint64_t bounds[2] = { 0, 0 };
	HIP_CHECK(hipMemcpyAsync(&bounds[0], t.list_offset + 3 * (uint64_t) c_begin, 8, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipMemcpyAsync(&bounds[1], t.list_offset + 3 * (uint64_t) c_end, 8, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	const uint64_t entries = bounds[1] - bounds[0];
	DeviceBuffer& buffer = ctx->scratch("lists.window");
	if (!buffer.allocate((size_t) std::max<uint64_t>(entries, 1) * 4)) { set_last_error("hipMalloc failed (a window of " + std::to_string(entries) + " read-list entries)"); return AGPU_ERR_NO_MEMORY; }
	window = t;
	window.read_lists = buffer.as<uint32_t>() - bounds[0]; window.discordant_before = nullptr;
	if (c_end == c_begin || entries == 0) return AGPU_OK;
	if (zero_fill) HIP_CHECK(hipMemsetAsync(buffer.ptr, 0, (size_t) entries * 4, s));
	DeviceBuffer& bucket_worklist = ctx->scratch("fusions.bucket_worklist"); DeviceBuffer& worklist_sizes = ctx->scratch("fusions.worklist_sizes");
	DiscordantBuckets buckets; BucketRanges ranges;
	{ const size_t Md1 = std::max<uint32_t>(ctx->lists_n_bucket_rows, 1);
	  const int32_t* columns = ctx->scratch("fusions.bucket_columns").as<int32_t>();
	  buckets.breakpoint1 = columns; buckets.breakpoint2 = columns + Md1; buckets.info = (const uint32_t*) (columns + 2 * Md1); buckets.read = buckets.info + Md1; buckets.anchor1 = (const int32_t*) (buckets.read + Md1); buckets.anchor2 = buckets.anchor1 + Md1;
	  ranges.begin = ctx->scratch("lists.bucket_begin").as<uint32_t>(); ranges.end = ctx->scratch("lists.bucket_end").as<uint32_t>(); ranges.had_split_reads = ctx->scratch("lists.had_split_reads").as<uint8_t>(); }
	uint32_t* worklist_count = worklist_sizes.as<uint32_t>() + 2;
	HIP_CHECK(hipMemsetAsync(worklist_count, 0, 4, s));
	const uint32_t n = c_end - c_begin, threshold = ctx->params.subsampling_threshold;
	{ KernelTimer timer(ctx, "window_split_copy_kernel", (uint64_t) n * 16);
	  for_each_wave_chunk(n, [&](uint64_t first, uint64_t count) { window_split_copy_kernel<<<grid_for(count * 64), BLOCK, 0, s>>>(ctx->annotation, t, window, c_begin + (uint32_t) first, c_begin + (uint32_t) (first + count), lists_of); }); }
	{ KernelTimer timer(ctx, "attach_discordant_kernel(window)", (uint64_t) n * 25);
	  attach_discordant_kernel<<<grid_for(n), BLOCK, 0, s>>>(ctx->annotation, window, c_begin, c_end, nullptr, buckets, ctx->lists_n_bucket_rows, ranges, ctx->lists_max_mate_gap, threshold, nullptr, nullptr, bucket_worklist.as<BucketRef>(), worklist_count, ATTACH_EXPAND, lists_of); }
	uint32_t queued = 0;
	HIP_CHECK(hipMemcpyAsync(&queued, worklist_count, 4, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	if (queued > 0) {
		KernelTimer timer(ctx, "attach_discordant_wave_kernel(window)", (uint64_t) entries * 4 + (uint64_t) queued * 25);
		for_each_wave_chunk(queued, [&](uint64_t first, uint64_t count) { attach_discordant_wave_kernel<<<grid_for(count * 64), BLOCK, 0, s>>>(ctx->annotation, window, buckets, ranges, ctx->lists_max_mate_gap, threshold, nullptr, nullptr, bucket_worklist.as<BucketRef>(), worklist_count, ATTACH_EXPAND, (uint32_t) first); });
	}
	return AGPU_OK;
}

}

// Runs `stage` over all candidates, a window at a time: once over [0, n) with the table of the sample while the lists are explicit; with implicit discordant lists over every window
// with the table of the window (the lists of its candidates expanded).  What `stage` launches must restrict itself to the candidates [begin, end).
int agpu::for_each_list_window(agpu_ctx* ctx, const std::function<int(const CandidateTable&, uint32_t, uint32_t)>& stage, int lists_of, bool zero_fill) {
	if (!ctx->lists_implicit) return stage(ctx->candidates, 0u, ctx->n_candidates);
	for (size_t w = 0; w + 1 < ctx->list_window_cuts.size(); ++w) {
		CandidateTable window;
		int status = expand_list_window(ctx, ctx->list_window_cuts[w], ctx->list_window_cuts[w + 1], window, lists_of, zero_fill);
		if (status == AGPU_OK) status = stage(window, ctx->list_window_cuts[w], ctx->list_window_cuts[w + 1]);
		if (status != AGPU_OK) return status;
	}
	return AGPU_OK;
}
int agpu::recut_list_windows(agpu_ctx* ctx) { return ctx->lists_implicit ? cut_list_windows(ctx) : AGPU_OK; }

namespace {

// emissions of the fragments of this context (one record per read x gene1 x gene2, name order); read ids are global name ranks
int build_emissions(agpu_ctx* ctx, uint32_t& M) {
	hipStream_t s = ctx->stream;
	const uint64_t n = ctx->n;
	Scratch scratch(ctx->scratch("fusions.rocprim"));
	size_t bytes = 0;
	DeviceBuffer& counts = ctx->scratch("fusions.counts"); DeviceBuffer& offsets = ctx->scratch("fusions.offsets");
	ALLOC(counts, (n + 1) * 4); ALLOC(offsets, (n + 1) * 4);
	HIP_CHECK(hipMemsetAsync(counts.ptr, 0, (n + 1) * 4, s));
	{ KernelTimer timer(ctx, "emission_count_kernel", (uint64_t) n * (1 + 3 * (2 + 4 + 4 + 1 + 1) + 4)); if (n > 0) emission_count_kernel<<<grid_for(n), BLOCK, 0, s>>>(ctx->batch, counts.as<uint32_t>()); }
	HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, counts.as<uint32_t>(), offsets.as<uint32_t>(), 0u, n + 1, rocprim::plus<uint32_t>(), s));
	if (scratch.ensure(bytes) != AGPU_OK) return AGPU_ERR_DEVICE;
	{ KernelTimer timer(ctx, "rocprim::exclusive_scan(emission offsets)", (uint64_t) n * 8);
	  HIP_CHECK(rocprim::exclusive_scan(scratch.buffer.ptr, bytes, counts.as<uint32_t>(), offsets.as<uint32_t>(), 0u, n + 1, rocprim::plus<uint32_t>(), s)); }
	M = 0;
	HIP_CHECK(hipMemcpyAsync(&M, offsets.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	if (M >= 0xFFFFFFF0u) { set_last_error("too many gene-pair emissions for one batch; shard the input"); return AGPU_ERR_CAPACITY; }
	ctx->n_emissions = M;
	ALLOC(ctx->emissions, (size_t) std::max<uint32_t>(M, 1) * sizeof(FusionEmission));
	if (M > 0) {
		KernelTimer timer(ctx, "emission_write_kernel", (uint64_t) n * (1 + 3 * (2 + 4 + 4 + 1 + 1 + GENE_INLINE * 4) + 1 + 8) + (uint64_t) M * sizeof(FusionEmission));
		emission_write_kernel<<<grid_for(n), BLOCK, 0, s>>>(ctx->batch, offsets.as<uint32_t>(), counts.as<uint32_t>(), ctx->emissions.as<FusionEmission>());
	}
	return AGPU_OK;
}

// candidates from M emissions in name order (reads carry global name ranks; n_reads = number of fragments the ranks refer to)
int candidates_from_emissions(agpu_ctx* ctx, FusionEmission* emissions, uint32_t M, uint64_t n_reads, int32_t max_mate_gap, uint64_t* n_candidates) {
	hipStream_t s = ctx->stream;
	const uint64_t n = n_reads;
	const uint32_t threshold = ctx->params.subsampling_threshold;
	Scratch scratch(ctx->scratch("fusions.rocprim"));
	size_t bytes = 0;
	ctx->n_candidates = 0;
	if (M == 0) {
		if (n_candidates) *n_candidates = 0;
		ctx->fusions_done = true; ctx->candidates_imported = false; ctx->n_list_entries = 0; ctx->genomic_support_marked = false; ctx->confidence_candidates = 0xFFFFFFFFu;
		ctx->lists_implicit = false; ctx->list_window_cuts.clear(); ctx->candidates.discordant_before = nullptr;
		return AGPU_OK;
	}

	// ---- group by candidate key
	uint64_t slots = 1024;
	while (slots < 2ull * M) slots <<= 1;
	const uint32_t mask = (uint32_t) (slots - 1);
	// The working arrays of this function (220 B per emission: 29 GB for the 1.3e8 emissions of 10^8 fragments) are carved out of the buffer that filter_mismappers uses for its
	// memo tables (agpu_mismappers.hip: 41 GB at the default size), which is idle until then and which nothing of this function outlives: what stays (the candidate table, the
	// bucket columns and ranges of the implicit lists) has buffers of its own.  On a session with two lanes and the finish ahead (agpu_keep_batch_buffers) this is what lets
	// 10^8 fragments fit: each lane then holds a batch of its own.
	DeviceBuffer table, keys, sorted_keys, sorted, heads, candidate_of, rank_in, ranks, fold_in, folds;
	{
		struct { DeviceBuffer* view; size_t bytes; } parts[] = { { &table, slots * 4 }, { &keys, (size_t) M * 8 }, { &sorted_keys, (size_t) M * 8 }, { &sorted, (size_t) M * sizeof(FusionEmission) }, { &heads, (size_t) M * 4 },
			{ &candidate_of, (size_t) M * 4 }, { &rank_in, (size_t) M * sizeof(RankState) }, { &ranks, (size_t) M * sizeof(RankState) }, { &fold_in, (size_t) M * sizeof(CandidateFold) }, { &folds, (size_t) M * sizeof(CandidateFold) } };
		size_t total = 0;
		for (auto& part : parts) total += (part.bytes + 255) & ~(size_t) 255;
		DeviceBuffer& arena = ctx->scratch("mismappers.memo_tables");
		ALLOC(arena, std::max(total, arena.capacity));
		size_t at = 0;
		for (auto& part : parts) { part.view->borrow((uint8_t*) arena.ptr + at, part.bytes); at += (part.bytes + 255) & ~(size_t) 255; }
	}
	HIP_CHECK(hipMemsetAsync(table.ptr, 0xFF, slots * 4, s));
	{ KernelTimer timer(ctx, "candidate_insert_kernel", (uint64_t) M * (sizeof(FusionEmission) + 4)); candidate_insert_kernel<<<grid_for(M), BLOCK, 0, s>>>(M, emissions, table.as<uint32_t>(), mask); }
	{ KernelTimer timer(ctx, "candidate_resolve_kernel", (uint64_t) M * (sizeof(FusionEmission) + 4 + 8)); candidate_resolve_kernel<<<grid_for(M), BLOCK, 0, s>>>(M, emissions, table.as<uint32_t>(), mask, keys.as<uint64_t>()); }
	HIP_CHECK(rocprim::radix_sort_keys(nullptr, bytes, keys.as<uint64_t>(), sorted_keys.as<uint64_t>(), M, 0, 64, s));
	if (scratch.ensure(bytes) != AGPU_OK) return AGPU_ERR_DEVICE;
	{ KernelTimer timer(ctx, "rocprim::radix_sort_keys(emissions by candidate)", (uint64_t) M * 16);
	  HIP_CHECK(rocprim::radix_sort_keys(scratch.buffer.ptr, bytes, keys.as<uint64_t>(), sorted_keys.as<uint64_t>(), M, 0, 64, s)); }
	{ KernelTimer timer(ctx, "gather_sorted_kernel", (uint64_t) M * (8 + 2 * sizeof(FusionEmission) + 4)); gather_sorted_kernel<<<grid_for(M), BLOCK, 0, s>>>(M, sorted_keys.as<uint64_t>(), emissions, sorted.as<FusionEmission>(), heads.as<uint32_t>()); }
	HIP_CHECK(rocprim::inclusive_scan(nullptr, bytes, heads.as<uint32_t>(), candidate_of.as<uint32_t>(), M, rocprim::plus<uint32_t>(), s));
	if (scratch.ensure(bytes) != AGPU_OK) return AGPU_ERR_DEVICE;
	{ KernelTimer timer(ctx, "rocprim::inclusive_scan(candidate ids)", (uint64_t) M * 8);
	  HIP_CHECK(rocprim::inclusive_scan(scratch.buffer.ptr, bytes, heads.as<uint32_t>(), candidate_of.as<uint32_t>(), M, rocprim::plus<uint32_t>(), s)); }
	uint32_t C = 0;
	HIP_CHECK(hipMemcpyAsync(&C, candidate_of.as<uint32_t>() + (M - 1), 4, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));

	// ---- prefix counts and per-candidate folds
	{ KernelTimer timer(ctx, "rank_input_kernel", (uint64_t) M * (sizeof(FusionEmission) + 4 + sizeof(RankState))); rank_input_kernel<<<grid_for(M), BLOCK, 0, s>>>(M, sorted.as<FusionEmission>(), heads.as<uint32_t>(), rank_in.as<RankState>()); }
	HIP_CHECK(rocprim::inclusive_scan(nullptr, bytes, rank_in.as<RankState>(), ranks.as<RankState>(), M, RankCombine(), s));
	if (scratch.ensure(bytes) != AGPU_OK) return AGPU_ERR_DEVICE;
	{ KernelTimer timer(ctx, "rocprim::inclusive_scan(RankCombine)", (uint64_t) M * 2 * sizeof(RankState));
	  HIP_CHECK(rocprim::inclusive_scan(scratch.buffer.ptr, bytes, rank_in.as<RankState>(), ranks.as<RankState>(), M, RankCombine(), s)); }
	{ KernelTimer timer(ctx, "fold_input_kernel", (uint64_t) M * (sizeof(FusionEmission) + 4 + sizeof(RankState) + sizeof(CandidateFold))); fold_input_kernel<<<grid_for(M), BLOCK, 0, s>>>(M, sorted.as<FusionEmission>(), heads.as<uint32_t>(), ranks.as<RankState>(), threshold, fold_in.as<CandidateFold>()); }
	HIP_CHECK(rocprim::inclusive_scan(nullptr, bytes, fold_in.as<CandidateFold>(), folds.as<CandidateFold>(), M, CandidateCombine(), s));
	if (scratch.ensure(bytes) != AGPU_OK) return AGPU_ERR_DEVICE;
	{ KernelTimer timer(ctx, "rocprim::inclusive_scan(CandidateCombine)", (uint64_t) M * 2 * sizeof(CandidateFold));
	  HIP_CHECK(rocprim::inclusive_scan(scratch.buffer.ptr, bytes, fold_in.as<CandidateFold>(), folds.as<CandidateFold>(), M, CandidateCombine(), s)); }

	// ---- candidate table
	ALLOC(ctx->cand_gene1, (size_t) C * 4); ALLOC(ctx->cand_gene2, (size_t) C * 4); ALLOC(ctx->cand_contigs, (size_t) C * 4); ALLOC(ctx->cand_breakpoint1, (size_t) C * 4); ALLOC(ctx->cand_breakpoint2, (size_t) C * 4);
	ALLOC(ctx->cand_flags, (size_t) C * 4); ALLOC(ctx->cand_filter, (size_t) C); ALLOC(ctx->cand_split_reads1, (size_t) C * 4); ALLOC(ctx->cand_split_reads2, (size_t) C * 4); ALLOC(ctx->cand_discordant_mates, (size_t) C * 4);
	ALLOC(ctx->cand_anchor1, (size_t) C * 4); ALLOC(ctx->cand_anchor2, (size_t) C * 4); ALLOC(ctx->cand_list_offset, (3 * (size_t) C + 1) * 8);
	ALLOC(ctx->discordant_swapped, n ? n : 1);
	HIP_CHECK(hipMemsetAsync(ctx->discordant_swapped.ptr, 0, n ? n : 1, s));
	CandidateTable& t = ctx->candidates;
	t.n = C; t.gene1 = ctx->cand_gene1.as<uint32_t>(); t.gene2 = ctx->cand_gene2.as<uint32_t>(); t.contigs = ctx->cand_contigs.as<uint32_t>();
	t.breakpoint1 = ctx->cand_breakpoint1.as<int32_t>(); t.breakpoint2 = ctx->cand_breakpoint2.as<int32_t>(); t.flags = ctx->cand_flags.as<uint32_t>(); t.filter = ctx->cand_filter.as<uint8_t>();
	t.split_reads1 = ctx->cand_split_reads1.as<uint32_t>(); t.split_reads2 = ctx->cand_split_reads2.as<uint32_t>(); t.discordant_mates = ctx->cand_discordant_mates.as<uint32_t>();
	t.anchor1 = ctx->cand_anchor1.as<int32_t>(); t.anchor2 = ctx->cand_anchor2.as<int32_t>(); t.list_offset = ctx->cand_list_offset.as<uint64_t>(); t.read_lists = nullptr;
	ALLOC(ctx->cand_votes, (size_t) C * 8);
	HIP_CHECK(hipMemsetAsync(ctx->cand_votes.ptr, 0, (size_t) C * 8, s));
	t.votes = ctx->cand_votes.as<uint32_t>();
	DeviceBuffer& list_size = ctx->scratch("fusions.list_size");
	ALLOC(list_size, (3 * (size_t) C + 1) * 4);
	HIP_CHECK(hipMemsetAsync(list_size.ptr, 0, (3 * (size_t) C + 1) * 4, s));
	{ KernelTimer timer(ctx, "candidate_write_kernel", (uint64_t) M * (4 + 4) + (uint64_t) C * (sizeof(FusionEmission) + sizeof(CandidateFold) + 53)); candidate_write_kernel<<<grid_for(M), BLOCK, 0, s>>>(M, sorted.as<FusionEmission>(), heads.as<uint32_t>(), candidate_of.as<uint32_t>(), folds.as<CandidateFold>(), t, list_size.as<uint32_t>()); }

	ALLOC(ctx->cand_first_occurrence, (size_t) C * 8);
	candidate_first_occurrence_kernel<<<grid_for(M), BLOCK, 0, s>>>(M, sorted.as<FusionEmission>(), heads.as<uint32_t>(), candidate_of.as<uint32_t>(), ctx->cand_first_occurrence.as<uint64_t>());

	// ---- discordant buckets by gene pair
	DeviceBuffer& discordant_flags = ctx->scratch("fusions.discordant_flags"); DeviceBuffer& discordant_indices = ctx->scratch("fusions.discordant_indices"); DeviceBuffer& selected_count = ctx->scratch("fusions.selected_count"); DeviceBuffer& bucket_keys_in = ctx->scratch("fusions.bucket_keys_in"); DeviceBuffer& bucket_keys = ctx->scratch("fusions.bucket_keys"); DeviceBuffer& bucket_indices = ctx->scratch("fusions.bucket_indices"); DeviceBuffer& bucket_columns = ctx->scratch("fusions.bucket_columns");
	ALLOC(discordant_flags, M); ALLOC(discordant_indices, (size_t) M * 4); ALLOC(selected_count, 8);
	{ KernelTimer timer(ctx, "discordant_flag_kernel", (uint64_t) M * (4 + 1)); discordant_flag_kernel<<<grid_for(M), BLOCK, 0, s>>>(M, emissions, discordant_flags.as<uint8_t>()); }
	HIP_CHECK(rocprim::select(nullptr, bytes, rocprim::counting_iterator<uint32_t>(0), discordant_flags.as<uint8_t>(), discordant_indices.as<uint32_t>(), selected_count.as<uint32_t>(), M, s));
	if (scratch.ensure(bytes) != AGPU_OK) return AGPU_ERR_DEVICE;
	{ KernelTimer timer(ctx, "rocprim::select(discordant)", (uint64_t) M * (1 + 4));
	  HIP_CHECK(rocprim::select(scratch.buffer.ptr, bytes, rocprim::counting_iterator<uint32_t>(0), discordant_flags.as<uint8_t>(), discordant_indices.as<uint32_t>(), selected_count.as<uint32_t>(), M, s)); }
	uint32_t Md = 0;
	HIP_CHECK(hipMemcpyAsync(&Md, selected_count.ptr, 4, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	const size_t Md1 = std::max<uint32_t>(Md, 1);
	ALLOC(bucket_keys, Md1 * 8); ALLOC(bucket_columns, Md1 * 6 * 4);
	DiscordantBucketColumns columns;
	columns.breakpoint1 = bucket_columns.as<int32_t>(); columns.breakpoint2 = columns.breakpoint1 + Md1; columns.info = (uint32_t*) (columns.breakpoint2 + Md1); columns.read = columns.info + Md1;
	columns.anchor1 = (int32_t*) (columns.read + Md1); columns.anchor2 = columns.anchor1 + Md1;
	DiscordantBuckets buckets;
	buckets.breakpoint1 = columns.breakpoint1; buckets.breakpoint2 = columns.breakpoint2; buckets.info = columns.info; buckets.read = columns.read; buckets.anchor1 = columns.anchor1; buckets.anchor2 = columns.anchor2;
	if (Md > 0) {
		ALLOC(bucket_keys_in, (size_t) Md * 8); ALLOC(bucket_indices, (size_t) Md * 4);
		{ KernelTimer timer(ctx, "discordant_key_kernel", (uint64_t) Md * (4 + 12 + 8)); discordant_key_kernel<<<grid_for(Md), BLOCK, 0, s>>>(Md, discordant_indices.as<uint32_t>(), emissions, bucket_keys_in.as<uint64_t>()); }
		HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, bucket_keys_in.as<uint64_t>(), bucket_keys.as<uint64_t>(), discordant_indices.as<uint32_t>(), bucket_indices.as<uint32_t>(), Md, 0, 64, s));
		if (scratch.ensure(bytes) != AGPU_OK) return AGPU_ERR_DEVICE;
		{ KernelTimer timer(ctx, "rocprim::radix_sort_pairs(buckets)", (uint64_t) Md * 2 * (8 + 4));
		  HIP_CHECK(rocprim::radix_sort_pairs(scratch.buffer.ptr, bytes, bucket_keys_in.as<uint64_t>(), bucket_keys.as<uint64_t>(), discordant_indices.as<uint32_t>(), bucket_indices.as<uint32_t>(), Md, 0, 64, s)); }
		{ KernelTimer timer(ctx, "discordant_gather_kernel", (uint64_t) Md * (4 + sizeof(FusionEmission) + 24)); discordant_gather_kernel<<<grid_for(Md), BLOCK, 0, s>>>(Md, bucket_indices.as<uint32_t>(), emissions, columns); }
	}

	// ---- discordant mates: count, offsets, fill
	DeviceBuffer& bucket_worklist = ctx->scratch("fusions.bucket_worklist"); DeviceBuffer& worklist_sizes = ctx->scratch("fusions.worklist_sizes");
	DeviceBuffer& range_begin = ctx->scratch("lists.bucket_begin"); DeviceBuffer& range_end = ctx->scratch("lists.bucket_end"); DeviceBuffer& range_split = ctx->scratch("lists.had_split_reads");
	ALLOC(bucket_worklist, (size_t) C * sizeof(BucketRef)); ALLOC(worklist_sizes, 16); ALLOC(range_begin, (size_t) C * 4); ALLOC(range_end, (size_t) C * 4); ALLOC(range_split, (size_t) C);
	HIP_CHECK(hipMemsetAsync(worklist_sizes.ptr, 0, 16, s));
	uint32_t* worklist_counts = worklist_sizes.as<uint32_t>(); // [0] attach (count pass), [1] attach (fill pass), [2] windows
	BucketRanges ranges; ranges.begin = range_begin.as<uint32_t>(); ranges.end = range_end.as<uint32_t>(); ranges.had_split_reads = range_split.as<uint8_t>();
	{ KernelTimer timer(ctx, "attach_discordant_kernel(count)", (uint64_t) C * 25);
	  attach_discordant_kernel<<<grid_for(C), BLOCK, 0, s>>>(ctx->annotation, t, 0, C, bucket_keys.as<uint64_t>(), buckets, Md, ranges, max_mate_gap, threshold, list_size.as<uint32_t>(), ctx->discordant_swapped.as<uint8_t>(), bucket_worklist.as<BucketRef>(), worklist_counts + 0, ATTACH_COUNT, LISTS_OF_ALL); }
	uint32_t queued = 0;
	HIP_CHECK(hipMemcpyAsync(&queued, worklist_counts + 0, 4, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	if (queued > 0) {
		KernelTimer timer(ctx, "attach_discordant_wave_kernel(count)", (uint64_t) Md * 24 + (uint64_t) queued * 25); // every bucket row read once, one size written per queued candidate
		for_each_wave_chunk(queued, [&](uint64_t first, uint64_t count) { attach_discordant_wave_kernel<<<grid_for(count * 64), BLOCK, 0, s>>>(ctx->annotation, t, buckets, ranges, max_mate_gap, threshold, list_size.as<uint32_t>(), ctx->discordant_swapped.as<uint8_t>(), bucket_worklist.as<BucketRef>(), worklist_counts + 0, ATTACH_COUNT, (uint32_t) first); });
	}
	// the lists are addressed with 64-bit offsets (sizes of single lists are 32-bit): with -U 32767 (BASELINE.json config 3) a few million fragments already make more than 2^32
	// entries, every candidate of a gene pair listing the discordant mates of the pair (source/fusions.cpp:398-407 lets a list grow to the subsampling threshold)
	HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, list_size.as<uint32_t>(), t.list_offset, (uint64_t) 0, 3 * (size_t) C + 1, rocprim::plus<uint64_t>(), s));
	if (scratch.ensure(bytes) != AGPU_OK) return AGPU_ERR_DEVICE;
	HIP_CHECK(rocprim::exclusive_scan(scratch.buffer.ptr, bytes, list_size.as<uint32_t>(), t.list_offset, (uint64_t) 0, 3 * (size_t) C + 1, rocprim::plus<uint64_t>(), s));
	uint64_t total_list = 0;
	HIP_CHECK(hipMemcpyAsync(&total_list, t.list_offset + 3 * (size_t) C, 8, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	// Explicit or implicit discordant lists (fusion_core.hpp: CandidateTable::discordant_before)?  Explicit while they fit a budget (ARRIBA_LIST_BUDGET_GB, default 48: the 10 M
	// stress sample of config 3 lists 9 G reads = 36 GB and runs fastest with them in memory) and the device has the memory; implicit otherwise, or when ARRIBA_IMPLICIT_LISTS=1 asks
	// for it (tests: the two ways give the same files).  A context that holds one shard of a sample keeps explicit lists: they stay with the owners of the gene pairs.
	ctx->lists_implicit = false; ctx->list_window_cuts.clear(); ctx->lists_max_mate_gap = max_mate_gap; ctx->lists_n_bucket_rows = Md;
	t.discordant_before = nullptr;
	{
		const char* knob = getenv("ARRIBA_IMPLICIT_LISTS"); const char* budget_knob = getenv("ARRIBA_LIST_BUDGET_GB");
		const double budget_gb = budget_knob != nullptr && atof(budget_knob) > 0 ? atof(budget_knob) : 48.0;
		const bool sharded = ctx->global_n != 0 && ctx->global_n != ctx->n && !ctx->read_sharded; // (the candidates of a gene pair with their owner: round 1.  The reads sharded and every candidate on every rank -- agpu_sharded.hip -- walks windows of lists like one context)
		bool implicit = !sharded && ((knob != nullptr && knob[0] == '1') || (double) total_list * 4 > budget_gb * 1e9);
		if (!implicit && !ctx->cand_read_lists.allocate((size_t) std::max<uint64_t>(total_list, 1) * 4)) {
			if (sharded) { set_last_error("the read lists of the candidates hold " + std::to_string(total_list) + " entries (" + std::to_string(total_list * 4 >> 30) + " GB): more than the device has free; lower -U or shard the input"); return AGPU_ERR_CAPACITY; }
			implicit = true;
		}
		ctx->lists_implicit = implicit;
	}
	uint64_t total_discordant = 0;
	if (ctx->lists_implicit) {
		DeviceBuffer& sizes = ctx->scratch("lists.discordant_sizes"); DeviceBuffer& before = ctx->scratch("lists.discordant_before");
		ALLOC(sizes, ((size_t) C + 1) * 8); ALLOC(before, ((size_t) C + 1) * 8);
		discordant_size_kernel<<<grid_for((uint64_t) C + 1), BLOCK, 0, s>>>(list_size.as<uint32_t>(), C, sizes.as<uint64_t>());
		HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, sizes.as<uint64_t>(), before.as<uint64_t>(), (uint64_t) 0, (size_t) C + 1, rocprim::plus<uint64_t>(), s));
		if (scratch.ensure(bytes) != AGPU_OK) return AGPU_ERR_DEVICE;
		HIP_CHECK(rocprim::exclusive_scan(scratch.buffer.ptr, bytes, sizes.as<uint64_t>(), before.as<uint64_t>(), (uint64_t) 0, (size_t) C + 1, rocprim::plus<uint64_t>(), s));
		HIP_CHECK(hipMemcpyAsync(&total_discordant, before.as<uint64_t>() + C, 8, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
		if (!ctx->cand_read_lists.allocate((size_t) std::max<uint64_t>(total_list - total_discordant, 1) * 4)) { set_last_error("hipMalloc failed (the split-read lists of the candidates)"); return AGPU_ERR_NO_MEMORY; }
		t.discordant_before = before.as<uint64_t>();
	}
	t.read_lists = ctx->cand_read_lists.as<uint32_t>();
	ctx->n_list_entries = total_list;
	{ KernelTimer timer(ctx, "split_list_fill_kernel", (uint64_t) M * (sizeof(FusionEmission) + sizeof(RankState)));
	  split_list_fill_kernel<<<grid_for(M), BLOCK, 0, s>>>(M, sorted.as<FusionEmission>(), candidate_of.as<uint32_t>(), ranks.as<RankState>(), folds.as<CandidateFold>(), threshold, t); }
	const int fill_mode = ctx->lists_implicit ? ATTACH_FOLD : ATTACH_FILL;
	{ KernelTimer timer(ctx, "attach_discordant_kernel(fill)", (uint64_t) C * 25);
	  attach_discordant_kernel<<<grid_for(C), BLOCK, 0, s>>>(ctx->annotation, t, 0, C, bucket_keys.as<uint64_t>(), buckets, Md, ranges, max_mate_gap, threshold, list_size.as<uint32_t>(), ctx->discordant_swapped.as<uint8_t>(), bucket_worklist.as<BucketRef>(), worklist_counts + 1, fill_mode, LISTS_OF_ALL); }
	if (queued > 0) { // the fill pass queues the same candidates (possibly in another order)
		// algorithmic bytes: every bucket row (two breakpoints, info, read, two anchors: 24 B) read once, every list entry written once (4 B; total_list
		// also counts the few split-read entries), the queued candidates' columns.  What the kernel really moves is several times more (PMC): the
		// candidates of one gene pair scan the same bucket rows one after the other.
		KernelTimer timer(ctx, "attach_discordant_wave_kernel(fill)", (uint64_t) Md * 24 + (ctx->lists_implicit ? 0 : (uint64_t) total_list * 4) + (uint64_t) queued * 25);
		for_each_wave_chunk(queued, [&](uint64_t first, uint64_t count) { attach_discordant_wave_kernel<<<grid_for(count * 64), BLOCK, 0, s>>>(ctx->annotation, t, buckets, ranges, max_mate_gap, threshold, list_size.as<uint32_t>(), ctx->discordant_swapped.as<uint8_t>(), bucket_worklist.as<BucketRef>(), worklist_counts + 1, fill_mode, (uint32_t) first); });
	}
	ctx->n_candidates = C;
	if (ctx->lists_implicit) { const int status = cut_list_windows(ctx); if (status != AGPU_OK) return status; }
	{ KernelTimer timer(ctx, "finish_kernel", (uint64_t) C * 53); finish_kernel<<<grid_for(C), BLOCK, 0, s>>>(ctx->annotation, t); }
	ctx->n_queued_buckets = queued;
	ctx->n_discordant_emissions = Md;

	HIP_CHECK(hipStreamSynchronize(s));
	// algorithmic bytes: fragment end columns + gene sets read once, one emission written, candidate table + lists written
	ctx->last_bytes = n * (3 * (2 + 4 + 4 + 1) + 3 * (1 + GENE_INLINE * 4) + 1) + (uint64_t) M * sizeof(FusionEmission) + (uint64_t) C * 53 + (uint64_t) total_list * 4;
	ctx->n_candidates = C;
	ctx->fusions_done = true; ctx->candidates_imported = false; ctx->genomic_support_marked = false; ctx->confidence_candidates = 0xFFFFFFFFu;
	if (n_candidates) *n_candidates = C;
	return AGPU_OK;
}

}

extern "C" int agpu_find_fusions(agpu_ctx* ctx, int32_t max_mate_gap, uint64_t* n_candidates) {
	if (!ctx || !ctx->stage2_done) { set_last_error("agpu_read_filters_stage2 must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	(void) hipEventRecord(ctx->event_start, ctx->stream);
	uint32_t M = 0;
	int status = build_emissions(ctx, M);
	if (status != AGPU_OK) return status;
	status = candidates_from_emissions(ctx, ctx->emissions.as<FusionEmission>(), M, ctx->n, max_mate_gap, n_candidates);
	if (status != AGPU_OK) return status;
	HIP_CHECK(hipEventRecord(ctx->event_stop, ctx->stream));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	return AGPU_OK;
}

extern "C" int agpu_get_candidates(agpu_ctx* ctx, uint32_t* gene1, uint32_t* gene2, uint32_t* contigs, int32_t* breakpoint1, int32_t* breakpoint2, uint32_t* flags, uint8_t* filter,
                                   uint32_t* split_reads1, uint32_t* split_reads2, uint32_t* discordant_mates, int32_t* anchor1, int32_t* anchor2, uint64_t* list_offset) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	if (!ctx->failed_launch.empty()) { set_last_error("a kernel of this sample was not launched (" + ctx->failed_launch + "): its results are not to be read"); return AGPU_ERR_DEVICE; } // (advisor, round 5: not only agpu_select_candidates)
	HIP_CHECK(hipSetDevice(ctx->device));
	HIP_CHECK(hipStreamSynchronize(ctx->stream));
	const size_t C = ctx->n_candidates;
	if (C == 0) { if (list_offset) list_offset[0] = 0; return AGPU_OK; }
	struct { void* host; const DeviceBuffer* device; size_t bytes; } copies[] = {
		{ gene1, &ctx->cand_gene1, C * 4 }, { gene2, &ctx->cand_gene2, C * 4 }, { contigs, &ctx->cand_contigs, C * 4 }, { breakpoint1, &ctx->cand_breakpoint1, C * 4 }, { breakpoint2, &ctx->cand_breakpoint2, C * 4 },
		{ flags, &ctx->cand_flags, C * 4 }, { filter, &ctx->cand_filter, C }, { split_reads1, &ctx->cand_split_reads1, C * 4 }, { split_reads2, &ctx->cand_split_reads2, C * 4 },
		{ discordant_mates, &ctx->cand_discordant_mates, C * 4 }, { anchor1, &ctx->cand_anchor1, C * 4 }, { anchor2, &ctx->cand_anchor2, C * 4 }, { list_offset, &ctx->cand_list_offset, (3 * C + 1) * 8 } };
	for (size_t k = 0; k < sizeof(copies) / sizeof(copies[0]); ++k)
		if (copies[k].host) HIP_CHECK(hipMemcpy(copies[k].host, copies[k].device->ptr, copies[k].bytes, hipMemcpyDefault)); // host or device destination
	return AGPU_OK;
}

extern "C" int agpu_get_candidate_read_lists(agpu_ctx* ctx, uint32_t* reads, uint64_t capacity, uint64_t* total) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	if (!ctx->failed_launch.empty()) { set_last_error("a kernel of this sample was not launched (" + ctx->failed_launch + "): its results are not to be read"); return AGPU_ERR_DEVICE; }
	HIP_CHECK(hipSetDevice(ctx->device));
	HIP_CHECK(hipStreamSynchronize(ctx->stream));
	if (total) *total = ctx->n_list_entries;
	if (reads && ctx->n_list_entries > 0 && ctx->lists_implicit) { // window after window (what the tests of the implicit lists compare; a sample whose lists are implicit for their number has no host that could take them)
		if (capacity < ctx->n_list_entries) { set_last_error("the read lists are implicit: all of them or none"); return AGPU_ERR_INVALID; }
		return for_each_list_window(ctx, [&](const CandidateTable& window, uint32_t begin, uint32_t end) -> int {
			uint64_t bounds[2] = { 0, 0 };
			HIP_CHECK(hipMemcpy(&bounds[0], window.list_offset + 3 * (uint64_t) begin, 8, hipMemcpyDeviceToHost));
			HIP_CHECK(hipMemcpy(&bounds[1], window.list_offset + 3 * (uint64_t) end, 8, hipMemcpyDeviceToHost));
			if (bounds[1] > bounds[0]) HIP_CHECK(hipMemcpy(reads + bounds[0], window.read_lists + bounds[0], (size_t) (bounds[1] - bounds[0]) * 4, hipMemcpyDeviceToHost));
			return AGPU_OK;
		});
	}
	if (reads && ctx->n_list_entries > 0) HIP_CHECK(hipMemcpy(reads, ctx->cand_read_lists.ptr, std::min<uint64_t>(capacity, ctx->n_list_entries) * 4, hipMemcpyDeviceToHost));
	return AGPU_OK;
}

// the read lists of some candidates only (the writer of the output files needs those of the candidates it prints, a few thousand of millions)
__global__ void list_sizes_of_kernel(CandidateTable t, const uint32_t* candidates, uint32_t n, uint32_t* sizes) {
	const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
	if (k > 3 * n) return;
	if (k == 3 * n) { sizes[k] = 0; return; }
	const uint64_t at = 3 * (uint64_t) candidates[k / 3] + k % 3;
	sizes[k] = (uint32_t) (t.list_offset[at + 1] - t.list_offset[at]);
}
// (one wavefront per list, in a loop: most lists hold a handful of reads, and there are three per candidate -- with a workgroup per list the 7.6 M discarded candidates of a 20 M-fragment
// sample were 22.7 M workgroups x 256 lanes, more work-items than a launch takes (2^32): the launch failed without a word and discarded.tsv counted the discarded reads of its
// candidates from whatever the buffer held.  Found by the SHA-256 pin of discarded.tsv at 20 M fragments, round 5; the launches of this file are checked now.)
__global__ void list_copy_of_kernel(CandidateTable t, const uint32_t* candidates, const uint64_t* compact_offset, uint32_t* reads, uint64_t n_lists) {
	const uint32_t lane = threadIdx.x % 64;
	for (uint64_t k = (uint64_t) blockIdx.x * (BLOCK / 64) + threadIdx.x / 64; k < n_lists; k += (uint64_t) gridDim.x * (BLOCK / 64)) {
		if (k % 3 == 2 && t.discordant_before != nullptr) continue; // (an implicit list: list_expand_of_kernel)
		const uint32_t c = candidates[k / 3];
		const uint64_t at = 3 * (uint64_t) c + k % 3;
		const uint64_t begin = t.list_offset[at], target = compact_offset[k]; const uint32_t size = (uint32_t) (t.list_offset[at + 1] - begin);
		for (uint32_t e = lane; e < size; e += 64) reads[target + e] = k % 3 == 2 ? t.read_lists[begin + e] : split_list_entry(t, c, begin + e);
	}
}
// the implicit discordant lists of some candidates: one wavefront per candidate walks its bucket again
__global__ void __launch_bounds__(64) list_expand_of_kernel(AnnotationView ann, CandidateTable t, DiscordantBuckets buckets, BucketRanges ranges, int32_t max_mate_gap, uint32_t threshold, const uint32_t* candidates, const uint64_t* compact_offset, uint32_t* reads, uint32_t first_candidate) {
	const uint32_t k = first_candidate + blockIdx.x, lane = threadIdx.x;
	const uint32_t c = candidates[k];
	if (t.list_offset[3 * (uint64_t) c + 3] == t.list_offset[3 * (uint64_t) c + 2]) return;
	BucketRef ref; ref.candidate = c; ref.begin = ranges.begin[c]; ref.end = ranges.end[c];
	const uint32_t flags = t.flags[c];
	uint32_t unfiltered = 0, appended = 0, lane_votes = 0; int32_t lane_anchor1 = 0, lane_anchor2 = 0; bool zero_seen = false; AnchorFold fold1 = anchor_identity(), fold2 = anchor_identity();
	scan_bucket<3>(ann, buckets, ref, t.gene1[c], t.gene2[c], t.breakpoint1[c], t.breakpoint2[c], flags & CFLAG_UPSTREAM1, flags & CFLAG_UPSTREAM2, ranges.had_split_reads[c], max_mate_gap, threshold, lane,
	               reads + compact_offset[3 * (uint64_t) k + 2], nullptr, unfiltered, appended, lane_anchor1, lane_anchor2, lane_votes, zero_seen, fold1, fold2);
}

extern "C" int agpu_get_candidate_read_lists_of(agpu_ctx* ctx, const uint32_t* candidates, uint64_t n, uint64_t* list_offset, uint32_t* reads, uint64_t capacity, uint64_t* total) {
	if (!ctx || !ctx->fusions_done) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	if (!ctx->failed_launch.empty()) { set_last_error("a kernel of this sample was not launched (" + ctx->failed_launch + "): its results are not to be read"); return AGPU_ERR_DEVICE; }
	if (n > 0x3FFFFFFFull) { set_last_error("too many candidates"); return AGPU_ERR_INVALID; }
	for (uint64_t k = 0; k < n; ++k) if (candidates[k] >= ctx->n_candidates) { set_last_error("candidate index out of range"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	if (n == 0) { if (list_offset) list_offset[0] = 0; if (total) *total = 0; return AGPU_OK; }
	DeviceBuffer& ids = ctx->scratch("lists_of.ids"); DeviceBuffer& sizes = ctx->scratch("lists_of.sizes"); DeviceBuffer& offsets = ctx->scratch("lists_of.offsets"); DeviceBuffer& out = ctx->scratch("lists_of.reads"); DeviceBuffer& scratch = ctx->scratch("lists_of.rocprim");
	ALLOC(ids, n * 4); ALLOC(sizes, (3 * n + 1) * 4); ALLOC(offsets, (3 * n + 1) * 8);
	HIP_CHECK(hipMemcpyAsync(ids.ptr, candidates, n * 4, hipMemcpyHostToDevice, s));
	list_sizes_of_kernel<<<(unsigned int) ((3 * n + 1 + BLOCK - 1) / BLOCK), BLOCK, 0, s>>>(ctx->candidates, ids.as<uint32_t>(), (uint32_t) n, sizes.as<uint32_t>());
	size_t bytes = 0;
	HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, sizes.as<uint32_t>(), offsets.as<uint64_t>(), (uint64_t) 0, 3 * n + 1, rocprim::plus<uint64_t>(), s));
	if (bytes > scratch.capacity) ALLOC(scratch, bytes);
	HIP_CHECK(rocprim::exclusive_scan(scratch.ptr, bytes, sizes.as<uint32_t>(), offsets.as<uint64_t>(), (uint64_t) 0, 3 * n + 1, rocprim::plus<uint64_t>(), s));
	uint64_t entries = 0;
	HIP_CHECK(hipMemcpyAsync(&entries, offsets.as<uint64_t>() + 3 * n, 8, hipMemcpyDeviceToHost, s));
	HIP_CHECK(hipStreamSynchronize(s));
	if (total) *total = entries;
	if (list_offset) HIP_CHECK(hipMemcpy(list_offset, offsets.ptr, (3 * n + 1) * 8, hipMemcpyDeviceToHost));
	if (reads && entries > 0) {
		if (capacity < entries) { set_last_error("capacity too small for the read lists"); return AGPU_ERR_INVALID; }
		ALLOC(out, (size_t) entries * 4);
		list_copy_of_kernel<<<(unsigned int) std::min<uint64_t>((3 * n + BLOCK / 64 - 1) / (BLOCK / 64), 1u << 20), BLOCK, 0, s>>>(ctx->candidates, ids.as<uint32_t>(), offsets.as<uint64_t>(), out.as<uint32_t>(), 3 * n);
		HIP_CHECK(hipGetLastError());
		if (ctx->lists_implicit) {
			DiscordantBuckets buckets; BucketRanges ranges;
			const size_t Md1 = std::max<uint32_t>(ctx->lists_n_bucket_rows, 1);
			const int32_t* columns = ctx->scratch("fusions.bucket_columns").as<int32_t>();
			buckets.breakpoint1 = columns; buckets.breakpoint2 = columns + Md1; buckets.info = (const uint32_t*) (columns + 2 * Md1); buckets.read = buckets.info + Md1; buckets.anchor1 = (const int32_t*) (buckets.read + Md1); buckets.anchor2 = buckets.anchor1 + Md1;
			ranges.begin = ctx->scratch("lists.bucket_begin").as<uint32_t>(); ranges.end = ctx->scratch("lists.bucket_end").as<uint32_t>(); ranges.had_split_reads = ctx->scratch("lists.had_split_reads").as<uint8_t>();
			for_each_wave_chunk(n, [&](uint64_t first, uint64_t count) { list_expand_of_kernel<<<(unsigned int) count, 64, 0, s>>>(ctx->annotation, ctx->candidates, buckets, ranges, ctx->lists_max_mate_gap, ctx->params.subsampling_threshold, ids.as<uint32_t>(), offsets.as<uint64_t>(), out.as<uint32_t>(), (uint32_t) first); });
			HIP_CHECK(hipGetLastError());
		}
		HIP_CHECK(hipMemcpyAsync(reads, out.ptr, (size_t) entries * 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipStreamSynchronize(s));
	}
	return AGPU_OK;
}

extern "C" int agpu_get_fusion_stats(agpu_ctx* ctx, uint64_t* stats) {
	if (!ctx || !ctx->fusions_done || !stats) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	stats[0] = ctx->n_emissions; stats[1] = ctx->n_candidates; stats[2] = ctx->n_list_entries; stats[3] = ctx->n_discordant_emissions; stats[4] = ctx->n_queued_buckets;
	return AGPU_OK;
}

extern "C" int agpu_get_discordant_swapped(agpu_ctx* ctx, uint8_t* swapped) {
	if (!ctx || !ctx->fusions_done || !swapped) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	HIP_CHECK(hipStreamSynchronize(ctx->stream));
	if (ctx->n) HIP_CHECK(hipMemcpy(swapped, ctx->discordant_swapped.ptr, ctx->n, hipMemcpyDeviceToHost));
	return AGPU_OK;
}

// ---- sharded samples -----------------------------------------------------------------------------------------

extern "C" int agpu_build_emissions(agpu_ctx* ctx, uint32_t n_partitions, uint64_t* counts) {
	if (!ctx || !ctx->stage2_done || n_partitions == 0 || n_partitions > 4096) { set_last_error("agpu_read_filters_stage2 must run first; 1..4096 partitions"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	(void) hipEventRecord(ctx->event_start, s);
	uint32_t M = 0;
	int status = build_emissions(ctx, M);
	if (status != AGPU_OK) return status;
	std::vector<uint32_t> offsets(n_partitions + 1, 0);
	offsets[n_partitions] = M;
	if (n_partitions > 1 && M > 0) {
		DeviceBuffer& owners = ctx->scratch("shard.owners"); DeviceBuffer& sorted_owners = ctx->scratch("shard.sorted_owners"); DeviceBuffer& order = ctx->scratch("shard.order");
		DeviceBuffer& partitioned = ctx->scratch("shard.partitioned"); DeviceBuffer& device_offsets = ctx->scratch("shard.offsets"); DeviceBuffer& scratch = ctx->scratch("shard.rocprim");
		ALLOC(owners, (size_t) M * 4); ALLOC(sorted_owners, (size_t) M * 4); ALLOC(order, (size_t) M * 4); ALLOC(partitioned, (size_t) M * sizeof(FusionEmission)); ALLOC(device_offsets, ((size_t) n_partitions + 1) * 4);
		emission_owner_kernel<<<grid_for(M), BLOCK, 0, s>>>(M, ctx->emissions.as<FusionEmission>(), n_partitions, owners.as<uint32_t>());
		uint32_t bits = 1;
		while ((1u << bits) < n_partitions) ++bits;
		size_t bytes = 0; // the radix sort is stable: name order is kept inside every partition
		HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, owners.as<uint32_t>(), sorted_owners.as<uint32_t>(), rocprim::counting_iterator<uint32_t>(0), order.as<uint32_t>(), M, 0, bits, s));
		if (bytes > scratch.capacity) ALLOC(scratch, bytes);
		HIP_CHECK(rocprim::radix_sort_pairs(scratch.ptr, bytes, owners.as<uint32_t>(), sorted_owners.as<uint32_t>(), rocprim::counting_iterator<uint32_t>(0), order.as<uint32_t>(), M, 0, bits, s));
		emission_permute_kernel<<<grid_for(M), BLOCK, 0, s>>>(M, order.as<uint32_t>(), ctx->emissions.as<FusionEmission>(), partitioned.as<FusionEmission>());
		partition_offsets_kernel<<<grid_for((uint64_t) n_partitions + 1), BLOCK, 0, s>>>(sorted_owners.as<uint32_t>(), M, n_partitions, device_offsets.as<uint32_t>());
		HIP_CHECK(hipMemcpyAsync(offsets.data(), device_offsets.ptr, ((size_t) n_partitions + 1) * 4, hipMemcpyDeviceToHost, s));
		HIP_CHECK(hipMemcpyAsync(ctx->emissions.ptr, partitioned.ptr, (size_t) M * sizeof(FusionEmission), hipMemcpyDeviceToDevice, s));
	}
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	ctx->last_bytes = ctx->n * 60 + (uint64_t) M * 2 * sizeof(FusionEmission);
	if (counts) for (uint32_t p = 0; p < n_partitions; ++p) counts[p] = offsets[p + 1] - offsets[p];
	return AGPU_OK;
}

extern "C" int agpu_copy_emissions(agpu_ctx* ctx, void* destination) {
	if (!ctx || !destination) { set_last_error("null argument"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	if (ctx->n_emissions) HIP_CHECK(hipMemcpy(destination, ctx->emissions.ptr, (size_t) ctx->n_emissions * sizeof(FusionEmission), hipMemcpyDefault));
	return AGPU_OK;
}

extern "C" int agpu_find_fusions_from_emissions(agpu_ctx* ctx, const void* emissions, uint64_t n_emissions, int32_t max_mate_gap, uint64_t* n_candidates) {
	if (!ctx || !ctx->annotated || (!emissions && n_emissions)) { set_last_error("the context must be annotated (dummy genes of the whole sample) first"); return AGPU_ERR_INVALID; }
	if (n_emissions >= 0xFFFFFFF0ull) { set_last_error("too many emissions for one owner"); return AGPU_ERR_CAPACITY; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	DeviceBuffer& received = ctx->scratch("shard.received");
	ALLOC(received, (size_t) std::max<uint64_t>(n_emissions, 1) * sizeof(FusionEmission));
	if (n_emissions) HIP_CHECK(hipMemcpyAsync(received.ptr, emissions, (size_t) n_emissions * sizeof(FusionEmission), hipMemcpyDefault, s));
	(void) hipEventRecord(ctx->event_start, s);
	int status = candidates_from_emissions(ctx, received.as<FusionEmission>(), (uint32_t) n_emissions, ctx->global_n ? ctx->global_n : ctx->n, max_mate_gap, n_candidates);
	if (status != AGPU_OK) return status;
	HIP_CHECK(hipEventRecord(ctx->event_stop, s));
	HIP_CHECK(hipEventSynchronize(ctx->event_stop));
	HIP_CHECK(hipEventElapsedTime(&ctx->last_ms, ctx->event_start, ctx->event_stop));
	collect_kernel_samples(ctx);
	return AGPU_OK;
}

extern "C" int agpu_get_candidate_first_occurrence(agpu_ctx* ctx, uint64_t* first_occurrence) {
	if (!ctx || !ctx->fusions_done || !first_occurrence) { set_last_error("agpu_find_fusions must run first"); return AGPU_ERR_INVALID; }
	HIP_CHECK(hipSetDevice(ctx->device));
	HIP_CHECK(hipStreamSynchronize(ctx->stream));
	if (ctx->n_candidates) HIP_CHECK(hipMemcpy(first_occurrence, ctx->cand_first_occurrence.ptr, (size_t) ctx->n_candidates * 8, hipMemcpyDefault));
	return AGPU_OK;
}

extern "C" int agpu_import_candidates(agpu_ctx* ctx, uint64_t n_candidates, const uint32_t* gene1, const uint32_t* gene2, const uint32_t* contigs, const int32_t* breakpoint1, const int32_t* breakpoint2,
                                      const uint32_t* flags, const uint8_t* filter, const uint32_t* split_reads1, const uint32_t* split_reads2, const uint32_t* discordant_mates, const int32_t* anchor1, const int32_t* anchor2) {
	if (!ctx || !ctx->annotated) { set_last_error("the context must be annotated first"); return AGPU_ERR_INVALID; }
	if (n_candidates >= 0x7FFFFFF0ull) { set_last_error("too many candidates"); return AGPU_ERR_CAPACITY; }
	HIP_CHECK(hipSetDevice(ctx->device));
	hipStream_t s = ctx->stream;
	const size_t C = n_candidates, C1 = std::max<size_t>(C, 1);
	struct { DeviceBuffer* buffer; const void* source; size_t width; } columns[] = {
		{ &ctx->cand_gene1, gene1, 4 }, { &ctx->cand_gene2, gene2, 4 }, { &ctx->cand_contigs, contigs, 4 }, { &ctx->cand_breakpoint1, breakpoint1, 4 }, { &ctx->cand_breakpoint2, breakpoint2, 4 },
		{ &ctx->cand_flags, flags, 4 }, { &ctx->cand_filter, filter, 1 }, { &ctx->cand_split_reads1, split_reads1, 4 }, { &ctx->cand_split_reads2, split_reads2, 4 }, { &ctx->cand_discordant_mates, discordant_mates, 4 },
		{ &ctx->cand_anchor1, anchor1, 4 }, { &ctx->cand_anchor2, anchor2, 4 } };
	for (size_t k = 0; k < sizeof(columns) / sizeof(columns[0]); ++k) {
		if (!columns[k].source && C) { set_last_error("null column"); return AGPU_ERR_INVALID; }
		ALLOC(*columns[k].buffer, C1 * columns[k].width);
		if (C) HIP_CHECK(hipMemcpyAsync(columns[k].buffer->ptr, columns[k].source, C * columns[k].width, hipMemcpyDefault, s));
	}
	if (ctx->fusions_done && ctx->global_n != 0 && !ctx->candidates_imported) { // the read lists stay with the owners of the gene pairs
		ctx->owned_list_offset.swap(ctx->cand_list_offset); ctx->owned_read_lists.swap(ctx->cand_read_lists);
		ctx->n_owned = ctx->n_candidates; ctx->n_owned_list_entries = ctx->n_list_entries;
		ctx->owned_index_set = false; ctx->multimappers_begun = false;
	}
	ALLOC(ctx->cand_list_offset, (3 * C + 1) * 8); ALLOC(ctx->cand_read_lists, 16); ALLOC(ctx->cand_votes, C1 * 8);
	HIP_CHECK(hipMemsetAsync(ctx->cand_list_offset.ptr, 0, (3 * C + 1) * 8, s));
	CandidateTable& t = ctx->candidates;
	t.n = (uint32_t) C; t.gene1 = ctx->cand_gene1.as<uint32_t>(); t.gene2 = ctx->cand_gene2.as<uint32_t>(); t.contigs = ctx->cand_contigs.as<uint32_t>();
	t.breakpoint1 = ctx->cand_breakpoint1.as<int32_t>(); t.breakpoint2 = ctx->cand_breakpoint2.as<int32_t>(); t.flags = ctx->cand_flags.as<uint32_t>(); t.filter = ctx->cand_filter.as<uint8_t>();
	t.split_reads1 = ctx->cand_split_reads1.as<uint32_t>(); t.split_reads2 = ctx->cand_split_reads2.as<uint32_t>(); t.discordant_mates = ctx->cand_discordant_mates.as<uint32_t>();
	t.anchor1 = ctx->cand_anchor1.as<int32_t>(); t.anchor2 = ctx->cand_anchor2.as<int32_t>(); t.list_offset = ctx->cand_list_offset.as<uint64_t>(); t.read_lists = ctx->cand_read_lists.as<uint32_t>();
	t.votes = ctx->cand_votes.as<uint32_t>();
	HIP_CHECK(hipStreamSynchronize(s));
	ctx->n_candidates = (uint32_t) C; ctx->n_list_entries = 0;
	ctx->fusions_done = true; ctx->evalue_done = false; ctx->iteration_order_done = false; ctx->kmer_index_done = false; ctx->candidates_imported = true; ctx->genomic_support_marked = false; ctx->confidence_candidates = 0xFFFFFFFFu;
	return AGPU_OK;
}
