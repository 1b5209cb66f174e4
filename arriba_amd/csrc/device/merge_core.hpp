// arriba_amd/csrc/device/merge_core.hpp -- merge_adjacent_fusions (reference: source/merge_adjacent_fusions.cpp:19-108).
//
// The reference sorts the unfiltered candidates by coordinate and sweeps them sequentially: a candidate absorbs the split reads of the
// candidates of the same gene pair and directions whose breakpoints are shifted by at most 5 bp along the same diagonal, if it has the
// most support; the absorbed ones get the filter `merge_adjacent`.  The sweep mutates the counters it later compares, so its order matters
// -- but only inside a CLUSTER: candidates of one gene pair and direction pair ON ONE DIAGONAL (breakpoint1 + breakpoint2 constant when the
// directions are equal, breakpoint2 - breakpoint1 constant otherwise) whose first breakpoints form a chain with gaps <= 5 bp.  Only
// internal tandem duplications are merged across diagonals, so the groups that can hold them (gene1 == gene2, upstream/downstream) keep
// all their diagonals in one cluster.  Clusters are independent and almost always tiny (a dense hot spot of 20 000 candidates within a
// few hundred bp falls apart into diagonals of at most ~100), so the device sorts the candidates by (gene pair + directions, diagonal,
// breakpoint1, breakpoint2), cuts the clusters and lets one thread run the reference's sweep over each cluster.
#ifndef AGPU_MERGE_CORE_HPP
#define AGPU_MERGE_CORE_HPP 1

#include "fusion_core.hpp"

namespace agpu {

const uint8_t FILTER_merge_adjacent = 23; // source/common.hpp:29-67

AGPU_HD bool candidate_is_internal_tandem_duplication(const CandidateTable& t, uint32_t c, uint32_t max_itd_length) { // source/common.hpp:270-274
	return t.gene1[c] == t.gene2[c] && (uint32_t) (t.breakpoint2[c] - t.breakpoint1[c]) < max_itd_length && (t.flags[c] & CFLAG_UPSTREAM1) && !(t.flags[c] & CFLAG_UPSTREAM2);
}
// candidates the reference puts into sorted_fusions (:24-27)
AGPU_HD bool takes_part_in_merge(const CandidateTable& t, uint32_t c, uint32_t max_itd_length) {
	return t.filter[c] == FILTER_none || candidate_is_internal_tandem_duplication(t, c, max_itd_length);
}
AGPU_HD uint32_t merge_supporting_reads(const CandidateTable& t, uint32_t c) { return t.split_reads1[c] + t.split_reads2[c] + t.discordant_mates[c]; }
// size of split_read1_list + split_read2_list; extra_split_list = entries appended by earlier ITD merges (sizes only)
AGPU_HD uint32_t merge_split_list_size(const CandidateTable& t, const uint32_t* extra_split_list, uint32_t c) {
	return (uint32_t) (t.list_offset[3 * (uint64_t) c + 2] - t.list_offset[3 * (uint64_t) c]) + extra_split_list[c];
}

// sort keys: candidates that do not take part sort behind all others
AGPU_HD uint64_t merge_group_key(const CandidateTable& t, uint32_t c, uint32_t max_itd_length) {
	const uint32_t flags = t.flags[c];
	return takes_part_in_merge(t, c, max_itd_length) ? (uint64_t) t.gene1[c] << 33 | (uint64_t) t.gene2[c] << 2 | ((flags & CFLAG_UPSTREAM1) ? 1u : 0u) | ((flags & CFLAG_UPSTREAM2) ? 2u : 0u) : ~0ull;
}
const int MERGE_DIAGONAL_KEY_BITS = 34;
AGPU_HD uint64_t merge_diagonal_key(const CandidateTable& t, uint32_t c) {
	const uint32_t flags = t.flags[c];
	const bool upstream1 = (flags & CFLAG_UPSTREAM1) != 0, upstream2 = (flags & CFLAG_UPSTREAM2) != 0;
	if (t.gene1[c] == t.gene2[c] && upstream1 && !upstream2) return 0; // this group may hold internal tandem duplications: one cluster for all diagonals
	const int64_t breakpoint1 = t.breakpoint1[c], breakpoint2 = t.breakpoint2[c];
	return 1 + (uint64_t) (upstream1 == upstream2 ? breakpoint1 + breakpoint2 : breakpoint2 - breakpoint1 + (int64_t(1) << 31));
}
// do the candidates at the sorted positions j-1 and j belong to one cluster?
AGPU_HD bool merge_cluster_continues(const CandidateTable& t, const uint32_t* order, uint32_t j, int32_t max_distance, uint32_t max_itd_length) {
	const uint32_t previous = order[j - 1], current = order[j];
	return merge_group_key(t, previous, max_itd_length) == merge_group_key(t, current, max_itd_length) && merge_diagonal_key(t, previous) == merge_diagonal_key(t, current) &&
	       t.breakpoint1[current] - t.breakpoint1[previous] <= max_distance;
}

// When an internal tandem duplication absorbs another one, the reference appends the split-read lists of the absorbed candidate to its own
// (source/merge_adjacent_fusions.cpp:99-103): later stages walk these reads (filter_multimappers, recover_internal_tandem_duplication, ...).
// The appended entries are collected here -- per candidate and split list a region of `pool` holding everything appended so far, in the
// order of the reference -- and merged into the read lists after the sweep (agpu_merge.hip: rebuild).  A candidate lives in one cluster and a
// cluster is swept by one thread, so only the pool cursor is shared.
struct ItdAppended {
	uint32_t* pool; uint32_t capacity; uint32_t* cursor;
	uint32_t* begin;   // [2 * n_candidates] region of split_read1_list / split_read2_list
	uint32_t* length;  // [2 * n_candidates]
	uint32_t* overflow;
};
AGPU_HD void itd_append_lists(const CandidateTable& t, const ItdAppended& appended, uint32_t fusion, uint32_t other) {
	for (uint32_t list = 0; list < 2; ++list) {
		const uint64_t own_begin = t.list_offset[3 * (uint64_t) other + list]; const uint32_t own_length = (uint32_t) (t.list_offset[3 * (uint64_t) other + list + 1] - own_begin);
		const uint32_t other_length = appended.length[2 * (uint64_t) other + list], old_length = appended.length[2 * (uint64_t) fusion + list];
		if (own_length + other_length == 0) continue;
		const uint32_t new_length = old_length + own_length + other_length;
		const uint32_t at = atomic_add_u32(appended.cursor, new_length);
		if ((uint64_t) at + new_length > appended.capacity) { *appended.overflow = 1; continue; }
		uint32_t* out = appended.pool + at;
		const uint32_t* old = appended.pool + appended.begin[2 * (uint64_t) fusion + list];
		for (uint32_t j = 0; j < old_length; ++j) out[j] = old[j];
		for (uint32_t j = 0; j < own_length; ++j) out[old_length + j] = split_list_entry(t, other, own_begin + j);
		const uint32_t* theirs = appended.pool + appended.begin[2 * (uint64_t) other + list];
		for (uint32_t j = 0; j < other_length; ++j) out[old_length + own_length + j] = theirs[j];
		appended.begin[2 * (uint64_t) fusion + list] = at; appended.length[2 * (uint64_t) fusion + list] = new_length;
	}
}

// The reference's sweep over the cluster order[begin .. end) (ascending breakpoint1, then breakpoint2).
AGPU_HD void merge_cluster(const CandidateTable& t, const uint32_t* order, uint32_t begin, uint32_t end, int32_t max_distance, uint32_t max_itd_length, uint32_t* extra_split_list, const ItdAppended& appended) {
	for (uint32_t e = begin; e < end; ++e) {
		const uint32_t fusion = order[e];
		const bool is_itd = candidate_is_internal_tandem_duplication(t, fusion, max_itd_length);
		if ((!is_itd && t.split_reads1[fusion] + t.split_reads2[fusion] == 0) || (is_itd && merge_split_list_size(t, extra_split_list, fusion) == 0))
			continue; // only merge fusions with exactly known breakpoints
		const int32_t breakpoint1 = t.breakpoint1[fusion], breakpoint2 = t.breakpoint2[fusion];
		const bool same_direction = ((t.flags[fusion] & CFLAG_UPSTREAM1) != 0) == ((t.flags[fusion] & CFLAG_UPSTREAM2) != 0);
		// two passes over the adjacent candidates in the reference's order (upstream ones nearest first, then downstream ones):
		// pass 0 decides whether `fusion` has the most support and sums the split reads, pass 1 marks the adjacent ones
		uint32_t sum_split_reads1 = 0, sum_split_reads2 = 0, sum_split_lists = 0;
		bool fusion_has_most_support = true;
		for (int pass = 0; pass < 2 && fusion_has_most_support; ++pass) {
			for (int side = 0; side < 2; ++side) {
				// side 0: e-1, e-2, ... while breakpoint1 >= breakpoint1(fusion) - max_distance; side 1: e+1, ... while <= + max_distance
				for (uint32_t k = 1; side == 0 ? e >= begin + k : e + k < end; ++k) {
					const uint32_t other = order[side == 0 ? e - k : e + k];
					const int32_t other1 = t.breakpoint1[other], other2 = t.breakpoint2[other];
					if (side == 0 ? other1 < breakpoint1 - max_distance : other1 > breakpoint1 + max_distance) break;
					const int32_t expected2 = side == 0 ? breakpoint2 + (breakpoint1 - other1) * (same_direction ? +1 : -1)
					                                    : breakpoint2 + (other1 - breakpoint1) * (same_direction ? -1 : +1); // shifted along the same diagonal
					int32_t distance2 = breakpoint2 - other2; if (distance2 < 0) distance2 = -distance2;
					if (!(other2 == expected2 || (is_itd && distance2 <= max_distance))) continue;
					if (!(t.split_reads1[other] + t.split_reads2[other] > 0 || (is_itd && merge_split_list_size(t, extra_split_list, other) > 0))) continue;
					if (pass == 0) {
						const uint32_t mine = merge_supporting_reads(t, fusion), theirs = merge_supporting_reads(t, other);
						if (mine < theirs || (mine == theirs && merge_split_list_size(t, extra_split_list, fusion) < merge_split_list_size(t, extra_split_list, other))) { fusion_has_most_support = false; break; }
						sum_split_reads1 += t.split_reads1[other]; sum_split_reads2 += t.split_reads2[other];
						sum_split_lists += merge_split_list_size(t, extra_split_list, other);
					} else {
						t.filter[other] = FILTER_merge_adjacent;
						if (is_itd) itd_append_lists(t, appended, fusion, other);
					}
				}
				if (!fusion_has_most_support) break;
			}
			if (pass == 0 && fusion_has_most_support) {
				t.split_reads1[fusion] = (t.split_reads1[fusion] + sum_split_reads1) & 0x7FFFu; // 15-bit counters in the reference (hazard H10)
				t.split_reads2[fusion] = (t.split_reads2[fusion] + sum_split_reads2) & 0x7FFFu;
				if (is_itd) extra_split_list[fusion] += sum_split_lists; // the reference appends the lists of the absorbed candidates
			}
		}
	}
}

}

#endif
