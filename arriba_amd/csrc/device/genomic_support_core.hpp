// arriba_amd/csrc/device/genomic_support_core.hpp -- structural variants from whole-genome sequencing (-d; reference: source/filter_genomic_support.cpp):
// mark_genomic_support (:81-219) finds for every candidate the closest pair of genomic breakpoints that can explain it, filter_no_genomic_support
// (:401-417) discards low-confidence candidates without one, recover_genomic_support (:419-444) brings back candidates with one that six filters
// discarded.  All three are per-candidate: one thread each.  The reference indexes the variants by (contig1, contig2, direction1, direction2) and
// first position; here they are one array sorted by (that key, first position, line of the file).
#ifndef AGPU_GENOMIC_SUPPORT_CORE_HPP
#define AGPU_GENOMIC_SUPPORT_CORE_HPP 1

#include "event_core.hpp"

namespace agpu {

const uint8_t FILTER_no_genomic_support = 29, FILTER_genomic_support = 34, FILTER_mismappers_id = 11; // source/common.hpp:29-67

struct GenomicBreakpoints { const uint64_t* keys; const int32_t* position1; const int32_t* position2; uint32_t n; }; // sorted by (key, position1, file order)
AGPU_HD uint64_t genomic_breakpoint_key(uint32_t contig1, uint32_t contig2, bool upstream1, bool upstream2) { return (uint64_t) contig1 << 34 | (uint64_t) contig2 << 2 | (upstream1 ? 1u : 0u) | (upstream2 ? 2u : 0u); }

// reference: is_genomic_breakpoint_close_enough (:62-79)
AGPU_HD bool genomic_breakpoint_is_close(const AnnotationView& ann, bool upstream, int32_t genomic_breakpoint, int32_t fusion_breakpoint, uint32_t gene, int32_t max_distance) {
	const bool dummy = ann.gene_bits[gene] & GBIT_DUMMY;
	if (upstream) return genomic_breakpoint >= (dummy ? fusion_breakpoint : ann.gene_start[gene]) - max_distance && genomic_breakpoint <= fusion_breakpoint + 5;
	return genomic_breakpoint <= (dummy ? fusion_breakpoint : ann.gene_end[gene]) + max_distance && genomic_breakpoint >= fusion_breakpoint - 5;
}

// reference: the loop over the candidates of mark_genomic_support (:168-212).  The walk goes towards smaller first positions whatever the direction
// (so a downstream breakpoint only ever looks at the first variant at or behind breakpoint - 5, as in the reference).
AGPU_HD void closest_genomic_breakpoints(const AnnotationView& ann, const CandidateTable& t, const GenomicBreakpoints& g, uint32_t c, int32_t max_distance, uint32_t max_itd_length, int32_t& closest1, int32_t& closest2) {
	closest1 = closest2 = -1;
	const uint32_t flags = t.flags[c];
	const bool upstream1 = flags & CFLAG_UPSTREAM1, upstream2 = flags & CFLAG_UPSTREAM2;
	const uint32_t contig1 = t.contigs[c] >> 16, contig2 = t.contigs[c] & 0xFFFF;
	const int32_t breakpoint1 = t.breakpoint1[c], breakpoint2 = t.breakpoint2[c];
	const uint64_t key = genomic_breakpoint_key(contig1, contig2, upstream1, upstream2);
	const uint32_t run_begin = lower_bound_u64(g.keys, g.n, key);
	uint32_t run_end = run_begin;
	{ uint32_t lo = run_begin, hi = g.n; while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (g.keys[mid] <= key) lo = mid + 1; else hi = mid; } run_end = lo; }
	if (run_begin == run_end) return;
	// lower_bound over the first positions of the run
	const int32_t wanted = breakpoint1 + (upstream1 ? +5 : -5); // +/-5: some flexibility of the alignment
	uint32_t at = run_begin;
	{ uint32_t lo = run_begin, hi = run_end; while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (g.position1[mid] < wanted) lo = mid + 1; else hi = mid; } at = lo; }
	if (upstream1) { if (at == run_begin) return; --at; } // the variant upstream of the breakpoint
	else if (at == run_end) return;
	// `at` is inside a group of equal first positions (one map entry of the reference): start at the first of the group
	while (at > run_begin && g.position1[at - 1] == g.position1[at]) --at;
	const bool is_itd = t.gene1[c] == t.gene2[c] && (uint32_t) breakpoint2 - (uint32_t) breakpoint1 < max_itd_length && upstream1 && !upstream2; // source/common.hpp:270-274
	while (true) {
		const int32_t genomic1 = g.position1[at];
		if (!genomic_breakpoint_is_close(ann, upstream1, genomic1, breakpoint1, t.gene1[c], max_distance)) break;
		uint32_t group_end = at;
		while (group_end < run_end && g.position1[group_end] == genomic1) ++group_end;
		for (uint32_t k = at; k < group_end; ++k) { // second positions in the order of the file
			const int32_t genomic2 = g.position2[k];
			if (!genomic_breakpoint_is_close(ann, upstream2, genomic2, breakpoint2, t.gene2[c], max_distance)) continue;
			const bool plausible = contig1 != contig2 ||
				(upstream1 && !upstream2 && (!is_itd || (breakpoint1 - genomic1 < (int32_t) max_itd_length && genomic2 - breakpoint2 < (int32_t) max_itd_length))) || // duplications; ITDs: not farther than their length
				(!upstream1 && upstream2 && genomic1 < breakpoint2 && genomic2 > breakpoint1) ||  // deletions: both genomic breakpoints between the transcriptomic ones
				(upstream1 && upstream2 && genomic2 > breakpoint1) ||                            // inversions: one of them in between
				(!upstream1 && !upstream2 && genomic1 < breakpoint2);
			if (!plausible) continue;
			int32_t new1 = genomic1 - breakpoint1, new2 = breakpoint2 - genomic2; if (new1 < 0) new1 = -new1; if (new2 < 0) new2 = -new2;
			int32_t old1 = breakpoint1 - closest1, old2 = breakpoint2 - closest2; if (old1 < 0) old1 = -old1; if (old2 < 0) old2 = -old2;
			if (closest1 < 0 || closest2 < 0 || old1 + old2 > new1 + new2) { closest1 = genomic1; closest2 = genomic2; }
		}
		if (at == run_begin) break; // the previous map entry
		--at;
		while (at > run_begin && g.position1[at - 1] == g.position1[at]) --at;
	}
}

// reference: filter_no_genomic_support (:401-417), recover_genomic_support (:419-444)
AGPU_HD bool lacks_genomic_support(const GenomeView& genome, const CandidateTable& t, const GenomicSupport& wgs, const uint8_t* confidence, uint32_t c) {
	return !wgs.has(c) && confidence[c] == CONFIDENCE_LOW && !contig_is_viral(genome, t.contigs[c] >> 16) && !contig_is_viral(genome, t.contigs[c] & 0xFFFF);
}
AGPU_HD bool recovered_by_genomic_support(const CandidateTable& t, const GenomicSupport& wgs, uint32_t c) {
	const uint8_t filter = t.filter[c];
	return wgs.has(c) && (filter == FILTER_end_to_end || filter == FILTER_intronic || filter == FILTER_mismappers_id || filter == FILTER_no_coverage || filter == FILTER_in_vitro || filter == FILTER_relative_support);
}

}

#endif
