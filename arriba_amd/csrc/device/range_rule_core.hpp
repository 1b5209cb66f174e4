// arriba_amd/csrc/device/range_rule_core.hpp -- filter_blacklisted_ranges (reference: source/filter_blacklisted_ranges.cpp:136-301, called at
// source/arriba.cpp:527-530) and recover_known_fusions (source/recover_known_fusions.cpp:14-100, called at source/arriba.cpp:475-478).
//
// Both files hold lines of two items (a gene, a position, a range, or -- second column of the blacklist -- a keyword); a candidate is discarded
// (blacklist) or recovered (known fusions) when at least one line matches it.  The reference looks lines and candidates up through 100 kb genome
// bins; a line is only compared with a candidate when they share a bin, which is part of the semantics and is reproduced here: the rules are
// indexed by bin (CSR, built on the host: a file is ~10^5 lines), one thread per candidate walks the bins of its breakpoints and genes.
// The verdict of a candidate does not depend on the other candidates: no ordering hazard.
#ifndef AGPU_RANGE_RULE_CORE_HPP
#define AGPU_RANGE_RULE_CORE_HPP 1

#include "event_core.hpp"
#include "../../../include/arriba_gpu.h"

namespace agpu {

const uint8_t FILTER_known_fusions = 18, FILTER_min_support = 17; // source/common.hpp:29-67 (FILTER_blacklist = 20: event_core.hpp)
const int32_t GENOME_BIN_SIZE = 100000; // source/filter_blacklisted_ranges.cpp:222

struct RangeRuleIndex {
	const agpu_range_rule* rules; uint32_t n_rules;
	const uint64_t* bin_keys;    // [n_bins] contig << 32 | (uint32_t) bin position, ascending
	const uint32_t* bin_offset;  // [n_bins + 1] into bin_rules
	const uint32_t* bin_rules;   // rule indices
	uint32_t n_bins;
};
AGPU_HD uint64_t genome_bin_key(uint32_t contig, int32_t bin) { return (uint64_t) contig << 32 | (uint32_t) (bin * GENOME_BIN_SIZE); }
// reference: get_genome_bins_from_range (:221-225): start / 100000 .. ceil(end / 100000) inclusive, integer division as in C++
AGPU_HD int32_t genome_bin_first(int32_t start) { return start / GENOME_BIN_SIZE; }
AGPU_HD int32_t genome_bin_last(int32_t end) { return (end + GENOME_BIN_SIZE - 1) / GENOME_BIN_SIZE; }

// reference: overlapping_fraction (:121-133): the fraction of range 1 that overlaps range 2 (a double expression returned as float)
AGPU_HD float overlapping_fraction(int32_t start1, int32_t end1, int32_t start2, int32_t end2) {
	AGPU_FP_AS_WRITTEN
	if (start1 >= start2 && end1 <= end2) return 1;
	if (start1 < start2 && end1 > end2) return (float) (1.0 * (end2 - start2) / (end1 - start1 + 1));
	if (start1 >= start2 && start1 <= end2) return (float) (1.0 * (end2 - start1) / (end1 - start1 + 1));
	if (end1 >= start2 && end1 <= end2) return (float) (1.0 * (end1 - start2) / (end1 - start1 + 1));
	return 0;
}

// reference: matches_blacklist_item (:136-218); which_breakpoint 1 or 2
AGPU_HD bool matches_range_item(const AnnotationView& ann, const CandidateTable& t, const float* evalues, uint32_t c, const agpu_range_item& item, int which_breakpoint, int32_t max_mate_gap, float evalue_cutoff) {
	AGPU_FP_AS_WRITTEN
	const uint32_t flags = t.flags[c], split_reads1 = t.split_reads1[c], split_reads2 = t.split_reads2[c], discordant_mates = t.discordant_mates[c];
	const bool spliced1 = flags & CFLAG_SPLICED1, spliced2 = flags & CFLAG_SPLICED2;
	switch (item.type) {
		case AGPU_RULE_ANY: return true;
		case AGPU_RULE_SPLIT_READ_DONOR: return (which_breakpoint == 1 && discordant_mates + split_reads1 == 0) || (which_breakpoint == 2 && discordant_mates + split_reads2 == 0);
		case AGPU_RULE_SPLIT_READ_ACCEPTOR: return (which_breakpoint == 1 && discordant_mates + split_reads2 == 0) || (which_breakpoint == 2 && discordant_mates + split_reads1 == 0);
		case AGPU_RULE_SPLIT_READ_ANY: return discordant_mates == 0;
		case AGPU_RULE_DISCORDANT_MATES: return split_reads1 + split_reads2 == 0;
		case AGPU_RULE_READ_THROUGH: return candidate_is_read_through(t, c);
		case AGPU_RULE_LOW_SUPPORT: return evalues[c] > evalue_cutoff;
		case AGPU_RULE_FILTER_SPLICED: return evalues[c] > evalue_cutoff && spliced1 && spliced2;
		case AGPU_RULE_NOT_BOTH_SPLICED: return !spliced1 || !spliced2;
		case AGPU_RULE_GENE: return (which_breakpoint == 1 ? t.gene1[c] : t.gene2[c]) == item.gene;
		case AGPU_RULE_POSITION:
		case AGPU_RULE_RANGE: {
			const uint32_t contig = which_breakpoint == 1 ? t.contigs[c] >> 16 : t.contigs[c] & 0xFFFF;
			if (contig != item.contig) return false;
			if (item.strand_defined && !(flags & CFLAG_PREDICTED_STRANDS_AMBIGUOUS)) { // a match is assumed if the strands could not be predicted
				const bool forward = flags & (which_breakpoint == 1 ? CFLAG_PREDICTED_STRAND1 : CFLAG_PREDICTED_STRAND2);
				if (forward != (item.strand != 0)) return false;
			}
			if (item.type == AGPU_RULE_RANGE) { // the gene of the breakpoint must overlap the range by more than half
				const uint32_t gene = which_breakpoint == 1 ? t.gene1[c] : t.gene2[c];
				return overlapping_fraction(ann.gene_start[gene], ann.gene_end[gene], item.start, item.end) > 0.5;
			}
			const int32_t breakpoint = which_breakpoint == 1 ? t.breakpoint1[c] : t.breakpoint2[c];
			if (breakpoint == item.start) return true;
			if (split_reads1 + split_reads2 == 0) { // only discordant mates: near the position and pointing towards it
				const bool upstream = flags & (which_breakpoint == 1 ? CFLAG_UPSTREAM1 : CFLAG_UPSTREAM2);
				if ((!upstream && breakpoint <= item.start && breakpoint >= item.start - max_mate_gap) || (upstream && breakpoint >= item.start && breakpoint <= item.start + max_mate_gap)) return true;
			}
			return false;
		}
	}
	return false;
}

// blacklist: does the line match in one of the two assignments of its columns to the breakpoints? (:280-283)
AGPU_HD bool blacklist_rule_matches(const AnnotationView& ann, const CandidateTable& t, const float* evalues, uint32_t c, const agpu_range_rule& rule, int32_t max_mate_gap, float evalue_cutoff) {
	return (matches_range_item(ann, t, evalues, c, rule.first, 1, max_mate_gap, evalue_cutoff) && matches_range_item(ann, t, evalues, c, rule.second, 2, max_mate_gap, evalue_cutoff)) ||
	       (matches_range_item(ann, t, evalues, c, rule.first, 2, max_mate_gap, evalue_cutoff) && matches_range_item(ann, t, evalues, c, rule.second, 1, max_mate_gap, evalue_cutoff));
}
// known fusions: the 5' gene must match the first column, the 3' gene the second; and the recovery conditions (:66-93)
AGPU_HD bool known_fusion_rule_recovers(const AnnotationView& ann, const CoverageView& coverage, const CandidateTable& t, const float* evalues, uint32_t c, const agpu_range_rule& rule, int32_t max_mate_gap) {
	const uint32_t flags = t.flags[c];
	const int gene_5 = (flags & CFLAG_TRANSCRIPT_START_GENE1) ? 1 : 2, gene_3 = 3 - gene_5;
	const bool same_contig = (t.contigs[c] >> 16) == (t.contigs[c] & 0xFFFF);
	int32_t distance = t.breakpoint2[c] - t.breakpoint1[c]; if (distance < 0) distance = -distance;
	bool match_found = matches_range_item(ann, t, evalues, c, rule.first, gene_5, max_mate_gap, 0) && matches_range_item(ann, t, evalues, c, rule.second, gene_3, max_mate_gap, 0);
	if (!match_found && (flags & CFLAG_TRANSCRIPT_START_AMBIGUOUS) && !(same_contig && distance < 1000000)) // unreliable transcript start: swapped genes match, too, unless the breakpoints are close
		match_found = matches_range_item(ann, t, evalues, c, rule.first, gene_3, max_mate_gap, 0) && matches_range_item(ann, t, evalues, c, rule.second, gene_5, max_mate_gap, 0);
	if (!match_found) return false;
	if (rule.first.type == AGPU_RULE_POSITION && rule.second.type == AGPU_RULE_POSITION) return true; // two exact breakpoints: always rescued
	if (t.split_reads1[c] + t.split_reads2[c] + t.discordant_mates[c] >= 2) return true;                // otherwise two reads, or there are too many false positives
	return both_breakpoints_spliced(ann, t, c) &&                                                        // unless the breakpoints are at splice sites
	       coverage_near(coverage, t.contigs[c] >> 16, t.breakpoint1[c], !(flags & CFLAG_UPSTREAM1)) + coverage_near(coverage, t.contigs[c] & 0xFFFF, t.breakpoint2[c], !(flags & CFLAG_UPSTREAM2)) < 200 &&
	       (!same_contig || distance > 1000000);
}

// which candidates are looked at (blacklist :229-230 without genomic support; known fusions :43-50)
AGPU_HD bool blacklist_considers(const CandidateTable& t, uint32_t c, const GenomicSupport& wgs = GenomicSupport{ nullptr, nullptr }) { return t.filter[c] == FILTER_none || wgs.has(c); } // a filtered candidate with genomic support may be recovered later
AGPU_HD bool known_fusions_considers(const CandidateTable& t, uint32_t c) { return t.gene1[c] != t.gene2[c] && (t.filter[c] == FILTER_relative_support || t.filter[c] == FILTER_min_support); }

// walks the rules in the genome bins of the candidate (breakpoint1, breakpoint2, gene1, gene2; :235-240, :53-57); mode 0 = blacklist, 1 = known fusions
AGPU_HD bool candidate_matches_any_rule(const AnnotationView& ann, const CoverageView& coverage, const CandidateTable& t, const float* evalues, uint32_t c, const RangeRuleIndex& index, int mode, int32_t max_mate_gap, float evalue_cutoff) {
	const uint32_t gene1 = t.gene1[c], gene2 = t.gene2[c];
	const uint32_t contig_of[4] = { t.contigs[c] >> 16, t.contigs[c] & 0xFFFF, t.contigs[c] >> 16, t.contigs[c] & 0xFFFF };
	const int32_t start_of[4] = { t.breakpoint1[c], t.breakpoint2[c], ann.gene_start[gene1], ann.gene_start[gene2] }, end_of[4] = { t.breakpoint1[c], t.breakpoint2[c], ann.gene_end[gene1], ann.gene_end[gene2] };
	for (int range = 0; range < 4; ++range)
		for (int32_t bin = genome_bin_first(start_of[range]); bin <= genome_bin_last(end_of[range]); ++bin) {
			const uint64_t key = genome_bin_key(contig_of[range], bin);
			const uint32_t at = lower_bound_u64(index.bin_keys, index.n_bins, key);
			if (at >= index.n_bins || index.bin_keys[at] != key) continue;
			for (uint32_t k = index.bin_offset[at]; k < index.bin_offset[at + 1]; ++k) {
				const agpu_range_rule& rule = index.rules[index.bin_rules[k]];
				if (mode == 0 ? blacklist_rule_matches(ann, t, evalues, c, rule, max_mate_gap, evalue_cutoff) : known_fusion_rule_recovers(ann, coverage, t, evalues, c, rule, max_mate_gap)) return true;
			}
		}
	return false;
}

}

#endif
