// arriba_amd/csrc/device/fusion_core.hpp -- candidate building (reference: find_fusions, source/fusions.cpp:203-473) on the device.
//
// The reference walks the reads in name order and updates a hash map of candidates keyed by
// (gene1, gene2, contig1, contig2, breakpoint1, breakpoint2, direction1, direction2).  Here every read EMITS one record per
// gene1 x gene2 pair; records of one key are brought together in name order (hash-assigned representative + one 64-bit radix
// sort), and the order-dependent update rules of the reference become prefix counts and associative reductions:
//
//   filter      none if any supporting read is unfiltered, else the filter of the first read that is not `duplicates`, else
//               duplicates (source/fusions.cpp:262-263)
//   read lists  a split read joins its list iff it is among the first T reads of its side or it is unfiltered and fewer than T
//               unfiltered reads precede it (T = subsampling threshold; derived from source/fusions.cpp:265-296)
//   anchors     running min/max with "0 means unset" over the reads that joined a list and over all discordant mates
//               (source/fusions.cpp:275-285, 347-357), expressed as an associative function composition
//   discordant  a wave scans the discordant mates of the candidate's gene pair in name order (source/fusions.cpp:367-437)
#ifndef AGPU_FUSION_CORE_HPP
#define AGPU_FUSION_CORE_HPP 1

#include "filter_core.hpp"

namespace agpu {

// info bits of an emission / candidate
// EINFO_SWAPPED: split read whose ends were exchanged to order the breakpoints; EINFO_MATES_SWAPPED: the same for a discordant
// fragment (find_fusions swaps MATE1/MATE2 of such a fragment in place when it attaches it to a candidate, source/fusions.cpp:414-421)
// EINFO_PREDICTED1/2, EINFO_AMBIGUOUS1: predicted strand of the alignment on side 1 / 2 of the (ordered) breakpoint pair -- what the strand
// vote of predict_fusion_strands (source/fusions.cpp:15-91) looks at, carried along so that the vote needs no second look at the read
enum : uint32_t { EINFO_UPSTREAM1 = 1, EINFO_UPSTREAM2 = 2, EINFO_SWAPPED = 4, EINFO_EXONIC1 = 8, EINFO_EXONIC2 = 16, EINFO_SPLIT = 32, EINFO_MATES_SWAPPED = 64, EINFO_FILTER_SHIFT = 8,
                  EINFO_PREDICTED1 = 1u << 16, EINFO_AMBIGUOUS1 = 1u << 17, EINFO_PREDICTED2 = 1u << 18,
                  EINFO_ORDINAL_SHIFT = 19 /* 8 bits: position of the emission among its read's gene1 x gene2 emissions */ };

struct FusionEmission {
	uint32_t gene1, gene2;
	int32_t breakpoint1, breakpoint2;
	uint32_t contigs;          // contig1 << 16 | contig2
	uint32_t info;             // EINFO_* | read filter << 8
	int32_t anchor1, anchor2;
	uint32_t read;             // name rank of the supporting fragment
};

AGPU_HD bool same_candidate(const FusionEmission& a, const FusionEmission& b) {
	return a.gene1 == b.gene1 && a.gene2 == b.gene2 && a.breakpoint1 == b.breakpoint1 && a.breakpoint2 == b.breakpoint2 && a.contigs == b.contigs && ((a.info ^ b.info) & 3u) == 0;
}
AGPU_HD uint64_t hash_candidate(const FusionEmission& e) {
	uint64_t h = ((uint64_t) e.gene1 << 32 | e.gene2) * 0x9E3779B97F4A7C15ULL;
	h ^= ((uint64_t) (uint32_t) e.breakpoint1 << 32 | (uint32_t) e.breakpoint2) * 0xC2B2AE3D27D4EB4FULL;
	h ^= ((uint64_t) e.contigs << 2 | (e.info & 3u)) * 0x165667B19E3779F9ULL;
	h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
	return h;
}
// discordant mates are bucketed by (gene1, gene2, direction1, direction2) (source/fusions.cpp:361)
AGPU_HD bool same_gene_pair(const FusionEmission& a, const FusionEmission& b) { return a.gene1 == b.gene1 && a.gene2 == b.gene2 && ((a.info ^ b.info) & 3u) == 0; }
AGPU_HD uint64_t hash_gene_pair(const FusionEmission& e) {
	uint64_t h = (((uint64_t) e.gene1 << 32 | e.gene2) ^ ((uint64_t) (e.info & 3u) << 62)) * 0x9E3779B97F4A7C15ULL;
	h ^= h >> 31; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
	return h;
}

// breakpoints, directions and anchors of the two ends of a fragment (reference: source/fusions.cpp:218-245, 302-327)
struct FragmentEnds {
	uint32_t contig1, contig2;
	int32_t breakpoint1, breakpoint2;
	int32_t anchor1, anchor2;
	bool upstream1, upstream2, exonic1, exonic2, swapped, is_split;
	int slot1, slot2; // alignment slots whose gene sets form genes1 / genes2
};

AGPU_HD void fragment_ends(const BatchView& b, uint64_t i, FragmentEnds& f) {
	f.is_split = b.n_aln[i] == 3;
	if (f.is_split) {
		uint8_t split = b.abits[SPLIT_READ][i], supp = b.abits[SUPPLEMENTARY][i], mate1 = b.abits[MATE1][i];
		f.contig1 = b.contig[SPLIT_READ][i]; f.contig2 = b.contig[SUPPLEMENTARY][i];
		f.breakpoint1 = (split & ABIT_STRAND) ? b.start[SPLIT_READ][i] : b.end[SPLIT_READ][i];
		f.breakpoint2 = (supp & ABIT_STRAND) ? b.end[SUPPLEMENTARY][i] : b.start[SUPPLEMENTARY][i];
		f.upstream1 = (split & ABIT_STRAND) != 0;
		f.upstream2 = (supp & ABIT_STRAND) == 0;
		f.exonic1 = split & ABIT_EXONIC; f.exonic2 = supp & ABIT_EXONIC;
		f.anchor1 = (mate1 & ABIT_STRAND) ? b.start[MATE1][i] : b.end[MATE1][i];
		f.anchor2 = (supp & ABIT_STRAND) ? b.start[SUPPLEMENTARY][i] : b.end[SUPPLEMENTARY][i];
		f.slot1 = SPLIT_READ; f.slot2 = SUPPLEMENTARY;
	} else {
		uint8_t mate1 = b.abits[MATE1][i], mate2 = b.abits[MATE2][i];
		f.contig1 = b.contig[MATE1][i]; f.contig2 = b.contig[MATE2][i];
		f.breakpoint1 = (mate1 & ABIT_STRAND) ? b.end[MATE1][i] : b.start[MATE1][i];
		f.breakpoint2 = (mate2 & ABIT_STRAND) ? b.end[MATE2][i] : b.start[MATE2][i];
		f.upstream1 = (mate1 & ABIT_STRAND) == 0;
		f.upstream2 = (mate2 & ABIT_STRAND) == 0;
		f.exonic1 = mate1 & ABIT_EXONIC; f.exonic2 = mate2 & ABIT_EXONIC;
		f.anchor1 = (mate1 & ABIT_STRAND) ? b.start[MATE1][i] : b.end[MATE1][i];
		f.anchor2 = (mate2 & ABIT_STRAND) ? b.start[MATE2][i] : b.end[MATE2][i];
		f.slot1 = MATE1; f.slot2 = MATE2;
	}
	f.swapped = false;
	if (f.contig1 > f.contig2 || (f.contig1 == f.contig2 && f.breakpoint1 > f.breakpoint2)) {
		uint32_t c = f.contig1; f.contig1 = f.contig2; f.contig2 = c;
		int32_t p = f.breakpoint1; f.breakpoint1 = f.breakpoint2; f.breakpoint2 = p;
		bool d = f.upstream1; f.upstream1 = f.upstream2; f.upstream2 = d;
		bool x = f.exonic1; f.exonic1 = f.exonic2; f.exonic2 = x;
		int32_t a = f.anchor1; f.anchor1 = f.anchor2; f.anchor2 = a;
		int s = f.slot1; f.slot1 = f.slot2; f.slot2 = s;
		f.swapped = true;
	}
}

AGPU_HD uint32_t emission_count(const BatchView& b, uint64_t i) {
	FragmentEnds f; fragment_ends(b, i, f);
	return (uint32_t) b.gene_count[f.slot1][i] * (uint32_t) b.gene_count[f.slot2][i];
}

AGPU_HD void write_emissions(const BatchView& b, uint64_t i, FusionEmission* out) {
	FragmentEnds f; fragment_ends(b, i, f);
	AGPU_IDSET(genes1); AGPU_IDSET(genes2);
	load_genes(b, f.slot1, i, genes1); load_genes(b, f.slot2, i, genes2);
	FusionEmission e;
	e.breakpoint1 = f.breakpoint1; e.breakpoint2 = f.breakpoint2; e.contigs = f.contig1 << 16 | f.contig2;
	e.info = (f.upstream1 ? EINFO_UPSTREAM1 : 0) | (f.upstream2 ? EINFO_UPSTREAM2 : 0) | (f.swapped && f.is_split ? EINFO_SWAPPED : 0) | (f.swapped && !f.is_split ? EINFO_MATES_SWAPPED : 0) | (f.exonic1 ? EINFO_EXONIC1 : 0) | (f.exonic2 ? EINFO_EXONIC2 : 0) |
	         (f.is_split ? EINFO_SPLIT : 0) | ((uint32_t) b.filter[i] << EINFO_FILTER_SHIFT) |
	         ((b.abits[f.slot1][i] & ABIT_PREDICTED_STRAND) ? EINFO_PREDICTED1 : 0) | ((b.abits[f.slot1][i] & ABIT_PREDICTED_STRAND_AMBIGUOUS) ? EINFO_AMBIGUOUS1 : 0) |
	         ((b.abits[f.slot2][i] & ABIT_PREDICTED_STRAND) ? EINFO_PREDICTED2 : 0);
	e.anchor1 = f.anchor1; e.anchor2 = f.anchor2; e.read = (uint32_t) (b.first_rank + i);
	uint32_t k = 0;
	for (uint32_t g1 = 0; g1 < genes1.n; ++g1)
		for (uint32_t g2 = 0; g2 < genes2.n; ++g2) {
			e.gene1 = genes1.get(g1); e.gene2 = genes2.get(g2);
			e.info = (e.info & ~(255u << EINFO_ORDINAL_SHIFT)) | (k & 255u) << EINFO_ORDINAL_SHIFT;
			out[k++] = e;
		}
}

// ---- prefix counts within a candidate (segmented inclusive scan) ---------------------------------------------------

struct RankState { // counts of split-read emissions seen so far in this candidate, per side
	uint32_t head;      // 1 at the first emission of a candidate (resets the scan)
	uint32_t reads[2];  // split reads on side 0 (not swapped) / 1 (swapped)
	uint32_t unfiltered[2];
};
struct RankCombine {
	AGPU_HD RankState operator()(const RankState& a, const RankState& b) const {
		if (b.head) return b;
		RankState r;
		r.head = a.head;
		r.reads[0] = a.reads[0] + b.reads[0]; r.reads[1] = a.reads[1] + b.reads[1];
		r.unfiltered[0] = a.unfiltered[0] + b.unfiltered[0]; r.unfiltered[1] = a.unfiltered[1] + b.unfiltered[1];
		return r;
	}
};
AGPU_HD RankState rank_input(const FusionEmission& e, bool is_head) {
	RankState r; r.head = is_head; r.reads[0] = r.reads[1] = r.unfiltered[0] = r.unfiltered[1] = 0;
	if (e.info & EINFO_SPLIT) {
		int side = (e.info & EINFO_SWAPPED) ? 1 : 0;
		r.reads[side] = 1;
		if ((e.info >> EINFO_FILTER_SHIFT & 255) == FILTER_none) r.unfiltered[side] = 1;
	}
	return r;
}
// does this split read join its read list? `inclusive` = scan result at this emission
AGPU_HD bool joins_split_read_list(const FusionEmission& e, const RankState& inclusive, uint32_t threshold) {
	int side = (e.info & EINFO_SWAPPED) ? 1 : 0;
	bool unfiltered = (e.info >> EINFO_FILTER_SHIFT & 255) == FILTER_none;
	uint32_t position = inclusive.reads[side] - 1;                         // reads of this side before this one
	uint32_t unfiltered_before = inclusive.unfiltered[side] - (unfiltered ? 1 : 0);
	return position < threshold || (unfiltered && unfiltered_before < threshold);
}

// running anchor with "0 means unset" (source/fusions.cpp:276-285): downstream keeps the minimum, upstream the maximum.
// A sequence of updates is summarised as (has_zero, value): if the sequence contains an anchor equal to 0, everything before the
// last zero is forgotten.
struct AnchorFold {
	uint32_t has_zero; int32_t value; // value 0 = no non-zero anchor (after the last zero)
};
AGPU_HD AnchorFold anchor_identity() { AnchorFold f; f.has_zero = 0; f.value = 0; return f; }
// an anchor of 0 resets a downstream (minimum) anchor but is a no-op for an upstream (maximum) one: `a > anchor || anchor == 0`
AGPU_HD AnchorFold anchor_single(int32_t anchor, bool upstream) { AnchorFold f; f.has_zero = !upstream && anchor == 0; f.value = anchor; return f; }
AGPU_HD AnchorFold anchor_combine(const AnchorFold& first, const AnchorFold& second, bool upstream) {
	if (second.has_zero) return second;
	AnchorFold r; r.has_zero = first.has_zero;
	if (first.value == 0) r.value = second.value;
	else if (second.value == 0) r.value = first.value;
	else r.value = upstream ? (first.value > second.value ? first.value : second.value) : (first.value < second.value ? first.value : second.value);
	return r;
}
AGPU_HD int32_t anchor_apply(int32_t current, const AnchorFold& fold, bool upstream) {
	if (fold.has_zero) return fold.value;
	if (fold.value == 0) return current;
	if (current == 0) return fold.value;
	return upstream ? (current > fold.value ? current : fold.value) : (current < fold.value ? current : fold.value);
}

// per-candidate aggregate of its emissions (all fields associative; order = name order)
struct CandidateFold {
	uint32_t head;
	uint32_t info_or;               // EINFO_EXONIC1/2 OR-ed over all emissions
	uint32_t any_unfiltered;
	uint32_t first_filter;          // filter of the first emission
	uint32_t first_not_duplicate;   // filter of the first emission whose filter is neither none nor duplicates; 0 = none seen
	uint32_t split_reads[2];        // unfiltered split reads that joined list 1 / 2
	uint32_t list_size[2];
	AnchorFold anchor1, anchor2;
	uint32_t upstream_bits;         // directions (needed by the anchor combine)
};
struct CandidateCombine {
	AGPU_HD CandidateFold operator()(const CandidateFold& a, const CandidateFold& b) const {
		if (b.head) return b;
		CandidateFold r;
		r.head = a.head; r.upstream_bits = a.upstream_bits;
		r.info_or = a.info_or | b.info_or;
		r.any_unfiltered = a.any_unfiltered | b.any_unfiltered;
		r.first_filter = a.first_filter;
		r.first_not_duplicate = a.first_not_duplicate ? a.first_not_duplicate : b.first_not_duplicate;
		r.split_reads[0] = a.split_reads[0] + b.split_reads[0]; r.split_reads[1] = a.split_reads[1] + b.split_reads[1];
		r.list_size[0] = a.list_size[0] + b.list_size[0]; r.list_size[1] = a.list_size[1] + b.list_size[1];
		r.anchor1 = anchor_combine(a.anchor1, b.anchor1, a.upstream_bits & EINFO_UPSTREAM1);
		r.anchor2 = anchor_combine(a.anchor2, b.anchor2, a.upstream_bits & EINFO_UPSTREAM2);
		return r;
	}
};
AGPU_HD CandidateFold candidate_input(const FusionEmission& e, bool is_head, bool joins_list) {
	CandidateFold f;
	uint32_t filter = e.info >> EINFO_FILTER_SHIFT & 255;
	f.head = is_head; f.upstream_bits = e.info & 3u;
	f.info_or = e.info & (EINFO_EXONIC1 | EINFO_EXONIC2);
	f.any_unfiltered = filter == FILTER_none;
	f.first_filter = filter;
	f.first_not_duplicate = (filter != FILTER_none && filter != FILTER_duplicates) ? filter : 0;
	f.split_reads[0] = f.split_reads[1] = f.list_size[0] = f.list_size[1] = 0;
	bool is_split = e.info & EINFO_SPLIT;
	bool contributes_anchor = !is_split || joins_list; // discordant mates always update the anchors in the first loop (:347-357)
	if (is_split && joins_list) {
		int side = (e.info & EINFO_SWAPPED) ? 1 : 0;
		f.list_size[side] = 1;
		if (filter == FILTER_none) f.split_reads[side] = 1;
	}
	f.anchor1 = contributes_anchor ? anchor_single(e.anchor1, e.info & EINFO_UPSTREAM1) : anchor_identity();
	f.anchor2 = contributes_anchor ? anchor_single(e.anchor2, e.info & EINFO_UPSTREAM2) : anchor_identity();
	return f;
}
AGPU_HD uint8_t candidate_filter(const CandidateFold& f) { // source/fusions.cpp:262-263
	if (f.any_unfiltered) return FILTER_none;
	if (f.first_not_duplicate) return (uint8_t) f.first_not_duplicate;
	return (uint8_t) f.first_filter; // all duplicates
}

// ---- candidates -------------------------------------------------------------------------------------------------------

// bits of CandidateTable::flags
enum : uint32_t { CFLAG_UPSTREAM1 = 1, CFLAG_UPSTREAM2 = 2, CFLAG_EXONIC1 = 4, CFLAG_EXONIC2 = 8, CFLAG_SPLICED1 = 16, CFLAG_SPLICED2 = 32, CFLAG_PREDICTED_STRAND1 = 64, CFLAG_PREDICTED_STRAND2 = 128,
                  CFLAG_PREDICTED_STRANDS_AMBIGUOUS = 256, CFLAG_TRANSCRIPT_START_GENE1 = 512, CFLAG_TRANSCRIPT_START_AMBIGUOUS = 1024 };

struct CandidateTable { // structure of arrays, index = order of first occurrence in name order (== the reference's insertion order)
	uint32_t n;
	uint32_t* gene1; uint32_t* gene2; uint32_t* contigs; int32_t* breakpoint1; int32_t* breakpoint2;
	uint32_t* flags; uint8_t* filter;
	uint32_t* split_reads1; uint32_t* split_reads2; uint32_t* discordant_mates;
	int32_t* anchor1; int32_t* anchor2;
	uint32_t* votes;                // [2*n] strand votes of the reads in the lists: forward, reverse (source/fusions.cpp:22-79)
	uint64_t* list_offset;          // [3*n + 1] into read_lists (64-bit: a candidate holds up to -U reads per list, and with -U 32767 the lists of a sample pass 2^32 entries): split_read1_list, split_read2_list, discordant_mate_list of candidate c at 3c, 3c+1, 3c+2
	uint32_t* read_lists;
	// Round 5: the discordant-mate lists may be IMPLICIT.  Every candidate of a gene pair lists the discordant mates of the pair that lie near its breakpoints, up to the subsampling
	// threshold (source/fusions.cpp:379-437) -- with -U 32767 (BASELINE.json config 3) that is 92.7 G entries for 10^8 fragments, which nobody can hold (the reference would
	// need 740 GB).  A discordant list is a function of (the bucket of its gene pair, the predicate of the candidate, the threshold); when the lists of a sample are too many to keep,
	// only that function is kept (agpu_fusions.hip: bucket range and predicate per candidate), list_offset stays what it is -- positions in lists that exist only in the mind --
	// and read_lists holds the split-read lists alone, packed: discordant_before != nullptr, and entry k of a split-read list of candidate c lies at read_lists[k - discordant_before[c]].
	// A stage that walks discordant lists gets them a WINDOW of candidates at a time (for_each_list_window): a table like this one whose read_lists is a buffer that holds all three
	// lists of the candidates of the window at their positions k, and discordant_before == nullptr -- the kernels of the stages cannot tell the difference.
	const uint64_t* discordant_before = nullptr; // [n + 1] entries of discordant lists in front of candidate c's
};
// entry k of split_read1_list / split_read2_list of candidate c (k between list_offset[3c] and list_offset[3c + 2]): valid on the table of the sample and on the table of a window
AGPU_HD uint32_t split_list_entry(const CandidateTable& t, uint32_t c, uint64_t k) { return t.read_lists[t.discordant_before != nullptr ? k - t.discordant_before[c] : k]; }

AGPU_HD bool candidate_is_intragenic(const AnnotationView& ann, uint32_t gene1, uint32_t gene2, int32_t breakpoint1, int32_t breakpoint2) { // source/common.hpp:275-279
	return gene1 == gene2 || (breakpoint1 >= ann.gene_start[gene2] - 10000 && breakpoint1 <= ann.gene_end[gene2] + 10000 &&
	                          breakpoint2 >= ann.gene_start[gene1] - 10000 && breakpoint2 <= ann.gene_end[gene1] + 10000);
}

// does a discordant mate of the same gene pair support this candidate? (source/fusions.cpp:379-396)  What depends on the candidate alone is worked out once (a candidate of a
// hot gene pair asks the question tens of thousands of times: the gene table is not looked up per mate).
struct DiscordantMatePredicate {
	int32_t breakpoint1, breakpoint2, limit1, limit2, max_mate_gap, gene1_start, gene1_end, gene2_start, gene2_end;
	bool upstream1, upstream2, intragenic;
	AGPU_HD void set(const AnnotationView& ann, uint32_t gene1, uint32_t gene2, int32_t candidate_breakpoint1, int32_t candidate_breakpoint2, bool candidate_upstream1, bool candidate_upstream2, bool has_split_reads, int32_t gap) {
		breakpoint1 = candidate_breakpoint1; breakpoint2 = candidate_breakpoint2; upstream1 = candidate_upstream1; upstream2 = candidate_upstream2; max_mate_gap = gap;
		const int32_t max_overlap = has_split_reads ? 2 : gap;
		limit1 = upstream1 ? breakpoint1 - max_overlap : breakpoint1 + max_overlap;
		limit2 = upstream2 ? breakpoint2 - max_overlap : breakpoint2 + max_overlap;
		gene1_start = ann.gene_start[gene1]; gene1_end = ann.gene_end[gene1]; gene2_start = ann.gene_start[gene2]; gene2_end = ann.gene_end[gene2];
		intragenic = candidate_is_intragenic(ann, gene1, gene2, breakpoint1, breakpoint2);
	}
	AGPU_HD bool supports(int32_t mate_breakpoint1, int32_t mate_breakpoint2) const {
		if (!(upstream1 ? mate_breakpoint1 >= limit1 : mate_breakpoint1 <= limit1)) return false;
		if (!(upstream2 ? mate_breakpoint2 >= limit2 : mate_breakpoint2 <= limit2)) return false;
		int32_t d1 = breakpoint1 - mate_breakpoint1; if (d1 < 0) d1 = -d1;
		int32_t d2 = breakpoint2 - mate_breakpoint2; if (d2 < 0) d2 = -d2;
		const bool outside_other_gene = !intragenic && !(mate_breakpoint1 >= gene2_start && mate_breakpoint1 <= gene2_end) && !(mate_breakpoint2 >= gene1_start && mate_breakpoint2 <= gene1_end);
		return outside_other_gene || (d1 <= max_mate_gap && d2 <= max_mate_gap);
	}
};
AGPU_HD bool discordant_mate_supports(const AnnotationView& ann, uint32_t gene1, uint32_t gene2, int32_t breakpoint1, int32_t breakpoint2, bool upstream1, bool upstream2,
                                      bool has_split_reads, int32_t max_mate_gap, int32_t mate_breakpoint1, int32_t mate_breakpoint2) {
	DiscordantMatePredicate predicate;
	predicate.set(ann, gene1, gene2, breakpoint1, breakpoint2, upstream1, upstream2, has_split_reads, max_mate_gap);
	return predicate.supports(mate_breakpoint1, mate_breakpoint2);
}

// which alignment slot of a discordant fragment holds the mate with the lower (contig, breakpoint)? (source/fusions.cpp:414-421)
AGPU_HD bool discordant_mates_need_swap(const BatchView& b, uint64_t i) {
	uint8_t bits1 = b.abits[MATE1][i], bits2 = b.abits[MATE2][i];
	int32_t breakpoint1 = (bits1 & ABIT_STRAND) ? b.end[MATE1][i] : b.start[MATE1][i];
	int32_t breakpoint2 = (bits2 & ABIT_STRAND) ? b.end[MATE2][i] : b.start[MATE2][i];
	uint32_t contig1 = b.contig[MATE1][i], contig2 = b.contig[MATE2][i];
	return contig1 > contig2 || (contig1 == contig2 && breakpoint1 > breakpoint2);
}

// Strand votes from the emission record alone; 0 no vote, 1 forward, 2 reverse.
// A split read votes with the predicted strand of SPLIT_READ (split_read1_list) or SUPPLEMENTARY (split_read2_list), i.e. with the
// alignment on side 1 of its ordered breakpoint pair (source/fusions.cpp:22-40).
AGPU_HD int split_read_vote(uint32_t info) { return (info & EINFO_AMBIGUOUS1) ? 0 : (info & EINFO_PREDICTED1) ? 1 : 2; }
// A discordant mate attached to a candidate shares its gene pair and directions; the reference picks the mate that belongs to gene1 by
// contig and strand (always the side-1 mate here) or, if both mates have the same strand, by distance to the breakpoints
// (source/fusions.cpp:42-79).  mate_breakpoint1/2 = ends of the side-1 / side-2 mate.
AGPU_HD int discordant_mate_vote(uint32_t info, bool upstream1, bool upstream2, int32_t breakpoint1, int32_t breakpoint2, int32_t mate_breakpoint1, int32_t mate_breakpoint2) {
	if ((info & EINFO_AMBIGUOUS1) || (info >> EINFO_FILTER_SHIFT & 255) == FILTER_hairpin) return 0;
	bool use_side2 = false;
	if (upstream1 == upstream2) { // both mates on the same strand
		int32_t a = breakpoint1 - mate_breakpoint1; if (a < 0) a = -a;
		int32_t c = breakpoint2 - mate_breakpoint2; if (c < 0) c = -c;
		int32_t d = breakpoint2 - mate_breakpoint1; if (d < 0) d = -d;
		int32_t e = breakpoint1 - mate_breakpoint2; if (e < 0) e = -e;
		uint32_t distance1 = (uint32_t) a + (uint32_t) c, distance2 = (uint32_t) d + (uint32_t) e;
		if (distance1 == distance2) return 0;
		use_side2 = distance2 < distance1;
	}
	return (info & (use_side2 ? EINFO_PREDICTED2 : EINFO_PREDICTED1)) ? 1 : 2;
}

// The discordant emissions grouped by gene pair and directions (name order inside a bucket), as columns
struct DiscordantBuckets { const int32_t* breakpoint1; const int32_t* breakpoint2; const uint32_t* info; const uint32_t* read; const int32_t* anchor1; const int32_t* anchor2; };

// Attach the discordant mates of the candidate's gene pair (bucket = their emissions in name order) to candidate c
// (source/fusions.cpp:367-437).  With out_list == NULL only the list size is returned (count pass); otherwise the list is
// written, the anchors and the unfiltered count are updated and the fragments whose mates the reference swaps are flagged.
// What a walk over the bucket does: ATTACH_COUNT the size of the list; ATTACH_FILL the pass behind it -- the list written, the anchors folded, the unfiltered mates and the
// strand votes counted, the fragments whose mates the reference swaps flagged; ATTACH_FOLD the same without writing the list (implicit lists); ATTACH_EXPAND a later expansion
// of an implicit list: the same entries, nothing else touched (the candidate may have been filtered since).  mode < 0: COUNT without out_list, FILL with one.
enum { ATTACH_COUNT = 0, ATTACH_FILL = 1, ATTACH_FOLD = 2, ATTACH_EXPAND = 3 };
AGPU_HD uint32_t attach_discordant_mates(const AnnotationView& ann, const CandidateTable& t, uint32_t c, const DiscordantBuckets& buckets, uint32_t bucket_begin, uint32_t bucket_size,
                                         int32_t max_mate_gap, uint32_t threshold, bool has_split_reads, uint32_t* out_list, uint8_t* discordant_swapped, int mode = -1) {
	if (mode < 0) mode = out_list != nullptr ? ATTACH_FILL : ATTACH_COUNT;
	if (mode != ATTACH_EXPAND && t.filter[c] != FILTER_none) return 0;
	uint32_t flags = t.flags[c];
	bool upstream1 = flags & CFLAG_UPSTREAM1, upstream2 = flags & CFLAG_UPSTREAM2;
	uint32_t gene1 = t.gene1[c], gene2 = t.gene2[c];
	int32_t breakpoint1 = t.breakpoint1[c], breakpoint2 = t.breakpoint2[c];
	uint32_t list_size = 0, unfiltered = 0, forward_votes = 0, reverse_votes = 0;
	AnchorFold fold1 = anchor_identity(), fold2 = anchor_identity();
	for (uint32_t k = bucket_begin; k < bucket_begin + bucket_size; ++k) {
		if (!discordant_mate_supports(ann, gene1, gene2, breakpoint1, breakpoint2, upstream1, upstream2, has_split_reads, max_mate_gap, buckets.breakpoint1[k], buckets.breakpoint2[k]))
			continue;
		uint32_t info = buckets.info[k];
		bool read_unfiltered = (info >> EINFO_FILTER_SHIFT & 255) == FILTER_none;
		if (!read_unfiltered && list_size >= threshold) continue;
		if (unfiltered >= threshold) break;
		if (mode == ATTACH_FILL || mode == ATTACH_EXPAND) out_list[list_size] = buckets.read[k];
		if (mode == ATTACH_FILL || mode == ATTACH_FOLD) {
			uint32_t read = buckets.read[k];
			if ((info & EINFO_MATES_SWAPPED) && !discordant_swapped[read]) discordant_swapped[read] = 1;
			fold1 = anchor_combine(fold1, anchor_single(buckets.anchor1[k], upstream1), upstream1);
			fold2 = anchor_combine(fold2, anchor_single(buckets.anchor2[k], upstream2), upstream2);
			int vote = discordant_mate_vote(info, upstream1, upstream2, breakpoint1, breakpoint2, buckets.breakpoint1[k], buckets.breakpoint2[k]);
			if (vote == 1) ++forward_votes; else if (vote == 2) ++reverse_votes;
		}
		++list_size;
		if (read_unfiltered) ++unfiltered;
	}
	if (mode == ATTACH_FILL || mode == ATTACH_FOLD) {
		t.discordant_mates[c] = unfiltered;
		t.anchor1[c] = anchor_apply(t.anchor1[c], fold1, upstream1);
		t.anchor2[c] = anchor_apply(t.anchor2[c], fold2, upstream2);
		t.votes[2 * (uint64_t) c] += forward_votes; t.votes[2 * (uint64_t) c + 1] += reverse_votes;
	}
	return list_size;
}

// ---- strands, splice sites, transcript start (source/fusions.cpp:15-200, 443-470) -----------------------------------

// vote of one discordant mate for the strand of gene1's side (source/fusions.cpp:42-79); returns 0 no vote, 1 forward, 2 reverse
AGPU_HD int discordant_strand_vote(const BatchView& b, uint64_t i, uint32_t candidate_contig1, bool upstream1, int32_t breakpoint1, int32_t breakpoint2, bool mates_swapped) {
	// after find_fusions' in-place swap MATE1 is the mate with the lower (contig, breakpoint)
	int first = mates_swapped ? MATE2 : MATE1, second = mates_swapped ? MATE1 : MATE2;
	if ((b.abits[first][i] & ABIT_PREDICTED_STRAND_AMBIGUOUS) || b.filter[i] == FILTER_hairpin) return 0;
	int mate1 = first, mate2 = second;
	bool forward1 = b.abits[mate1][i] & ABIT_STRAND;
	if (b.contig[mate1][i] != candidate_contig1 || (forward1 != !upstream1)) {
		int t = mate1; mate1 = mate2; mate2 = t;
	} else if (((b.abits[mate1][i] ^ b.abits[mate2][i]) & ABIT_STRAND) == 0) {
		int32_t end1 = upstream1 ? b.start[mate1][i] : b.end[mate1][i];
		int32_t end2 = upstream1 ? b.start[mate2][i] : b.end[mate2][i];
		int32_t a = breakpoint1 - end1; if (a < 0) a = -a;
		int32_t c = breakpoint2 - end2; if (c < 0) c = -c;
		int32_t d = breakpoint2 - end1; if (d < 0) d = -d;
		int32_t e = breakpoint1 - end2; if (e < 0) e = -e;
		uint32_t distance1 = (uint32_t) a + (uint32_t) c, distance2 = (uint32_t) d + (uint32_t) e;
		if (distance1 == distance2) return 0;
		if (distance2 < distance1) { int t = mate1; mate1 = mate2; mate2 = t; }
	}
	return (b.abits[mate1][i] & ABIT_PREDICTED_STRAND) ? 1 : 2;
}

struct TranscriptStart { bool gene1; bool ambiguous; };

// reference: predict_transcript_start, source/fusions.cpp:93-200; may also resolve ambiguous strands (in/out through flags)
AGPU_HD void predict_transcript_start(const AnnotationView& ann, uint32_t gene1, uint32_t gene2, uint32_t contigs, int32_t breakpoint1, int32_t breakpoint2,
                                      uint32_t split_reads, uint32_t& flags) {
	bool upstream1 = flags & CFLAG_UPSTREAM1, upstream2 = flags & CFLAG_UPSTREAM2;
	bool spliced1 = flags & CFLAG_SPLICED1, spliced2 = flags & CFLAG_SPLICED2, exonic1 = flags & CFLAG_EXONIC1, exonic2 = flags & CFLAG_EXONIC2;
	bool strands_ambiguous = flags & CFLAG_PREDICTED_STRANDS_AMBIGUOUS;
	bool strand1 = flags & CFLAG_PREDICTED_STRAND1, strand2 = flags & CFLAG_PREDICTED_STRAND2;
	bool gene1_forward = ann.gene_bits[gene1] & GBIT_STRAND, gene2_forward = ann.gene_bits[gene2] & GBIT_STRAND;
	bool gene1_dummy = ann.gene_bits[gene1] & GBIT_DUMMY, gene2_dummy = ann.gene_bits[gene2] & GBIT_DUMMY;
	bool is_read_through = (contigs >> 16) == (contigs & 0xFFFF) && breakpoint2 - breakpoint1 < 400000 && !upstream1 && upstream2; // source/common.hpp:265-269
	bool start_gene1 = true, ambiguous = false;
	if (spliced1 || (!strands_ambiguous && !gene1_dummy && strand1 == gene1_forward)) {
		start_gene1 = (gene1_forward && !upstream1) || (!gene1_forward && upstream1);
	} else if (spliced2 || (!strands_ambiguous && !gene2_dummy && strand2 == gene2_forward)) {
		start_gene1 = !((gene2_forward && !upstream2) || (!gene2_forward && upstream2));
	} else if (!strands_ambiguous) {
		if (((strand1 && !upstream1) || (!strand1 && upstream1)) && ((!strand2 && !upstream2) || (strand2 && upstream2))) start_gene1 = true;
		else if (((strand2 && !upstream2) || (!strand2 && upstream2)) && ((!strand1 && !upstream1) || (strand1 && upstream1))) start_gene1 = false;
		else ambiguous = true;
	} else if (!exonic1 && !exonic2) {
		ambiguous = true;
	} else if (!exonic1 && exonic2) {
		if (gene2_forward && !upstream2) start_gene1 = false;
		else if (!gene2_forward && upstream2) start_gene1 = false;
		else if (split_reads == 0 && is_read_through && ((gene2_forward && upstream2) || (!gene2_forward && !upstream2))) start_gene1 = true;
		else ambiguous = true;
	} else if (!exonic2 && exonic1) {
		if (gene1_forward && !upstream1) start_gene1 = true;
		else if (!gene1_forward && upstream1) start_gene1 = true;
		else if (split_reads == 0 && is_read_through && ((gene1_forward && upstream1) || (!gene1_forward && !upstream1))) start_gene1 = true;
		else ambiguous = true;
	} else {
		if ((!gene1_dummy && gene1_forward && !upstream1) || (!gene1_forward && upstream1)) start_gene1 = true;
		else if ((!gene2_dummy && gene2_forward && !upstream2) || (!gene2_forward && upstream2)) start_gene1 = false;
		else ambiguous = true;
	}
	if (ambiguous) start_gene1 = true;
	if (!ambiguous && strands_ambiguous) {
		strands_ambiguous = false;
		if (start_gene1) { strand1 = gene1_forward; strand2 = complement_strand_if(strand1, upstream1 == upstream2); }
		else { strand2 = gene2_forward; strand1 = complement_strand_if(strand2, upstream1 == upstream2); }
	}
	flags &= ~(CFLAG_PREDICTED_STRAND1 | CFLAG_PREDICTED_STRAND2 | CFLAG_PREDICTED_STRANDS_AMBIGUOUS | CFLAG_TRANSCRIPT_START_GENE1 | CFLAG_TRANSCRIPT_START_AMBIGUOUS);
	flags |= (strand1 ? CFLAG_PREDICTED_STRAND1 : 0) | (strand2 ? CFLAG_PREDICTED_STRAND2 : 0) | (strands_ambiguous ? CFLAG_PREDICTED_STRANDS_AMBIGUOUS : 0) |
	         (start_gene1 ? CFLAG_TRANSCRIPT_START_GENE1 : 0) | (ambiguous ? CFLAG_TRANSCRIPT_START_AMBIGUOUS : 0);
}

// strand vote of entry k of the concatenated read lists of candidate c: 0 none, 1 forward, 2 reverse (source/fusions.cpp:22-79)
AGPU_HD int list_entry_strand_vote(const BatchView& b, const CandidateTable& t, const uint8_t* discordant_swapped, uint32_t c, uint64_t k) {
	const uint64_t* offsets = t.list_offset + 3 * (uint64_t) c;
	uint32_t read = t.read_lists[k];
	if (k < offsets[2]) { // split_read1_list votes with SPLIT_READ's predicted strand, split_read2_list with SUPPLEMENTARY's
		uint8_t bits = b.abits[k < offsets[1] ? SPLIT_READ : SUPPLEMENTARY][read];
		if (bits & ABIT_PREDICTED_STRAND_AMBIGUOUS) return 0;
		return (bits & ABIT_PREDICTED_STRAND) ? 1 : 2;
	}
	return discordant_strand_vote(b, read, t.contigs[c] >> 16, t.flags[c] & CFLAG_UPSTREAM1, t.breakpoint1[c], t.breakpoint2[c], discordant_swapped[read]);
}

// strands, splice sites and transcript start from the vote counts (source/fusions.cpp:81-87, 443-470)
AGPU_HD void finalize_candidate(const AnnotationView& ann, const CandidateTable& t, uint32_t c, uint32_t forward, uint32_t reverse) {
	uint32_t flags = t.flags[c];
	uint32_t gene1 = t.gene1[c], gene2 = t.gene2[c];
	int32_t breakpoint1 = t.breakpoint1[c], breakpoint2 = t.breakpoint2[c];
	bool upstream1 = flags & CFLAG_UPSTREAM1, upstream2 = flags & CFLAG_UPSTREAM2;
	const uint64_t* offsets = t.list_offset + 3 * (uint64_t) c;
	flags &= ~(CFLAG_PREDICTED_STRAND1 | CFLAG_PREDICTED_STRAND2 | CFLAG_PREDICTED_STRANDS_AMBIGUOUS | CFLAG_SPLICED1 | CFLAG_SPLICED2);
	if (forward == reverse) {
		flags |= CFLAG_PREDICTED_STRANDS_AMBIGUOUS | CFLAG_PREDICTED_STRAND1 | CFLAG_PREDICTED_STRAND2; // fusion_t() initialises both strands to FORWARD
	} else {
		bool strand1 = forward > reverse;
		bool strand2 = complement_strand_if(strand1, upstream1 == upstream2);
		flags |= (strand1 ? CFLAG_PREDICTED_STRAND1 : 0) | (strand2 ? CFLAG_PREDICTED_STRAND2 : 0);
	}
	bool has_split_reads = offsets[2] > offsets[0];
	if (has_split_reads && !(flags & CFLAG_PREDICTED_STRANDS_AMBIGUOUS)) {
		bool gene1_forward = ann.gene_bits[gene1] & GBIT_STRAND, gene2_forward = ann.gene_bits[gene2] & GBIT_STRAND;
		if ((flags & CFLAG_EXONIC1) && gene1_forward == ((flags & CFLAG_PREDICTED_STRAND1) != 0) && is_breakpoint_spliced(ann, gene1, upstream1, breakpoint1)) flags |= CFLAG_SPLICED1;
		if ((flags & CFLAG_EXONIC2) && gene2_forward == ((flags & CFLAG_PREDICTED_STRAND2) != 0) && is_breakpoint_spliced(ann, gene2, upstream2, breakpoint2)) flags |= CFLAG_SPLICED2;
	}
	predict_transcript_start(ann, gene1, gene2, t.contigs[c], breakpoint1, breakpoint2, t.split_reads1[c] + t.split_reads2[c], flags);
	t.flags[c] = flags;
}

// sequential form of the strand vote: walks the read lists of candidate c and looks at the reads themselves (the device path collects
// the same votes from the emission records while it fills the lists; tests/emu checks the two against each other).
// discordant_swapped[i] tells whether find_fusions swapped MATE1/MATE2 of fragment i in place.
AGPU_HD void count_list_votes(const BatchView& b, const CandidateTable& t, const uint8_t* discordant_swapped, uint32_t c, uint32_t& forward, uint32_t& reverse) {
	const uint64_t* offsets = t.list_offset + 3 * (uint64_t) c;
	forward = 0; reverse = 0;
	for (uint64_t k = offsets[0]; k < offsets[3]; ++k) {
		int vote = list_entry_strand_vote(b, t, discordant_swapped, c, k);
		if (vote == 1) ++forward; else if (vote == 2) ++reverse;
	}
}

}

#endif
