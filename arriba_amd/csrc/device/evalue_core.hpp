// arriba_amd/csrc/device/evalue_core.hpp -- e-value of a candidate (reference: estimate_expected_fusions and
// filter_relative_support, source/filter_relative_support.cpp:17-224).
//
// The reference updates a `float` e-value by a chain of `*=` whose right-hand sides are doubles (round to float after
// every step) or unsigned counters (single-precision multiply), hazard H12.  Every pow() in that chain has an argument
// derived from a bounded integer (supporting reads, a distance < 1000 / < 400 / < 400000), so the host tabulates them
// once with its own libm and the device only multiplies and rounds -- IEEE-exact, hence bit-identical to the reference
// built on the same host.
#ifndef AGPU_EVALUE_CORE_HPP
#define AGPU_EVALUE_CORE_HPP 1

#include "fusion_core.hpp"

namespace agpu {

struct EvalueGlobals { // sample-wide covariates (source/filter_relative_support.cpp:43-126) after the "reasonable defaults" rules
	uint32_t spliced_breakpoints, exonic_breakpoints, intronic_breakpoints, exonic_intronic_breakpoints;
	uint32_t intragenic_duplications, intragenic_inversions;
	double intragenic_scale;      // 2.0 / (duplications + inversions)
	double intragenic_excess;     // max(1.0, spliced_events_in_same_gene / 0.25 / spliced_events_in_different_genes)
	double location_scale;        // 4.0 / (sum of the four breakpoint classes)
	double read_through_penalty;  // 1 + pow((fraction_of_genes_with_read_through_fusions - 0.25) * 20, 2)
	uint32_t read_through_penalty_applies; // fraction > 0.25
};

struct EvalueTables { // host-tabulated factors, indexed by the integer the reference feeds into pow()
	const double* support_scale;       // [max_support+1] max(1.0, mapped_reads / 20000000.0 * pow(0.02, supporting_reads - 2u))
	const double* intragenic_support;  // [max_support+1] pow(supporting_reads - 0.42, -2.11) * pow(10, -1.11)
	const double* intergenic_support;  // [max_support+1] pow(supporting_reads - 0.73, -2.28) * pow(10, -1.75)
	uint32_t max_support;
	const double* distance_1000;       // [1000]   pow(d / 1000.0, -2),    d = max(400, spliced_distance)
	const double* distance_400;        // [400]    pow(d / 400.0, -4.58),  d = max(1, spliced_distance)
	const double* read_through_distance; // [400000] pow(d / 400000.0, -0.63), d = max(1, breakpoint2 - breakpoint1)
	const double* proximal_distance;     // [400000] pow(d / 400000.0, -1.53)
};

AGPU_HD bool candidate_is_read_through(uint32_t contigs, int32_t breakpoint1, int32_t breakpoint2, uint32_t flags) { // source/common.hpp:265-269
	return (contigs >> 16) == (contigs & 0xFFFF) && breakpoint2 - breakpoint1 < 400000 && !(flags & CFLAG_UPSTREAM1) && (flags & CFLAG_UPSTREAM2);
}

AGPU_HD float times_double(float evalue, double factor) { return (float) ((double) evalue * factor); } // float *= double
AGPU_HD float times_count(float evalue, uint32_t count) { return evalue * (float) count; }              // float *= unsigned int

// source/filter_relative_support.cpp:130-206 for candidate c; partner_count = fusion_partner_count of :33-41 (0 = gene without entry)
AGPU_HD float candidate_evalue(const AnnotationView& ann, const CandidateTable& t, uint32_t c, const int32_t* partner_count, const EvalueGlobals& g, const EvalueTables& tables) {
	const uint32_t gene1 = t.gene1[c], gene2 = t.gene2[c], flags = t.flags[c], contigs = t.contigs[c];
	const int32_t breakpoint1 = t.breakpoint1[c], breakpoint2 = t.breakpoint2[c];
	const bool upstream1 = flags & CFLAG_UPSTREAM1, upstream2 = flags & CFLAG_UPSTREAM2;
	uint32_t supporting_reads = t.split_reads1[c] + t.split_reads2[c] + t.discordant_mates[c];
	if (supporting_reads > tables.max_support) supporting_reads = tables.max_support; // never: the tables are sized from the data

	int32_t count1 = partner_count[gene1] - 1, count2 = partner_count[gene2] - 1;
	if (count1 < 1) count1 = 1;
	if (count2 < 1) count2 = 1;
	double partners1 = 10000.0 / ann.gene_exonic_length[gene1] * count1, partners2 = 10000.0 / ann.gene_exonic_length[gene2] * count2;
	float max_fusion_partners = (float) ((partners1 < partners2) ? partners2 : partners1);

	float evalue = times_double(max_fusion_partners, tables.support_scale[supporting_reads]);

	if (candidate_is_intragenic(ann, gene1, gene2, breakpoint1, breakpoint2)) {
		evalue = times_double(evalue, g.intragenic_scale);
		if (upstream1 && !upstream2) evalue = times_count(evalue, g.intragenic_duplications);
		else if (upstream1 == upstream2) evalue = times_count(evalue, g.intragenic_inversions);
		if (supporting_reads >= 1) {
			evalue = times_double(evalue, tables.intragenic_support[supporting_reads]);
			int32_t distance = spliced_distance(ann, contigs >> 16, breakpoint1, breakpoint2, gene1);
			if (distance < 1000) {
				evalue = times_double(evalue, tables.distance_1000[distance > 400 ? distance : 400]);
				if (distance < 400) evalue = times_double(evalue, tables.distance_400[distance > 1 ? distance : 1]);
			}
		}
		evalue = times_double(evalue, g.intragenic_excess);
	} else if (supporting_reads >= 1) {
		evalue = times_double(evalue, tables.intergenic_support[supporting_reads]);
		int32_t distance = breakpoint2 - breakpoint1;
		if (distance < 1) distance = 1;
		if (candidate_is_read_through(contigs, breakpoint1, breakpoint2, flags)) evalue = times_double(evalue, tables.read_through_distance[distance]);
		else if ((contigs >> 16) == (contigs & 0xFFFF) && breakpoint2 - breakpoint1 < 400000) evalue = times_double(evalue, tables.proximal_distance[distance]);
	}

	evalue = times_double(evalue, g.location_scale);
	const bool exonic1 = flags & CFLAG_EXONIC1, exonic2 = flags & CFLAG_EXONIC2;
	uint32_t location = g.spliced_breakpoints;
	if (!(flags & (CFLAG_SPLICED1 | CFLAG_SPLICED2))) {
		uint32_t other = (exonic1 && exonic2) ? g.exonic_breakpoints : (!exonic1 && !exonic2) ? g.intronic_breakpoints : g.exonic_intronic_breakpoints;
		if (other > location) location = other;
	}
	evalue = times_count(evalue, location);

	if (g.read_through_penalty_applies && candidate_is_read_through(contigs, breakpoint1, breakpoint2, flags))
		evalue = times_double(evalue, g.read_through_penalty);
	return evalue;
}

// counters of source/filter_relative_support.cpp:43-126 contributed by candidate c
enum { EG_SPLICED = 0, EG_EXONIC = 1, EG_INTRONIC = 2, EG_MIXED = 3, EG_DUPLICATIONS = 4, EG_INVERSIONS = 5, EG_SPLICED_SAME_GENE = 6, EG_SPLICED_DIFFERENT_GENES = 7, EG_COUNT = 8 };
struct EvalueContribution { int breakpoint_class, intragenic_class, spliced_class; bool marks_genes, marks_read_through; };

AGPU_HD EvalueContribution evalue_contribution(const AnnotationView& ann, const CandidateTable& t, uint32_t c) {
	EvalueContribution r; r.breakpoint_class = -1; r.intragenic_class = -1; r.spliced_class = -1; r.marks_genes = false; r.marks_read_through = false;
	const uint32_t gene1 = t.gene1[c], gene2 = t.gene2[c], flags = t.flags[c], contigs = t.contigs[c];
	const int32_t breakpoint1 = t.breakpoint1[c], breakpoint2 = t.breakpoint2[c];
	const uint32_t split_reads = t.split_reads1[c] + t.split_reads2[c], supporting_reads = split_reads + t.discordant_mates[c];
	const bool unfiltered = t.filter[c] == FILTER_none;
	const bool dummy1 = ann.gene_bits[gene1] & GBIT_DUMMY, dummy2 = ann.gene_bits[gene2] & GBIT_DUMMY;
	const bool spliced1 = flags & CFLAG_SPLICED1, spliced2 = flags & CFLAG_SPLICED2, exonic1 = flags & CFLAG_EXONIC1, exonic2 = flags & CFLAG_EXONIC2;
	if (unfiltered && ((contigs >> 16) != (contigs & 0xFFFF) || breakpoint2 - breakpoint1 > 500000) && supporting_reads >= 2 && split_reads > 0 && !dummy1 && !dummy2)
		r.breakpoint_class = (spliced1 || spliced2) ? EG_SPLICED : (exonic1 && exonic2) ? EG_EXONIC : (!exonic1 && !exonic2) ? EG_INTRONIC : EG_MIXED;
	if (unfiltered && gene1 == gene2 && split_reads >= 2) {
		const bool upstream1 = flags & CFLAG_UPSTREAM1, upstream2 = flags & CFLAG_UPSTREAM2;
		if (upstream1 && !upstream2) r.intragenic_class = EG_DUPLICATIONS;
		else if (upstream1 == upstream2) r.intragenic_class = EG_INVERSIONS;
	}
	if (spliced1 && spliced2) r.spliced_class = (gene1 == gene2) ? EG_SPLICED_SAME_GENE : EG_SPLICED_DIFFERENT_GENES;
	if (!dummy1 && !dummy2 && split_reads > 0) {
		r.marks_genes = true;
		r.marks_read_through = candidate_is_read_through(contigs, breakpoint1, breakpoint2, flags);
	}
	return r;
}

// The three candidate predicates main() runs between the e-value and filter_relative_support (source/arriba.cpp:437-455):
// filter_non_coding_neighbors (source/filter_non_coding_neighbors.cpp:7-19), filter_intragenic_both_exonic
// (source/filter_intragenic_both_exonic.cpp:9-35) and filter_min_support (source/filter_min_support.cpp:7-19).  Each skips candidates
// that already have a filter, so for one candidate they are a cascade: returns the stage (0, 1, 2) that discards candidate c, or 3.
AGPU_HD bool breakpoint_overlaps_both_genes(const AnnotationView& ann, const CandidateTable& t, uint32_t c) { // source/common.hpp:260-264 (coordinates only, as in the reference)
	const uint32_t gene1 = t.gene1[c], gene2 = t.gene2[c];
	return (t.breakpoint1[c] >= ann.gene_start[gene2] && t.breakpoint1[c] <= ann.gene_end[gene2]) || (t.breakpoint2[c] >= ann.gene_start[gene1] && t.breakpoint2[c] <= ann.gene_end[gene1]);
}
AGPU_HD int candidate_predicate_stage(const AnnotationView& ann, const CandidateTable& t, uint32_t c, const uint8_t* enabled, float exonic_fraction, int32_t min_support) {
	const uint32_t gene1 = t.gene1[c], gene2 = t.gene2[c], flags = t.flags[c];
	if (enabled[14 /* non_coding_neighbors */] && !(ann.gene_bits[gene1] & GBIT_PROTEIN_CODING) && !(ann.gene_bits[gene2] & GBIT_PROTEIN_CODING) &&
	    candidate_is_read_through(t.contigs[c], t.breakpoint1[c], t.breakpoint2[c], flags))
		return 0;
	const bool overlaps = breakpoint_overlaps_both_genes(ann, t, c);
	if (enabled[15 /* intragenic_exonic */] && (overlaps || gene1 == gene2) && (flags & CFLAG_EXONIC1) && (flags & CFLAG_EXONIC2) && !((flags & CFLAG_SPLICED1) && (flags & CFLAG_SPLICED2))) {
		const int32_t distance_spliced = spliced_distance(ann, t.contigs[c] >> 16, t.breakpoint1[c], t.breakpoint2[c], gene1);
		const int32_t distance = t.breakpoint2[c] - t.breakpoint1[c];
		if (distance_spliced == distance || 1.0 * distance_spliced / distance < (double) exonic_fraction) return 1;
	}
	const int32_t split_reads = (int32_t) (t.split_reads1[c] + t.split_reads2[c]);
	if (enabled[17 /* min_support */] && (split_reads + (int32_t) t.discordant_mates[c] < min_support || (overlaps && split_reads < min_support))) return 2;
	return 3;
}

// filter_relative_support (source/filter_relative_support.cpp:209-224): true = discard
AGPU_HD bool fails_relative_support(const AnnotationView& ann, const CandidateTable& t, uint32_t c, float evalue, float evalue_cutoff) {
	bool keep = evalue < evalue_cutoff &&
	            !(candidate_is_intragenic(ann, t.gene1[c], t.gene2[c], t.breakpoint1[c], t.breakpoint2[c]) && t.split_reads1[c] + t.split_reads2[c] == 0);
	return !keep;
}

// partner dedup (source/filter_relative_support.cpp:20-29): every unfiltered candidate with gene1 != gene2 raises two events in
// iteration order, first for the key (gene2, breakpoints) contributing gene1, then for (gene1, breakpoints) contributing gene2; the
// first event of a key wins.
struct PartnerKey { uint32_t gene; int32_t breakpoint1, breakpoint2; };
AGPU_HD bool partner_keys_equal(const PartnerKey& a, const PartnerKey& b) { return a.gene == b.gene && a.breakpoint1 == b.breakpoint1 && a.breakpoint2 == b.breakpoint2; }
AGPU_HD uint64_t hash_partner_key(const PartnerKey& key) {
	uint64_t h = ((uint64_t) (uint32_t) key.breakpoint1 << 32 | (uint32_t) key.breakpoint2) * 0x9E3779B97F4A7C15ULL;
	h ^= (uint64_t) key.gene * 0xC2B2AE3D27D4EB4FULL;
	h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
	return h;
}
// event handle = 2 * candidate + which (0: key gene2 / partner gene1, 1: key gene1 / partner gene2)
AGPU_HD PartnerKey partner_event_key(const CandidateTable& t, uint32_t handle) {
	uint32_t c = handle >> 1;
	PartnerKey key; key.gene = (handle & 1) ? t.gene1[c] : t.gene2[c]; key.breakpoint1 = t.breakpoint1[c]; key.breakpoint2 = t.breakpoint2[c];
	return key;
}
AGPU_HD uint32_t partner_event_partner(const CandidateTable& t, uint32_t handle) { uint32_t c = handle >> 1; return (handle & 1) ? t.gene2[c] : t.gene1[c]; }
AGPU_HD bool raises_partner_events(const CandidateTable& t, uint32_t c) { return t.filter[c] == FILTER_none && t.gene1[c] != t.gene2[c]; }

}

#endif
