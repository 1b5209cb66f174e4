// arriba_amd/csrc/device/event_core.hpp -- event-level predicates behind filter_relative_support: pure functions of one
// candidate, its read lists, the annotation and coverage_t.
//   filter_both_intronic      source/filter_both_intronic.cpp:8-36
//   filter_short_anchor       source/filter_short_anchor.cpp:7-24
//   filter_end_to_end_fusions source/filter_end_to_end.cpp:8-78
//   filter_no_coverage        source/filter_no_coverage.cpp:9-103
//   filter_marginal_read_through source/filter_marginal_read_through.cpp:8-46
#ifndef AGPU_EVENT_CORE_HPP
#define AGPU_EVENT_CORE_HPP 1

#include "fusion_core.hpp"
#include "merge_core.hpp"

namespace agpu {

const uint8_t FILTER_intronic = 13, FILTER_end_to_end = 21, FILTER_marginal_read_through = 25, FILTER_short_anchor = 26, FILTER_no_coverage = 27; // source/common.hpp:29-67
const int32_t COVERAGE_RESOLUTION = 20; // source/read_stats.hpp:12

// coverage_t (source/read_stats.hpp:17-27) flattened: the windows of contig c are [window_offset[c], window_offset[c + 1])
struct CoverageView {
	uint32_t n_contigs;
	const uint64_t* window_offset;
	const uint16_t* coverage;
	const uint8_t* fragment_starts;
	const uint8_t* fragment_ends;
};

// reference: coverage_t::fragment_starts_here / fragment_ends_here, source/read_stats.cpp:269-292
AGPU_HD bool fragment_starts_here(const CoverageView& coverage, uint32_t contig, int32_t start, int32_t end) {
	if (contig >= coverage.n_contigs) return false;
	const uint64_t begin = coverage.window_offset[contig], size = coverage.window_offset[contig + 1] - begin;
	for (int32_t window = start / COVERAGE_RESOLUTION + 1; window <= end / COVERAGE_RESOLUTION; ++window) {
		if ((uint64_t) (uint32_t) window >= size) return false;
		if (coverage.fragment_starts[begin + (uint32_t) window]) return true;
	}
	return false;
}
AGPU_HD bool fragment_ends_here(const CoverageView& coverage, uint32_t contig, int32_t start, int32_t end) {
	if (contig >= coverage.n_contigs) return false;
	const uint64_t begin = coverage.window_offset[contig], size = coverage.window_offset[contig + 1] - begin;
	for (int32_t window = start / COVERAGE_RESOLUTION; window < end / COVERAGE_RESOLUTION; ++window) {
		if ((uint64_t) (uint32_t) window >= size) return false;
		if (coverage.fragment_ends[begin + (uint32_t) window]) return true;
	}
	return false;
}

// reference: coverage_t::get_coverage, source/read_stats.cpp:295-306: the window before (upstream) / behind (downstream) the position
AGPU_HD int32_t coverage_near(const CoverageView& coverage, uint32_t contig, int32_t position, bool upstream) {
	if (contig >= coverage.n_contigs) return -1;
	const uint64_t begin = coverage.window_offset[contig], size = coverage.window_offset[contig + 1] - begin;
	if (size == 0) return -1;
	if (upstream) return position < COVERAGE_RESOLUTION ? 0 : coverage.coverage[begin + (uint64_t) (position / COVERAGE_RESOLUTION - 1)];
	return coverage.coverage[begin + (uint64_t) (position / COVERAGE_RESOLUTION + 1)];
}

AGPU_HD bool candidate_is_read_through(const CandidateTable& t, uint32_t c) { // source/common.hpp:265-269
	const uint32_t flags = t.flags[c];
	return (t.contigs[c] >> 16) == (t.contigs[c] & 0xFFFF) && t.breakpoint2[c] - t.breakpoint1[c] < 400000 && !(flags & CFLAG_UPSTREAM1) && (flags & CFLAG_UPSTREAM2);
}
AGPU_HD bool candidate_overlaps_both_genes(const AnnotationView& ann, const CandidateTable& t, uint32_t c) { // source/common.hpp:260-264
	const uint32_t gene1 = t.gene1[c], gene2 = t.gene2[c];
	return (t.breakpoint1[c] >= ann.gene_start[gene2] && t.breakpoint1[c] <= ann.gene_end[gene2]) || (t.breakpoint2[c] >= ann.gene_start[gene1] && t.breakpoint2[c] <= ann.gene_end[gene1]);
}
AGPU_HD bool contig_is_viral(const GenomeView& genome, uint32_t contig) { return contig < genome.n_contigs && (genome.contig_bits[contig] & CBIT_VIRAL); }

// Who walks the read lists of a candidate.  The predicates below that do are sums over list entries followed by a decision: one thread takes every entry (ListLanes()), or the 64
// lanes of a wavefront take every 64th and add their counts up (agpu_events.hip: the candidates with long lists -- a list holds up to -U reads, 32 767 in config 3 -- get a
// wavefront each; a thread walking 98 000 entries alone was the whole time of these kernels).  sum() must give every lane the total.
struct ListLanes {
	uint32_t lane = 0, lanes = 1;
	AGPU_HD uint32_t sum(uint32_t mine) const { return mine; }
	AGPU_HD bool any(bool mine) const { return mine; }
};

// ---- filter_both_intronic: true = no unfiltered read of the candidate has an exonic alignment
template <class Lanes = ListLanes> AGPU_HD bool has_only_intronic_reads(const BatchView& b, const GenomeView& genome, const CandidateTable& t, uint32_t c, const Lanes& lanes = Lanes()) {
	if (contig_is_viral(genome, t.contigs[c] >> 16) || contig_is_viral(genome, t.contigs[c] & 0xFFFF)) return false; // viral contigs are often not annotated
	const uint64_t end = t.list_offset[3 * (uint64_t) c + 3];
	for (uint64_t base = t.list_offset[3 * (uint64_t) c]; base < end; base += lanes.lanes) { // (all lanes make the same number of turns: any() asks every one of them)
		bool exonic = false;
		const uint64_t k = base + lanes.lane;
		if (k < end) {
			const uint32_t read = t.read_lists[k];
			if (b.walk != nullptr) { const uint8_t walk = b.walk[read]; exonic = (walk & WALK_UNFILTERED) && (walk & WALK_EXONIC); }
			else if (b.filter[read] == FILTER_none)
				for (int slot = 0; slot < b.n_aln[read]; ++slot)
					if (b.abits[slot][read] & ABIT_EXONIC) exonic = true;
		}
		if (lanes.any(exonic)) return false; // (the reference stops at the first one, too)
	}
	return true;
}

// ---- filter_short_anchor
AGPU_HD bool has_short_anchor(const CandidateTable& t, uint32_t c, uint32_t min_length) {
	const uint32_t flags = t.flags[c];
	if ((flags & CFLAG_SPLICED1) && (flags & CFLAG_SPLICED2)) return false;
	int32_t distance1 = t.anchor1[c] - t.breakpoint1[c]; if (distance1 < 0) distance1 = -distance1;
	int32_t distance2 = t.anchor2[c] - t.breakpoint2[c]; if (distance2 < 0) distance2 = -distance2;
	return (uint32_t) distance1 < min_length || (uint32_t) distance2 < min_length;
}

// ---- filter_end_to_end_fusions
// reference: calculate_intronic_fraction (:9-26): bases of the gene not covered by the exon found first in every boundary bucket
AGPU_HD float intronic_fraction(const AnnotationView& ann, uint32_t gene) {
	AGPU_FP_AS_WRITTEN
	const FlatIndexView& index = ann.exon_index;
	const uint32_t contig = ann.gene_contig[gene];
	const int32_t gene_start = ann.gene_start[gene], gene_end = ann.gene_end[gene];
	uint32_t intronic_bases = 0;
	int32_t previous_position = gene_start;
	if (contig < index.n_contigs) {
		const uint32_t contig_end = index.contig_offset[contig + 1];
		for (uint32_t k = index_lower_bound(index, contig, gene_start); k != contig_end && index.keys[k] <= gene_end; ++k) {
			const ListRef exons = index_bucket(index, k);
			for (uint32_t m = 0; m < exons.n; ++m) {
				const uint32_t exon = exons.p[m];
				if (ann.exon_gene[exon] != gene) continue;
				if (previous_position < ann.exon_start[exon]) intronic_bases += (uint32_t) (ann.exon_start[exon] - previous_position);
				if (previous_position < ann.exon_end[exon]) previous_position = ann.exon_end[exon] + 1;
				break;
			}
		}
	}
	return ((float) intronic_bases) / (float) (gene_end - gene_start + 1);
}
AGPU_HD bool gene_is_fused_end_to_end(const AnnotationView& ann, uint32_t gene, bool upstream) { // the gene's start of transcription points away from the breakpoint
	const uint8_t bits = ann.gene_bits[gene];
	return (bits & GBIT_DUMMY) || (((bits & GBIT_STRAND) != 0) == upstream);
}
AGPU_HD bool is_end_to_end_fusion(const AnnotationView& ann, const GenomeView& genome, const CandidateTable& t, uint32_t c) {
	if (contig_is_viral(genome, t.contigs[c] >> 16) || contig_is_viral(genome, t.contigs[c] & 0xFFFF)) return false;
	const uint32_t flags = t.flags[c], gene1 = t.gene1[c], gene2 = t.gene2[c];
	if (!candidate_is_read_through(t, c) && gene1 != gene2 && (flags & (CFLAG_SPLICED1 | CFLAG_SPLICED2))) return false; // spliced breakpoints are likely true
	const uint32_t split_reads1 = t.split_reads1[c], split_reads2 = t.split_reads2[c], discordant_mates = t.discordant_mates[c];
	if (!(discordant_mates + split_reads1 == 0 || discordant_mates + split_reads2 == 0 || split_reads1 + split_reads2 == 0 ||
	      (candidate_overlaps_both_genes(ann, t, c) && (split_reads1 == 0 || split_reads2 == 0)))) return false; // only breakpoints with low support
	if (!gene_is_fused_end_to_end(ann, gene1, flags & CFLAG_UPSTREAM1) || !gene_is_fused_end_to_end(ann, gene2, flags & CFLAG_UPSTREAM2)) return false;
	const uint32_t many_discordant_mates = 10;
	const int32_t min_breakpoint_distance = 1000000;
	const float max_intronic_fraction = 0.66f;
	if (discordant_mates < many_discordant_mates) return true;
	int32_t distance = t.breakpoint1[c] - t.breakpoint2[c]; if (distance < 0) distance = -distance;
	if ((t.contigs[c] >> 16) == (t.contigs[c] & 0xFFFF) && distance < min_breakpoint_distance) return true;
	return (flags & CFLAG_EXONIC1) && (flags & CFLAG_EXONIC2) && intronic_fraction(ann, gene1) > max_intronic_fraction && intronic_fraction(ann, gene2) > max_intronic_fraction;
}

// ---- filter_no_coverage
AGPU_HD bool breakpoint_in_terminal_exon(const AnnotationView& ann, uint32_t contig, int32_t breakpoint, uint32_t gene) {
	const FlatIndexView& index = ann.exon_index;
	if (contig >= index.n_contigs) return false;
	const uint32_t k = index_lower_bound(index, contig, breakpoint); // get_annotation_by_coordinate(contig, bp, bp): the bucket of the first boundary >= bp
	if (k == index.contig_offset[contig + 1]) return false;
	const ListRef exons = index_bucket(index, k);
	for (uint32_t m = 0; m < exons.n; ++m) {
		const uint32_t exon = exons.p[m];
		if (ann.exon_gene[exon] == gene && (ann.exon_previous[exon] == -1 || ann.exon_next[exon] == -1)) return true;
	}
	return false;
}
AGPU_HD bool breakpoint_lacks_coverage(const AnnotationView& ann, const CoverageView& coverage, uint32_t contig, int32_t breakpoint, int32_t anchor_start, bool upstream, uint32_t gene, bool no_split_reads) {
	const int32_t scan_range = 200;
	if (breakpoint_in_terminal_exon(ann, contig, breakpoint, gene)) return false;
	int32_t start, end;
	if (upstream) {
		start = breakpoint;
		if (no_split_reads) start -= scan_range;
		end = breakpoint + scan_range > anchor_start ? breakpoint + scan_range : anchor_start;
		return !fragment_starts_here(coverage, contig, start, end);
	}
	start = breakpoint - scan_range < anchor_start ? breakpoint - scan_range : anchor_start;
	end = breakpoint;
	if (no_split_reads) end += scan_range;
	return !fragment_ends_here(coverage, contig, start, end);
}
AGPU_HD bool has_no_coverage(const AnnotationView& ann, const CoverageView& coverage, const CandidateTable& t, uint32_t c) {
	const uint32_t flags = t.flags[c];
	const uint32_t split_reads1 = t.split_reads1[c], split_reads2 = t.split_reads2[c], discordant_mates = t.discordant_mates[c];
	if (!candidate_is_read_through(t, c)) {
		if (split_reads1 + split_reads2 != 0 && split_reads1 + discordant_mates != 0 && split_reads2 + discordant_mates != 0) return false; // high support
		if (flags & (CFLAG_SPLICED1 | CFLAG_SPLICED2)) return false; // spliced breakpoints are more credible
	} else if ((flags & CFLAG_SPLICED1) && (flags & CFLAG_SPLICED2)) {
		return false;
	}
	const bool no_split_reads = split_reads1 + split_reads2 == 0;
	if (breakpoint_lacks_coverage(ann, coverage, t.contigs[c] >> 16, t.breakpoint1[c], t.anchor1[c], flags & CFLAG_UPSTREAM1, t.gene1[c], no_split_reads)) return true;
	return breakpoint_lacks_coverage(ann, coverage, t.contigs[c] & 0xFFFF, t.breakpoint2[c], t.anchor2[c], flags & CFLAG_UPSTREAM2, t.gene2[c], no_split_reads);
}

// ---- filter_marginal_read_through: read-through events whose breakpoints sit in the last percent of donor and acceptor and whose reads are a
// small fraction of the coverage.  The arithmetic types are the reference's: double positions, float margin and allele fraction.
AGPU_HD bool is_marginal_read_through(const AnnotationView& ann, const CoverageView& coverage, const CandidateTable& t, uint32_t c) {
	AGPU_FP_AS_WRITTEN
	if (!candidate_is_read_through(t, c)) return false;
	const float margin = 0.01f, min_vaf = 0.07f;
	const uint32_t flags = t.flags[c], gene1 = t.gene1[c], gene2 = t.gene2[c];
	const bool upstream1 = flags & CFLAG_UPSTREAM1, upstream2 = flags & CFLAG_UPSTREAM2;
	const uint8_t bits1 = ann.gene_bits[gene1], bits2 = ann.gene_bits[gene2];
	const bool dummy1 = bits1 & GBIT_DUMMY, dummy2 = bits2 & GBIT_DUMMY, forward1 = bits1 & GBIT_STRAND, forward2 = bits2 & GBIT_STRAND;
	const int32_t breakpoint1 = t.breakpoint1[c], breakpoint2 = t.breakpoint2[c];
	double position_in_donor = 1, position_in_acceptor = 1;
	if (!dummy1 && forward1 && !upstream1) position_in_donor = 1.0 * (breakpoint1 - ann.gene_start[gene1]) / (ann.gene_end[gene1] - ann.gene_start[gene1]);
	else if (!dummy2 && !forward2 && upstream2) position_in_donor = 1.0 * (ann.gene_end[gene2] - breakpoint2) / (ann.gene_end[gene2] - ann.gene_start[gene2]);
	else if (!dummy1 && !forward1 && !upstream1) position_in_acceptor = 1.0 * (breakpoint1 - ann.gene_start[gene1]) / (ann.gene_end[gene1] - ann.gene_start[gene1]);
	else if (!dummy2 && forward2 && upstream2) position_in_acceptor = 1.0 * (ann.gene_end[gene2] - breakpoint2) / (ann.gene_end[gene2] - ann.gene_start[gene2]);
	else return false; // both breakpoints are intergenic
	const int32_t coverage1 = coverage_near(coverage, t.contigs[c] >> 16, breakpoint1, !upstream1), coverage2 = coverage_near(coverage, t.contigs[c] & 0xFFFF, breakpoint2, !upstream2);
	const uint32_t supporting_reads = t.split_reads1[c] + t.split_reads2[c] + t.discordant_mates[c];
	const float one_minus_margin = 1 - margin;
	return position_in_donor > one_minus_margin && position_in_acceptor > one_minus_margin && (float) supporting_reads < min_vaf * (float) (coverage1 > coverage2 ? coverage1 : coverage2);
}

// ---- select_most_supported_breakpoints (source/select_best.cpp:8-80): of the unfiltered candidates of one gene pair and direction pair the
// "best" one stays.  The reference folds the candidates in the iteration order of fusions_t (hazard H2) with a rule that is not antisymmetric
// (the exonic clause), so the result depends on that order: the device sorts the unfiltered candidates by (gene pair + directions, iteration
// rank) and one thread replays the fold over each group.
const uint8_t FILTER_select_best = 24;
AGPU_HD uint32_t select_best_rank(const CandidateTable& t, uint32_t c) { // reference: rank_fusion (:8-19)
	const bool split1 = t.split_reads1[c] != 0, split2 = t.split_reads2[c] != 0, discordant = t.discordant_mates[c] != 0;
	if (split1 && split2) return 3;
	if ((split1 || split2) && discordant) return 2;
	if (split1 || split2) return 1;
	return 0;
}
// does `fusion` replace `best` as the best breakpoint of their gene pair? (:33-60)
AGPU_HD bool select_best_replaces(const CandidateTable& t, uint32_t fusion, uint32_t best) {
	const uint32_t rank_fusion = select_best_rank(t, fusion), rank_best = select_best_rank(t, best);
	if (rank_fusion != rank_best) return rank_fusion > rank_best;
	const uint32_t support_fusion = t.split_reads1[fusion] + t.split_reads2[fusion] + t.discordant_mates[fusion], support_best = t.split_reads1[best] + t.split_reads2[best] + t.discordant_mates[best];
	if (support_fusion != support_best) return support_fusion > support_best;
	const uint32_t flags = t.flags[fusion], flags_best = t.flags[best];
	const bool exonic1 = flags & CFLAG_EXONIC1, exonic2 = flags & CFLAG_EXONIC2, best_exonic1 = flags_best & CFLAG_EXONIC1, best_exonic2 = flags_best & CFLAG_EXONIC2;
	if ((exonic1 && !best_exonic1) || (exonic2 && !best_exonic2)) return true;
	if (!((!best_exonic1 || exonic1 == best_exonic1) && (!best_exonic2 || exonic2 == best_exonic2))) return false;
	const bool upstream1 = flags & CFLAG_UPSTREAM1, upstream2 = flags & CFLAG_UPSTREAM2; // the same for both: the directions are part of the group
	if (upstream1 ? t.breakpoint1[fusion] < t.breakpoint1[best] : t.breakpoint1[fusion] > t.breakpoint1[best]) return true;
	if (t.breakpoint1[fusion] != t.breakpoint1[best]) return false;
	return upstream2 ? t.breakpoint2[fusion] < t.breakpoint2[best] : t.breakpoint2[fusion] > t.breakpoint2[best];
}
AGPU_HD uint64_t select_best_group_key(const CandidateTable& t, uint32_t c) { // filtered candidates sort behind all groups
	const uint32_t flags = t.flags[c];
	return t.filter[c] == FILTER_none ? (uint64_t) t.gene1[c] << 33 | (uint64_t) t.gene2[c] << 2 | ((flags & CFLAG_UPSTREAM1) ? 1u : 0u) | ((flags & CFLAG_UPSTREAM2) ? 2u : 0u) : ~0ull;
}
// the group order[begin .. end) is in iteration order; everything but the best gets the filter
AGPU_HD void select_best_in_group(const CandidateTable& t, const uint32_t* order, uint32_t begin, uint32_t end) {
	uint32_t best = order[begin];
	for (uint32_t j = begin + 1; j < end; ++j)
		if (select_best_replaces(t, order[j], best)) best = order[j];
	for (uint32_t j = begin; j < end; ++j)
		if (order[j] != best) t.filter[order[j]] = FILTER_select_best;
}

// ---- recover_many_spliced (source/recover_many_spliced.cpp:8-51): a gene pair with >= min_spliced_events distinct spliced breakpoint pairs (binned
// to 10 bp) gets its spliced candidates back that were discarded by inconsistently_clipped, relative_support, min_support or select_best.  Every
// candidate that can be recovered also counts towards the events of its pair, so one sorted list serves both passes of the reference.
AGPU_HD bool many_spliced_takes_part(const AnnotationView& ann, const CandidateTable& t, uint32_t c) {
	const uint8_t filter = t.filter[c];
	return !candidate_is_read_through(t, c) && (t.flags[c] & (CFLAG_SPLICED1 | CFLAG_SPLICED2)) && t.gene1[c] != t.gene2[c] && !candidate_overlaps_both_genes(ann, t, c) &&
	       (filter == FILTER_none || filter == FILTER_inconsistently_clipped || filter == FILTER_relative_support || filter == 17 /* min_support */ || filter == FILTER_select_best);
}
AGPU_HD uint64_t many_spliced_pair_key(const AnnotationView& ann, const CandidateTable& t, uint32_t c) { return many_spliced_takes_part(ann, t, c) ? (uint64_t) t.gene1[c] << 32 | t.gene2[c] : ~0ull; }
AGPU_HD uint64_t many_spliced_bin_key(const CandidateTable& t, uint32_t c) { return (uint64_t) (uint32_t) (t.breakpoint1[c] / 10) << 32 | (uint32_t) (t.breakpoint2[c] / 10); }
// the gene pair order[begin .. end), sorted by bin key
AGPU_HD void recover_many_spliced_in_pair(const CandidateTable& t, const uint32_t* order, uint32_t begin, uint32_t end, uint32_t min_spliced_events) {
	uint32_t events = 1;
	for (uint32_t j = begin + 1; j < end; ++j)
		if (many_spliced_bin_key(t, order[j]) != many_spliced_bin_key(t, order[j - 1])) ++events;
	if (events < min_spliced_events) return;
	for (uint32_t j = begin; j < end; ++j) t.filter[order[j]] = FILTER_none;
}

// ---- filter_in_vitro (source/filter_in_vitro.cpp:17-228): events with the characteristics of fusions made during reverse transcription -- few split
// reads, partners expressed in the top quantile (chimeric reads per gene as the proxy), breakpoints inside exons.
const uint8_t FILTER_in_vitro = 22;
// What filter_in_vitro asks of every alignment of every discordant mate of every candidate (:133-158): is it clipped by >= 3 bases at its far end, and where does that
// end lie?  Answered once per alignment (8 bytes) instead of once per list entry from the CIGAR, strand, contig, start and end columns (the lists hold every
// fragment ~40 times: at 10^7 fragments the kernel moved 218 GB).
struct ClipSummary { int32_t position; uint16_t contig; uint16_t clipped; }; // clipped: 1 = the alignment is clipped by >= 3 bases at the end `position` (forward: its end, reverse: its start)
const uint32_t CLIP_SUMMARIES_PER_READ = 4; // three alignments and a word of padding: the summaries of a read are one 32-byte piece of one line
AGPU_HD ClipSummary clip_summary_of(const BatchView& b, uint64_t read, int slot) {
	ClipSummary summary = { 0, 0, 0 };
	if (slot >= (int) b.n_aln[read] || b.filter[read] != FILTER_none) return summary; // (is_in_vitro_artifact looks at the reads no filter discarded: source/filter_in_vitro.cpp:133-158)
	const uint32_t min_clipped_length = 3;
	const uint32_t* cigar = cigar_of(b, slot, read); const uint32_t n_cigar = b.cigar_count[slot][read];
	const bool forward = b.abits[slot][read] & ABIT_STRAND;
	if (forward && postclipping(cigar, n_cigar) >= min_clipped_length) { summary.position = b.end[slot][read]; summary.contig = b.contig[slot][read]; summary.clipped = 1; }
	else if (!forward && preclipping(cigar, n_cigar) >= min_clipped_length) { summary.position = b.start[slot][read]; summary.contig = b.contig[slot][read]; summary.clipped = 1; }
	return summary;
}
struct InVitroTables {
	const uint32_t* gene_read_count;       // chimeric fragments per gene (find_top_expressed_genes, :49-58), GTF genes and dummy genes
	uint32_t high_expression_threshold;    // the quantile of the non-zero counts (:60-80)
	const uint64_t* pair_keys; const uint32_t* pair_counts; uint32_t n_pairs; // exonic_breakpoints_by_gene_pair (:93-105): sorted keys gene1 << 32 | gene2
	const ClipSummary* clip_summaries;     // [3 * fragments] or null: then every list entry reads the columns of the batch
	const uint8_t* clip_any;               // with the summaries, or null: [fragments] 1 = one of the read's summaries is a clipped end.  One byte per list entry decides whether the 32
	                                       // bytes of its summaries are looked at at all (round 6: few discordant mates are clipped; the bytes of 10^8 reads stay in the last-level cache)
};
// genes of a fragment that count towards the expression proxy: those of MATE1 and of MATE2 (discordant mates) or SUPPLEMENTARY (split read) (:52-57)
AGPU_HD int in_vitro_second_slot(const BatchView& b, uint64_t i) { return b.n_aln[i] == 2 ? MATE2 : SUPPLEMENTARY; }
AGPU_HD bool counts_as_exonic_breakpoint(const CandidateTable& t, uint32_t c) { // :96-101
	const uint32_t flags = t.flags[c];
	return t.gene1[c] != t.gene2[c] && !(flags & (CFLAG_SPLICED1 | CFLAG_SPLICED2)) && (flags & CFLAG_EXONIC1) && (flags & CFLAG_EXONIC2) &&
	       t.list_offset[3 * (uint64_t) c + 2] - t.list_offset[3 * (uint64_t) c] > 0 && t.filter[c] != 23 /* merge_adjacent */ && t.filter[c] != FILTER_uninteresting_contigs;
}
AGPU_HD uint32_t exonic_breakpoints_of_pair(const InVitroTables& tables, uint32_t gene1, uint32_t gene2) {
	const uint64_t key = (uint64_t) gene1 << 32 | gene2;
	const uint32_t at = lower_bound_u64(tables.pair_keys, tables.n_pairs, key);
	return (at < tables.n_pairs && tables.pair_keys[at] == key) ? tables.pair_counts[at] : 0;
}
// reference: find_higher_expressed_gene (:17-29): of the genes overlapping the breakpoint (dummy genes included) the first with a strictly higher count
AGPU_HD uint32_t higher_expressed_gene(const AnnotationView& ann, const InVitroTables& tables, uint32_t contig, int32_t breakpoint, uint32_t gene, GeneQuery& overlapping) {
	uint32_t highest_expression = tables.gene_read_count[gene];
	query_point_with_dummy_genes(ann, contig, breakpoint, overlapping);
	for (uint32_t k = 0; k < overlapping.size(); ++k) {
		const uint32_t other = overlapping.element(ann, k);
		if (tables.gene_read_count[other] > highest_expression) { highest_expression = tables.gene_read_count[other]; gene = other; }
	}
	return gene;
}
// the candidates filter_in_vitro judges: also filtered events are tagged if they are spliced, so that the filters 'spliced' and 'many_spliced' do not recover them (:112-115)
AGPU_HD bool in_vitro_looks_at(const CandidateTable& t, uint32_t c) {
	const uint32_t flags = t.flags[c];
	const uint8_t filter = t.filter[c];
	return filter == FILTER_none || ((flags & (CFLAG_SPLICED1 | CFLAG_SPLICED2)) && (filter == FILTER_relative_support || filter == 17 /* min_support */ || filter == FILTER_homopolymer));
}
// the walk of the verdict: the discordant mates of the candidate that are clipped right at one of its breakpoints (:133-158).  Reads outside [b.first_rank, b.first_rank + b.n) are
// skipped: a context that holds one shard of the sample counts its own reads (agpu_in_vitro_clipped_mates), the counts of all shards add up to those of the sample.
template <class Lanes = ListLanes> AGPU_HD void in_vitro_clipped_mates(const BatchView& b, const InVitroTables& tables, const CandidateTable& t, uint32_t c, uint32_t& clipped_discordant_mates1, uint32_t& clipped_discordant_mates2, const Lanes& lanes = Lanes()) {
	const uint32_t contig1 = t.contigs[c] >> 16, contig2 = t.contigs[c] & 0xFFFF;
	const int32_t breakpoint1 = t.breakpoint1[c], breakpoint2 = t.breakpoint2[c];
	const uint32_t min_clipped_length = 3;
	clipped_discordant_mates1 = 0; clipped_discordant_mates2 = 0;
	for (uint64_t k = t.list_offset[3 * (uint64_t) c + 2] + lanes.lane; k < t.list_offset[3 * (uint64_t) c + 3]; k += lanes.lanes) {
		const uint64_t read = (uint64_t) t.read_lists[k] - b.first_rank;
		if (read >= b.n) continue; // (a read of another shard; unsigned: also those in front of this one)
		if (tables.clip_summaries != nullptr) { // (a read that a filter discarded has no clipped end in its summaries: clip_summary_of; 32 bytes per read, one line per list entry)
			if (tables.clip_any != nullptr && !tables.clip_any[read]) continue;
			for (int slot = 0; slot < 3; ++slot) {
				const ClipSummary summary = tables.clip_summaries[CLIP_SUMMARIES_PER_READ * read + slot];
				if (!summary.clipped) continue;
				if (summary.contig == contig1 && summary.position == breakpoint1) clipped_discordant_mates1++;
				else if (summary.contig == contig2 && summary.position == breakpoint2) clipped_discordant_mates2++;
			}
			continue;
		}
		if (b.filter[read] != FILTER_none) continue;
		for (int slot = 0; slot < b.n_aln[read]; ++slot) {
			const uint32_t* cigar = cigar_of(b, slot, read); const uint32_t n_cigar = b.cigar_count[slot][read];
			const bool forward = b.abits[slot][read] & ABIT_STRAND;
			if (forward && postclipping(cigar, n_cigar) >= min_clipped_length) {
				if (b.contig[slot][read] == contig1 && b.end[slot][read] == breakpoint1) clipped_discordant_mates1++;
				else if (b.contig[slot][read] == contig2 && b.end[slot][read] == breakpoint2) clipped_discordant_mates2++;
			} else if (!forward && preclipping(cigar, n_cigar) >= min_clipped_length) {
				if (b.contig[slot][read] == contig1 && b.start[slot][read] == breakpoint1) clipped_discordant_mates1++;
				else if (b.contig[slot][read] == contig2 && b.start[slot][read] == breakpoint2) clipped_discordant_mates2++;
			}
		}
	}
	clipped_discordant_mates1 = lanes.sum(clipped_discordant_mates1); clipped_discordant_mates2 = lanes.sum(clipped_discordant_mates2);
}
// ... and the decision, from the counts of the walk
AGPU_HD bool in_vitro_verdict(const AnnotationView& ann, const CoverageView& coverage, const InVitroTables& tables, const CandidateTable& t, uint32_t c, uint32_t clipped_discordant_mates1, uint32_t clipped_discordant_mates2) {
	AGPU_FP_AS_WRITTEN
	const uint32_t flags = t.flags[c];
	const bool spliced1 = flags & CFLAG_SPLICED1, spliced2 = flags & CFLAG_SPLICED2, exonic1 = flags & CFLAG_EXONIC1, exonic2 = flags & CFLAG_EXONIC2;
	float potential_rt_breakpoints = 0;
	if (!exonic1) potential_rt_breakpoints += 0.5; else if (!spliced1) potential_rt_breakpoints += 1;
	if (!exonic2) potential_rt_breakpoints += 0.5; else if (!spliced2) potential_rt_breakpoints += 1;
	const uint32_t contig1 = t.contigs[c] >> 16, contig2 = t.contigs[c] & 0xFFFF;
	const int32_t breakpoint1 = t.breakpoint1[c], breakpoint2 = t.breakpoint2[c];
	const uint32_t split_reads1 = t.split_reads1[c], split_reads2 = t.split_reads2[c], discordant_mates = t.discordant_mates[c];
	const uint32_t total_split_reads = (clipped_discordant_mates1 < clipped_discordant_mates2 ? clipped_discordant_mates1 : clipped_discordant_mates2) + split_reads1 + split_reads2;
	IdSetTail overlapping_tail; GeneQuery overlapping(overlapping_tail.words);
	const uint32_t gene1 = higher_expressed_gene(ann, tables, contig1, breakpoint1, t.gene1[c], overlapping);
	const uint32_t gene2 = higher_expressed_gene(ann, tables, contig2, breakpoint2, t.gene2[c], overlapping);
	const uint32_t gene1_expression = tables.gene_read_count[gene1], gene2_expression = tables.gene_read_count[gene2];
	const uint32_t own_pair = exonic_breakpoints_of_pair(tables, t.gene1[c], t.gene2[c]), expressed_pair = exonic_breakpoints_of_pair(tables, gene1, gene2);
	const uint32_t exonic_breakpoints = expressed_pair > own_pair ? expressed_pair : own_pair;
	const int32_t coverage1 = coverage_near(coverage, contig1, breakpoint1, !(flags & CFLAG_UPSTREAM1)), coverage2 = coverage_near(coverage, contig2, breakpoint2, !(flags & CFLAG_UPSTREAM2));
	const uint32_t supporting_reads = split_reads1 + split_reads2 + discordant_mates;
	const uint32_t threshold = tables.high_expression_threshold, max_exonic_breakpoints_by_gene_pair = 8;
	return (double) total_split_reads <= 2 + 0.0001 * (double) (gene1_expression + gene2_expression) &&
	       (total_split_reads * 2 <= discordant_mates || total_split_reads <= 2) &&
	       gene1_expression + gene2_expression > threshold &&
	       !(supporting_reads >= 10 && ((int32_t) supporting_reads * 4) >= (coverage1 > coverage2 ? coverage1 : coverage2) && coverage1 > (int32_t) supporting_reads && coverage2 > (int32_t) supporting_reads &&
	         (spliced1 || spliced2) && ((spliced1 || !exonic1) && (spliced2 || !exonic2))) &&
	       (potential_rt_breakpoints > 1 ||
	        (potential_rt_breakpoints > 0 && (gene1_expression > threshold || gene2_expression > threshold)) ||
	        gene1_expression > 2 * threshold || gene2_expression > 2 * threshold || (gene1_expression > threshold && gene2_expression > threshold) ||
	        exonic_breakpoints > max_exonic_breakpoints_by_gene_pair ||
	        supporting_reads <= 1);
}
template <class Lanes = ListLanes> AGPU_HD bool is_in_vitro_artifact(const BatchView& b, const AnnotationView& ann, const CoverageView& coverage, const InVitroTables& tables, const CandidateTable& t, uint32_t c, const Lanes& lanes = Lanes()) {
	if (!in_vitro_looks_at(t, c)) return false;
	uint32_t clipped_discordant_mates1, clipped_discordant_mates2;
	in_vitro_clipped_mates(b, tables, t, c, clipped_discordant_mates1, clipped_discordant_mates2, lanes);
	return in_vitro_verdict(ann, coverage, tables, t, c, clipped_discordant_mates1, clipped_discordant_mates2);
}

// ---- recover_both_spliced (source/recover_both_spliced.cpp:13-182): candidates with two spliced breakpoints that were discarded for low support come
// back when the reads of all candidates of their gene pair (same orientation, or the reciprocal orientation) add up to >= 2.
AGPU_HD bool both_breakpoints_spliced(const AnnotationView& ann, const CandidateTable& t, uint32_t c) { // source/common.hpp:280-284
	const uint32_t flags = t.flags[c];
	if (!(flags & CFLAG_SPLICED1) || !(flags & CFLAG_SPLICED2)) return false;
	const bool same_strand = ((ann.gene_bits[t.gene1[c]] ^ ann.gene_bits[t.gene2[c]]) & GBIT_STRAND) == 0, same_direction = ((flags & CFLAG_UPSTREAM1) != 0) == ((flags & CFLAG_UPSTREAM2) != 0);
	return same_strand ? !same_direction : same_direction;
}
AGPU_HD bool breakpoint_in_large_exon(const AnnotationView& ann, uint32_t contig, int32_t breakpoint, int32_t max_exon_size) {
	const FlatIndexView& index = ann.exon_index;
	if (contig >= index.n_contigs) return false;
	const uint32_t k = index_lower_bound(index, contig, breakpoint);
	if (k == index.contig_offset[contig + 1]) return false;
	const ListRef exons = index_bucket(index, k);
	for (uint32_t m = 0; m < exons.n; ++m)
		if (ann.exon_end[exons.p[m]] + 1 - ann.exon_start[exons.p[m]] > max_exon_size) return true;
	return false;
}
// reference: count_supporting_reads (:13-70)
template <class Lanes = ListLanes> AGPU_HD uint32_t both_spliced_supporting_reads(const BatchView& b, const AnnotationView& ann, const CoverageView& coverage, const uint32_t* gene_read_count, uint32_t high_expression_threshold,
                                               const CandidateTable& t, uint32_t c, int32_t max_exon_size, uint32_t max_coverage, const Lanes& lanes = Lanes()) {
	const bool both_spliced = both_breakpoints_spliced(ann, t, c);
	const uint32_t split_reads1 = t.split_reads1[c], split_reads2 = t.split_reads2[c], discordant_mates = t.discordant_mates[c];
	if (gene_read_count[t.gene1[c]] > high_expression_threshold || gene_read_count[t.gene2[c]] > high_expression_threshold)
		return (both_spliced && discordant_mates <= split_reads1 + split_reads2) ? 1 : 0; // highly expressed partner: count the event as one read at most
	if (!both_spliced) {
		const uint32_t flags = t.flags[c];
		const uint32_t coverage1 = (uint32_t) coverage_near(coverage, t.contigs[c] >> 16, t.breakpoint1[c], !(flags & CFLAG_UPSTREAM1)); // unsigned as in the reference: -1 wraps
		const uint32_t coverage2 = (uint32_t) coverage_near(coverage, t.contigs[c] & 0xFFFF, t.breakpoint2[c], !(flags & CFLAG_UPSTREAM2));
		if (coverage1 + coverage2 > (split_reads1 + split_reads2 + discordant_mates) * max_coverage) return 0;
		if (breakpoint_in_large_exon(ann, t.contigs[c] >> 16, t.breakpoint1[c], max_exon_size) || breakpoint_in_large_exon(ann, t.contigs[c] & 0xFFFF, t.breakpoint2[c], max_exon_size)) return 0;
	}
	uint32_t multimappers = 0, unique_mappers = 0;
	const uint64_t begin = t.list_offset[3 * (uint64_t) c], end = t.list_offset[3 * (uint64_t) c + 3];
	for (uint64_t k = begin + lanes.lane; k < end; k += lanes.lanes) {
		const uint32_t read = t.read_lists[k];
		if (b.walk != nullptr) { const uint8_t walk = b.walk[read]; if (walk & WALK_MULTIMAPPER) multimappers++; else if (walk & WALK_UNFILTERED) unique_mappers++; }
		else if (b.fbits[read] & FBIT_MULTIMAPPER) multimappers++; else if (b.filter[read] == FILTER_none) unique_mappers++;
	}
	multimappers = lanes.sum(multimappers); unique_mappers = lanes.sum(unique_mappers);
	if ((double) multimappers >= 0.5 * (double) (end - begin)) return 0;
	return unique_mappers == 0 ? 1 : unique_mappers;
}
// candidates whose reads count for their gene pair (:78-87); the key carries the orientation.  Intragenic groups are never looked up.
AGPU_HD bool both_spliced_is_member(const AnnotationView& ann, const CandidateTable& t, uint32_t c) {
	const uint8_t filter = t.filter[c];
	return t.gene1[c] != t.gene2[c] && filter != 23 /* merge_adjacent */ &&
	       (filter == FILTER_none || filter == FILTER_in_vitro || filter == FILTER_intronic || filter == FILTER_relative_support || filter == 17 /* min_support */ ||
	        (filter == FILTER_inconsistently_clipped && both_breakpoints_spliced(ann, t, c)));
}
AGPU_HD uint64_t both_spliced_group_key(const CandidateTable& t, uint32_t c, bool reciprocal) {
	uint32_t directions = ((t.flags[c] & CFLAG_UPSTREAM1) ? 1u : 0u) | ((t.flags[c] & CFLAG_UPSTREAM2) ? 2u : 0u);
	if (reciprocal) directions ^= 3u;
	return (uint64_t) t.gene1[c] << 33 | (uint64_t) t.gene2[c] << 2 | directions;
}
AGPU_HD bool both_spliced_is_recoverable(const AnnotationView& ann, const CandidateTable& t, uint32_t c) { // :100-113
	const uint8_t filter = t.filter[c];
	return filter != FILTER_none && both_breakpoints_spliced(ann, t, c) && t.gene1[c] != t.gene2[c] && !candidate_overlaps_both_genes(ann, t, c) && !candidate_is_read_through(t, c) &&
	       (filter == FILTER_relative_support || filter == 17 /* min_support */ || filter == FILTER_in_vitro);
}
// sum over the members of the candidate's gene pair; members[] = candidates sorted by group key, member_keys[] their keys, reads[] = their counts (> 0)
AGPU_HD uint32_t both_spliced_pair_support(const AnnotationView& ann, const CandidateTable& t, uint32_t c, const uint64_t* member_keys, const uint32_t* members, const uint32_t* reads, uint32_t n_members) {
	uint32_t sum = 0;
	const uint64_t same = both_spliced_group_key(t, c, false), reciprocal = both_spliced_group_key(t, c, true);
	for (uint32_t j = lower_bound_u64(member_keys, n_members, same); j < n_members && member_keys[j] == same; ++j) sum += reads[j];
	const bool downstream1 = !(t.flags[c] & CFLAG_UPSTREAM1), downstream2 = !(t.flags[c] & CFLAG_UPSTREAM2);
	for (uint32_t j = lower_bound_u64(member_keys, n_members, reciprocal); j < n_members && member_keys[j] == reciprocal; ++j) {
		const uint32_t other = members[j];
		if (candidate_is_read_through(t, other)) continue;
		// two events with all breakpoints spliced are not questioned; otherwise the reciprocal pair must support a common genomic breakpoint (:125-130)
		if (both_breakpoints_spliced(ann, t, other) || ((downstream1 != (t.breakpoint1[c] > t.breakpoint1[other])) && (downstream2 != (t.breakpoint2[c] > t.breakpoint2[other])))) sum += reads[j];
	}
	return sum;
}
AGPU_HD uint32_t both_spliced_proximal_bonus(const CandidateTable& t, uint32_t c) { // :135
	int32_t distance = t.breakpoint1[c] - t.breakpoint2[c]; if (distance < 0) distance = -distance;
	return ((t.contigs[c] >> 16) == (t.contigs[c] & 0xFFFF) && distance < 1000000) ? 1 : 0;
}

// ---- recover_internal_tandem_duplication (source/recover_internal_tandem_duplication.cpp:11-85): internal tandem duplications inside a coding exon
// that were discarded for one of five reasons come back when enough split reads (unfiltered, or discarded as hairpin / inconsistently clipped /
// mismatches) support them; those reads are un-filtered and counted.  A read listed by several recovered candidates is counted by the first of
// them in the iteration order of fusions_t (hazard H2): the device finds that one with an atomicMin over the iteration ranks.
const uint8_t FILTER_internal_tandem_duplication = 16, FILTER_intragenic_exonic = 15;
AGPU_HD bool itd_read_counts(uint8_t filter) { return filter == FILTER_none || filter == FILTER_hairpin || filter == FILTER_inconsistently_clipped || filter == FILTER_mismatches; }
AGPU_HD bool itd_read_is_cleared(uint8_t filter) { return filter == FILTER_hairpin || filter == FILTER_inconsistently_clipped || filter == FILTER_mismatches; }
// returns 1 = recover, 0 = leave, 2 = the exons at the breakpoints do not fit a device set (capacity error)
AGPU_HD int itd_verdict(const BatchView& b, const AnnotationView& ann, const CoverageView& coverage, const CandidateTable& t, uint32_t c, uint32_t max_itd_length, uint32_t min_supporting_reads,
                        float min_fraction_of_coverage, uint32_t subsampling_threshold, float duplication_rate) {
	AGPU_FP_AS_WRITTEN
	const uint8_t filter = t.filter[c];
	if (filter != FILTER_relative_support && filter != FILTER_intragenic_exonic && filter != FILTER_hairpin && filter != FILTER_inconsistently_clipped && filter != FILTER_mismatches) return 0;
	const uint32_t flags = t.flags[c], gene = t.gene1[c];
	const int32_t breakpoint1 = t.breakpoint1[c], breakpoint2 = t.breakpoint2[c];
	if (!(gene == t.gene2[c] && (flags & CFLAG_EXONIC1) && (flags & CFLAG_EXONIC2) && (flags & CFLAG_UPSTREAM1) && !(flags & CFLAG_UPSTREAM2) && (ann.gene_bits[gene] & GBIT_PROTEIN_CODING) &&
	      (uint32_t) (breakpoint2 - breakpoint1) < max_itd_length)) return 0;
	// both breakpoints in the coding region of one exon of the gene (they may protrude into the introns by 7 bases)
	const int32_t protrude_into_introns = 7;
	AGPU_IDSET(exons);
	IdentityMap identity;
	query_by_coordinate(ann.exon_index, t.contigs[c] >> 16, breakpoint1, breakpoint2, identity, exons);
	if (exons.overflow) return 2;
	bool is_in_coding_region = false;
	for (uint32_t k = 0; k < exons.n; ++k) {
		const uint32_t exon = exons.get(k);
		if (ann.exon_gene[exon] == gene && ann.exon_cds_start[exon] <= breakpoint1 + protrude_into_introns && ann.exon_cds_end[exon] + protrude_into_introns >= breakpoint1 &&
		    ann.exon_cds_start[exon] <= breakpoint2 + protrude_into_introns && ann.exon_cds_end[exon] + protrude_into_introns >= breakpoint2) is_in_coding_region = true;
	}
	if (!is_in_coding_region) return 0;
	const int32_t coverage1 = coverage_near(coverage, t.contigs[c] >> 16, breakpoint1, false), coverage2 = coverage_near(coverage, t.contigs[c] & 0xFFFF, breakpoint2, true); // directions: upstream / downstream
	uint32_t split_reads = 0;
	for (uint64_t k = t.list_offset[3 * (uint64_t) c]; k < t.list_offset[3 * (uint64_t) c + 2]; ++k)
		if (itd_read_counts(b.filter[split_list_entry(t, c, k)])) split_reads++;
	return split_reads >= min_supporting_reads &&
	       (1.0 * split_reads / (coverage1 > coverage2 ? coverage1 : coverage2) / (1 - duplication_rate) > min_fraction_of_coverage || split_reads >= subsampling_threshold);
}
const uint32_t ITD_READ_UNCLAIMED = 0xFFFFFFFFu, ITD_READ_COUNTED = 0xFFFFFFFEu;
// a recovered candidate counts the reads it is the first to clear, split_read1_list before split_read2_list (owner[read] = iteration rank of the first
// recovered candidate that lists the read; set to ITD_READ_COUNTED once counted)
AGPU_HD void itd_count_cleared_reads(const BatchView& b, const CandidateTable& t, uint32_t c, uint32_t my_rank, uint32_t* owner) {
	for (uint32_t list = 0; list < 2; ++list) {
		uint32_t cleared = 0;
		for (uint64_t k = t.list_offset[3 * (uint64_t) c + list]; k < t.list_offset[3 * (uint64_t) c + list + 1]; ++k) {
			const uint32_t read = split_list_entry(t, c, k);
			if (itd_read_is_cleared(b.filter[read]) && owner[read] == my_rank) { owner[read] = ITD_READ_COUNTED; ++cleared; }
		}
		uint32_t* counter = list == 0 ? t.split_reads1 + c : t.split_reads2 + c;
		*counter = (*counter + cleared) & 0x7FFFu; // 15-bit counters (hazard H10)
	}
}

// ---- recover_isoforms (source/recover_isoforms.cpp:10-47): a discarded candidate with both breakpoints at splice sites comes back when its gene pair
// (with the same directions) has a candidate that passed all filters and whose breakpoints are not the same splice sites.  The reference keeps ONE
// unfiltered candidate per gene pair in a std::map filled while iterating fusions_t: the last one in iteration order (hazard H2).  Here the
// unfiltered candidates are sorted by (gene pair + directions, iteration rank); the last of a run is that candidate.  The table is complete before
// the first recovery, as in the reference.
const uint8_t FILTER_isoforms = 35, FILTER_blacklist = 20;
AGPU_HD uint64_t isoform_pair_key(const CandidateTable& t, uint32_t c) { return t.filter[c] == FILTER_none ? both_spliced_group_key(t, c, false) : ~0ull; } // ~0: not in the table
AGPU_HD bool isoform_may_be_recovered(const CandidateTable& t, uint32_t c) { // :26-34
	const uint8_t filter = t.filter[c];
	if (filter == FILTER_none) return false;
	if (filter == FILTER_merge_adjacent || // alternative alignments
	    filter == FILTER_blacklist ||         // normal splice variants and artifacts
	    filter == FILTER_end_to_end ||        // alignments that happen to end at splice sites
	    filter == FILTER_duplicates ||        // nothing but duplicates
	    t.gene1[c] == t.gene2[c])             // circular RNAs
		return false;
	return (t.flags[c] & CFLAG_SPLICED1) && (t.flags[c] & CFLAG_SPLICED2);
}
// member_keys[] = isoform_pair_key of the candidates sorted by (key, iteration rank), members[] the candidates in that order
AGPU_HD bool isoform_is_recovered(const CandidateTable& t, uint32_t c, const uint64_t* member_keys, const uint32_t* members, uint32_t n_members) {
	const uint64_t key = both_spliced_group_key(t, c, false);
	uint32_t lo = 0, hi = n_members; // first position behind the run of `key`
	while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (member_keys[mid] <= key) lo = mid + 1; else hi = mid; }
	if (lo == 0 || member_keys[lo - 1] != key) return false;
	const uint32_t passed = members[lo - 1];
	int32_t distance1 = t.breakpoint1[passed] - t.breakpoint1[c], distance2 = t.breakpoint2[passed] - t.breakpoint2[c];
	if (distance1 < 0) distance1 = -distance1;
	if (distance2 < 0) distance2 = -distance2;
	return distance1 > MAX_SPLICE_SITE_DISTANCE || distance2 > MAX_SPLICE_SITE_DISTANCE; // the same splice sites: an alternative alignment
}

// ---- assign_confidence (source/filter_genomic_support.cpp:222-399, called at source/arriba.cpp:587-589; without structural variants from WGS:
// closest_genomic_breakpoint1 is -1 and the last clause never applies).  A verdict per candidate from its own columns, the coverage, and two
// questions about its neighbourhood that only ask whether at least one other candidate exists: (a) another deletion-like event of one of its
// genes spanning it (read-through events), (b) another spliced event of the same gene pair at other splice sites.  The reference walks
// fusions_by_gene[gene1] and [gene2]; every candidate that can satisfy (a) or (b) shares gene1 (first list) or gene2 (second list) with the
// candidate, so the walks are runs of two sorted orders here: by (gene1, gene2) and by gene2.
enum { CONFIDENCE_LOW = 0, CONFIDENCE_MEDIUM = 1, CONFIDENCE_HIGH = 2 };
// closest pair of genomic breakpoints per candidate (structural variants from WGS, -d; genomic_support_core.hpp); null = no file given: every candidate -1
struct GenomicSupport {
	const int32_t* closest1; const int32_t* closest2;
	AGPU_HD bool has(uint32_t c) const { return closest1 != nullptr && closest1[c] >= 0; }
};
struct ConfidenceTables {
	const uint64_t* pair_keys; const uint32_t* pair_members;   // candidates sorted by gene1 << 32 | gene2
	const uint64_t* gene2_keys; const uint32_t* gene2_members; // candidates sorted by gene2
	uint32_t n;
};
AGPU_HD uint64_t confidence_sort_key(const CandidateTable& t, uint32_t c, int pass) { return pass == 0 ? ((uint64_t) t.gene1[c] << 32 | t.gene2[c]) : (uint64_t) t.gene2[c]; }
AGPU_HD bool confidence_is_spanning_deletion(const CandidateTable& t, uint32_t other, uint32_t c) { // :268-276
	return t.filter[other] == FILTER_none && t.split_reads1[other] + t.split_reads2[other] > 0 &&
	       !(t.flags[other] & CFLAG_UPSTREAM1) && (t.flags[other] & CFLAG_UPSTREAM2) &&
	       ((t.gene1[other] == t.gene1[c] && t.gene2[other] != t.gene2[c]) || (t.gene1[other] != t.gene1[c] && t.gene2[other] == t.gene2[c])) && // not a different isoform
	       (t.breakpoint1[other] != t.breakpoint1[c] || t.breakpoint2[other] != t.breakpoint2[c]) &&
	       t.breakpoint2[other] > t.breakpoint1[c] && t.breakpoint1[other] < t.breakpoint2[c];
}
AGPU_HD bool confidence_has_spanning_deletion(const CandidateTable& t, const ConfidenceTables& tables, uint32_t c) {
	const uint64_t first = (uint64_t) t.gene1[c] << 32;
	for (uint32_t j = lower_bound_u64(tables.pair_keys, tables.n, first); j < tables.n && (tables.pair_keys[j] >> 32) == t.gene1[c]; ++j)
		if (confidence_is_spanning_deletion(t, tables.pair_members[j], c)) return true;
	for (uint32_t j = lower_bound_u64(tables.gene2_keys, tables.n, t.gene2[c]); j < tables.n && tables.gene2_keys[j] == t.gene2[c]; ++j)
		if (confidence_is_spanning_deletion(t, tables.gene2_members[j], c)) return true;
	return false;
}
AGPU_HD bool confidence_has_other_spliced_event(const CandidateTable& t, const ConfidenceTables& tables, uint32_t c) { // :331-347, filtered candidates count too
	const uint64_t key = (uint64_t) t.gene1[c] << 32 | t.gene2[c];
	for (uint32_t j = lower_bound_u64(tables.pair_keys, tables.n, key); j < tables.n && tables.pair_keys[j] == key; ++j) {
		const uint32_t other = tables.pair_members[j];
		if (!(t.flags[other] & CFLAG_SPLICED1) || !(t.flags[other] & CFLAG_SPLICED2)) continue;
		int32_t distance1 = t.breakpoint1[other] - t.breakpoint1[c], distance2 = t.breakpoint2[other] - t.breakpoint2[c];
		if (distance1 < 0) distance1 = -distance1;
		if (distance2 < 0) distance2 = -distance2;
		if (distance1 > 2 || distance2 > 2) return true;
	}
	return false;
}
AGPU_HD uint8_t candidate_confidence(const AnnotationView& ann, const CoverageView& coverage, const CandidateTable& t, const float* evalues, const ConfidenceTables& tables, uint32_t c, const GenomicSupport& wgs = GenomicSupport{ nullptr, nullptr }) {
	AGPU_FP_AS_WRITTEN
	if (t.filter[c] != FILTER_none) return CONFIDENCE_LOW; // discarded events get low confidence, no matter what
	const uint32_t flags = t.flags[c], split_reads1 = t.split_reads1[c], split_reads2 = t.split_reads2[c], discordant_mates = t.discordant_mates[c], supporting_reads = split_reads1 + split_reads2 + discordant_mates;
	const bool spliced1 = flags & CFLAG_SPLICED1, spliced2 = flags & CFLAG_SPLICED2, exonic1 = flags & CFLAG_EXONIC1, exonic2 = flags & CFLAG_EXONIC2, same_gene = t.gene1[c] == t.gene2[c];
	const float evalue = evalues[c];
	// supporting reads as a fraction of the coverage; the sizes of the read lists, because the coverage includes duplicates, too (:235-239)
	const int32_t coverage1 = coverage_near(coverage, t.contigs[c] >> 16, t.breakpoint1[c], !(flags & CFLAG_UPSTREAM1)), coverage2 = coverage_near(coverage, t.contigs[c] & 0xFFFF, t.breakpoint2[c], !(flags & CFLAG_UPSTREAM2));
	const int32_t larger = coverage1 > coverage2 ? coverage1 : coverage2;
	const float coverage_fraction = ((float) (t.list_offset[3 * (uint64_t) c + 3] - t.list_offset[3 * (uint64_t) c])) / (larger > 1 ? larger : 1);
	const bool read_through = candidate_is_read_through(t, c);
	int confidence = CONFIDENCE_HIGH; // high by default, reduced by penalties
	if (evalue > 0.3 || supporting_reads < 2) {
		confidence = CONFIDENCE_LOW; // poor support
	} else if (read_through) {
		confidence = CONFIDENCE_LOW; // unless well supported, or other deletions hint at the same genomic event
		if (((split_reads1 > 0 && split_reads2 > 0) || (split_reads1 > 0 && discordant_mates > 0) || (split_reads2 > 0 && discordant_mates > 0)) && supporting_reads >= 10)
			confidence = (split_reads1 + split_reads2 >= 10 && coverage_fraction > 0.07) ? CONFIDENCE_HIGH : CONFIDENCE_MEDIUM;
		else if (confidence_has_spanning_deletion(t, tables, c))
			confidence = CONFIDENCE_MEDIUM;
	} else if (candidate_overlaps_both_genes(ann, t, c) || same_gene) { // intragenic: low by default, there are so many artifacts
		confidence = CONFIDENCE_LOW;
		if (split_reads1 + split_reads2 > 0) {
			if (!exonic1 && !exonic2) confidence = (split_reads1 > 0 && split_reads2 > 0) ? CONFIDENCE_HIGH : CONFIDENCE_MEDIUM;   // most intragenic artifacts have both breakpoints in exons
			else if (!exonic1 || !exonic2) confidence = (split_reads1 > 3 && split_reads2 > 3) ? CONFIDENCE_HIGH : CONFIDENCE_MEDIUM; // one breakpoint in an intron: more split reads
		}
	}
	// rescued internal tandem duplications (:318-328)
	if (confidence == CONFIDENCE_LOW && same_gene && exonic1 && exonic2 && !spliced1 && !spliced2 && t.breakpoint2[c] - t.breakpoint1[c] < 100 && split_reads1 > 0 && split_reads2 > 0 &&
	    split_reads1 + split_reads2 >= 10 && coverage_fraction > 0.15 && (flags & CFLAG_UPSTREAM1) && !(flags & CFLAG_UPSTREAM2))
		confidence = CONFIDENCE_MEDIUM;
	// several spliced events between the same pair of genes (:331-350)
	if (confidence < CONFIDENCE_HIGH && spliced1 && spliced2 && !read_through && !same_gene && confidence_has_other_spliced_event(t, tables, c)) ++confidence;
	// true events are likely to have at least one spliced breakpoint, unless intragenic (:354-357)
	if (!same_gene && confidence > CONFIDENCE_LOW && !spliced1 && !spliced2) --confidence;
	// excellent support (:360-361)
	if (split_reads1 > 20 && split_reads2 > 20 && supporting_reads > 60) confidence = CONFIDENCE_HIGH;
	// something does not look right with the number of supporting reads (:364-388)
	if (confidence > CONFIDENCE_LOW) {
		if (split_reads1 + split_reads2 == 0 || split_reads1 + discordant_mates == 0 || split_reads2 + discordant_mates == 0) --confidence; // reads from both ends are expected
		else if ((split_reads1 + split_reads2) * 20 < discordant_mates) --confidence;                                                 // split reads and discordant mates should be balanced
		else if (evalue > 0.2 || coverage_fraction < 0.01) confidence = CONFIDENCE_MEDIUM;                                            // not overwhelming compared to the coverage
	}
	// a supporting structural variant (:391-397)
	if (confidence < CONFIDENCE_HIGH && wgs.has(c)) {
		int32_t distance1 = t.breakpoint1[c] - wgs.closest1[c], distance2 = t.breakpoint2[c] - wgs.closest2[c], span = t.breakpoint2[c] - t.breakpoint1[c];
		if (distance1 < 0) distance1 = -distance1;
		if (distance2 < 0) distance2 = -distance2;
		if (span < 0) span = -span;
		if ((evalue < 0.3 && supporting_reads >= 2) ||                      // good e-value, or
		    (spliced1 && spliced2 && !same_gene) ||                         // recovered due to splicing, or
		    distance1 + distance2 < 20000 ||                                // genomic breakpoints very close to the transcriptomic ones, or
		    (t.contigs[c] >> 16) != (t.contigs[c] & 0xFFFF) || (span > 1000000 && !same_gene)) // a distant translocation
			++confidence;
	}
	return (uint8_t) confidence;
}

// the stage as one switch (kernel and host stepping share it); returns the filter id the candidate gets, FILTER_none if it stays, or
// EVENT_KEPT_UNCOUNTED if it stays without entering the "(remaining=N)" of the stage: filter_both_intronic and filter_end_to_end_fusions skip
// the candidates on viral contigs with `continue` before they count (source/filter_both_intronic.cpp:25-26, source/filter_end_to_end.cpp:38-39)
const uint8_t EVENT_KEPT_UNCOUNTED = 0xFF;
enum { EVENT_count_only = -1, EVENT_both_intronic = 0, EVENT_short_anchor = 1, EVENT_end_to_end = 2, EVENT_no_coverage = 3, EVENT_marginal_read_through = 4 };
template <class Lanes = ListLanes> AGPU_HD uint8_t event_predicate(int stage, const BatchView& b, const AnnotationView& ann, const GenomeView& genome, const CoverageView& coverage, const CandidateTable& t, uint32_t c, uint32_t min_anchor_length, const Lanes& lanes = Lanes()) {
	switch (stage) {
		case EVENT_both_intronic:
			if (contig_is_viral(genome, t.contigs[c] >> 16) || contig_is_viral(genome, t.contigs[c] & 0xFFFF)) return EVENT_KEPT_UNCOUNTED;
			return has_only_intronic_reads(b, genome, t, c, lanes) ? FILTER_intronic : FILTER_none;
		case EVENT_short_anchor: return has_short_anchor(t, c, min_anchor_length) ? FILTER_short_anchor : FILTER_none;
		case EVENT_end_to_end:
			if (contig_is_viral(genome, t.contigs[c] >> 16) || contig_is_viral(genome, t.contigs[c] & 0xFFFF)) return EVENT_KEPT_UNCOUNTED;
			return is_end_to_end_fusion(ann, genome, t, c) ? FILTER_end_to_end : FILTER_none;
		case EVENT_no_coverage: return has_no_coverage(ann, coverage, t, c) ? FILTER_no_coverage : FILTER_none;
		case EVENT_marginal_read_through: return is_marginal_read_through(ann, coverage, t, c) ? FILTER_marginal_read_through : FILTER_none;
		default: return FILTER_none; // EVENT_count_only: the stage is switched off (-f), the caller still gets the number of unfiltered candidates
	}
}

}

#endif
