// arriba_amd/csrc/device/filter_core.hpp -- the read-level filter cascade, one thread per chimeric fragment.
//
// Each predicate restates one reference stage as a pure function of the fragment's columns:
//   duplicates key          source/filter_duplicates.cpp:22-45
//   uninteresting_contigs   source/filter_uninteresting_contigs.cpp:8-26
//   viral_contigs           source/filter_viral_contigs.cpp:8-27
//   top_expressed / low_coverage viral contigs (per-read part)   source/filter_top_expressed_viral_contigs.cpp:128-151, source/filter_low_coverage_viral_contigs.cpp:28-48
//   read_through            source/filter_proximal_read_through.cpp:8-47
//   inconsistently_clipped  source/filter_inconsistently_clipped.cpp:6-25
//   homopolymer             source/filter_homopolymer.cpp:7-62
//   small_insert_size       source/filter_small_insert_size.cpp:7-30
//   long_gap                source/filter_long_gap.cpp:6-89
//   same_gene               source/filter_same_gene.cpp:8-46
//   hairpin                 source/filter_hairpin.cpp:7-80
//   mismatches              source/filter_mismatches.cpp:12-135 (verdict via a host-built table, hazard H5)
//   low_entropy             source/filter_low_entropy.cpp:9-112, kmer_to_int source/filter_mismappers.cpp:33-45
#ifndef AGPU_FILTER_CORE_HPP
#define AGPU_FILTER_CORE_HPP 1

#include "annotate_core.hpp"

namespace agpu {

struct FilterTables {
	// mismatch verdict for (mismatches k, aligned length n), k <= n <= mismatch_max_length: bit (n*(n+1)/2 + k); k > n is always a discard
	const uint32_t* mismatch_verdict;
	uint32_t mismatch_max_length;
	// low-entropy thresholds per segment length: unsigned(len * max_kmer_content / 3 + 0.5) computed with the reference's float/double mix
	const uint32_t* kmer_threshold;
	uint32_t kmer_threshold_size;
	float max_kmer_content;
	uint32_t homopolymer_length;
	int32_t min_read_through_distance;
	uint32_t max_itd_length;
	uint8_t external_duplicate_marking;
	const uint8_t* top_expressed_viral_verdict; // per contig, may be null
	const uint8_t* low_coverage_viral_verdict;  // per contig, may be null
};

AGPU_HD uint32_t preclipping(const uint32_t* cigar, uint32_t n) { uint32_t op = cigar[0] & 15; return (op == CIGAR_S || op == CIGAR_H) ? cigar[0] >> 4 : 0; }
AGPU_HD uint32_t postclipping(const uint32_t* cigar, uint32_t n) { uint32_t op = cigar[n - 1] & 15; return (op == CIGAR_S || op == CIGAR_H) ? cigar[n - 1] >> 4 : 0; }
AGPU_HD const uint32_t* cigar_of(const BatchView& b, int slot, uint64_t i) { return b.cigar_pool + b.cigar_offset[slot][i]; }

// BAM 4-bit code -> ASCII as the reference stores it (seq_nt16_str = "=ACMGRSVTWYHKDBN"), from two 64-bit register constants
AGPU_HD char base_char(uint32_t code) {
	const uint64_t low = 0x56535247'4d43413dULL, high = 0x4e42444b'48595754ULL; // "=ACMGRSV" / "TWYHKDBN", little endian
	return (char) (((code & 8) ? high : low) >> ((code & 7) * 8));
}
// sequences are packed two bases per byte (high nibble first) and start on a 4-byte boundary: read them as 32-bit words
AGPU_HD uint32_t base_code_from_words(const uint32_t* words, uint32_t position) {
	uint32_t word = words[position >> 3];
	return (word >> ((((position & 7) >> 1) << 3) + ((~position & 1) << 2))) & 15;
}
// complement on codes (reference: dna_to_complement, source/assembly.hpp:9-21 swaps only A<->T and C<->G)
AGPU_HD uint32_t complement_code(uint32_t code) { return code == 1 ? 8 : code == 8 ? 1 : code == 2 ? 4 : code == 4 ? 2 : code; }

struct SequenceRef {
	const uint32_t* words; uint32_t length; bool reverse_complement;
	AGPU_HD uint32_t code(uint32_t position) const { return reverse_complement ? complement_code(base_code_from_words(words, length - 1 - position)) : base_code_from_words(words, position); }
	AGPU_HD char at(uint32_t position) const { return base_char(code(position)); }
};
// The kernels may stage the sequence pool span of a workgroup in LDS; `staged` then points at the LDS copy of pool word
// `staged_first_word` and covers `staged_words` words.  Sequences outside the staged span are read from HBM.
struct SequenceStage { const uint32_t* staged; uint32_t staged_first_word, staged_words; };
AGPU_HD SequenceStage no_stage() { SequenceStage s; s.staged = 0; s.staged_first_word = 0; s.staged_words = 0; return s; }
AGPU_HD SequenceRef sequence_of(const BatchView& b, int slot, uint64_t i, const SequenceStage& stage) {
	SequenceRef s;
	uint32_t first_word = b.seq_offset[slot][i];
	s.length = b.seq_length[slot][i];
	uint32_t words = (s.length + 7) >> 3;
	if (stage.staged != 0 && first_word >= stage.staged_first_word && first_word + words <= stage.staged_first_word + stage.staged_words)
		s.words = stage.staged + (first_word - stage.staged_first_word);
	else
		s.words = (const uint32_t*) b.seq_pool + first_word;
	s.reverse_complement = false;
	return s;
}

// ---- duplicates -----------------------------------------------------------------------------------

struct DuplicateKey { uint32_t contigs; int32_t position1, position2; }; // contigs = contig1 << 16 | contig2
AGPU_HD bool keys_equal(const DuplicateKey& a, const DuplicateKey& b) { return a.contigs == b.contigs && a.position1 == b.position1 && a.position2 == b.position2; }

AGPU_HD DuplicateKey duplicate_key(const BatchView& b, uint64_t i) {
	int mate2 = (b.n_aln[i] == 2) ? MATE2 : SUPPLEMENTARY;
	const uint32_t* cigar1 = cigar_of(b, MATE1, i); uint32_t n1 = b.cigar_count[MATE1][i];
	const uint32_t* cigar2 = cigar_of(b, mate2, i); uint32_t n2 = b.cigar_count[mate2][i];
	int32_t position1 = (b.abits[MATE1][i] & ABIT_STRAND) ? b.start[MATE1][i] - (int32_t) preclipping(cigar1, n1) : b.end[MATE1][i] + (int32_t) postclipping(cigar1, n1);
	int32_t position2 = (b.abits[mate2][i] & ABIT_STRAND) ? b.start[mate2][i] - (int32_t) preclipping(cigar2, n2) : b.end[mate2][i] + (int32_t) postclipping(cigar2, n2);
	uint32_t contig1 = b.contig[MATE1][i], contig2 = b.contig[mate2][i];
	if (position1 > position2) { int32_t t = position1; position1 = position2; position2 = t; uint32_t c = contig1; contig1 = contig2; contig2 = c; }
	DuplicateKey key; key.contigs = contig1 << 16 | contig2; key.position1 = position1; key.position2 = position2;
	return key;
}
AGPU_HD uint64_t hash_duplicate_key(const DuplicateKey& key) {
	uint64_t h = ((uint64_t) (uint32_t) key.position1 << 32 | (uint32_t) key.position2) * 0x9E3779B97F4A7C15ULL;
	h ^= (uint64_t) key.contigs * 0xC2B2AE3D27D4EB4FULL;
	h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
	return h;
}

// ---- contig based filters (stage group 1) ----------------------------------------------------------

// returns the filter id that discards fragment i, or FILTER_none
AGPU_HD uint8_t contig_filters(const BatchView& b, const GenomeView& genome, const FilterTables& t, uint64_t i) {
	int n_aln = b.n_aln[i];
	bool all_viral = true;
	for (int s = 0; s < n_aln; ++s) {
		uint8_t bits = genome.contig_bits[b.contig[s][i]];
		if (!(bits & CBIT_INTERESTING)) return FILTER_uninteresting_contigs;
	}
	for (int s = 0; s < n_aln; ++s)
		if (!(genome.contig_bits[b.contig[s][i]] & CBIT_VIRAL)) all_viral = false;
	if (all_viral) return FILTER_viral_contigs;
	if (t.top_expressed_viral_verdict != 0)
		for (int s = 0; s < n_aln; ++s) {
			uint32_t contig = b.contig[s][i];
			if ((genome.contig_bits[contig] & CBIT_VIRAL) && t.top_expressed_viral_verdict[contig]) return FILTER_top_expressed_viral_contigs;
		}
	if (t.low_coverage_viral_verdict != 0)
		for (int s = 0; s < n_aln; ++s) {
			uint32_t contig = b.contig[s][i];
			if ((genome.contig_bits[contig] & CBIT_VIRAL) && t.low_coverage_viral_verdict[contig]) return FILTER_low_coverage_viral_contigs;
		}
	return FILTER_none;
}

// ---- spliced distance (for the fragment-length samples) -------------------------------------------

// reference: get_spliced_distance, source/annotation.cpp:570-618
AGPU_HD int32_t spliced_distance(const AnnotationView& ann, uint32_t contig, int32_t position1, int32_t position2, uint32_t gene) {
	if (position1 > position2) { int32_t t = position1; position1 = position2; position2 = t; }
	const FlatIndexView& index = ann.exon_index;
	if (contig >= index.n_contigs || index.contig_offset[contig] == index.contig_offset[contig + 1])
		return position2 - position1;
	uint32_t k = index_lower_bound(index, contig, position1);
	uint32_t contig_end = index.contig_offset[contig + 1];
	int32_t distance = 0;
	if (k != contig_end && index.keys[k] < position2) {
		distance += index.keys[k] - position1;
		position1 = index.keys[k];
	}
	for (; k != contig_end && index.keys[k] < position2; ++k) {
		if (index.keys[k] < position1) continue;
		int32_t best_start = -1, best_end = -1, best_skip = -1;
		ListRef exons = index_bucket(index, k);
		for (uint32_t m = 0; m < exons.n; ++m) {
			uint32_t e = exons.p[m];
			if (ann.exon_gene[e] != gene) continue;
			int32_t next = ann.exon_next[e];
			if (next == -1 || ann.exon_start[next] > position2) continue;
			int32_t exon_start = ann.exon_start[e] > position1 ? ann.exon_start[e] : position1;
			int32_t exon_end = ann.exon_end[e] < position2 ? ann.exon_end[e] : position2;
			int32_t exon_skip = ann.exon_start[next] - exon_start + 1;
			if (best_start == -1 || 1.0 * (exon_end - exon_start) / exon_skip < 1.0 * (best_end - best_start) / best_skip) {
				best_start = exon_start; best_end = exon_end; best_skip = exon_skip;
			}
		}
		if (best_start != -1) {
			distance += best_end - best_start + 1;
			position1 = best_start + best_skip - 1;
		}
	}
	distance += position2 - position1;
	return distance;
}

// mate gap of a paired split read as estimate_fragment_length samples it (source/read_stats.cpp:27-39)
AGPU_HD int32_t mate_gap_sample(const BatchView& b, const AnnotationView& ann, uint64_t i) {
	int forward = MATE1, reverse = SPLIT_READ;
	if (!(b.abits[MATE1][i] & ABIT_STRAND)) { forward = SPLIT_READ; reverse = MATE1; }
	AGPU_IDSET(genes); load_genes(b, forward, i, genes);
	int32_t forward_end = b.end[forward][i], reverse_start = b.start[reverse][i];
	int32_t distance = spliced_distance(ann, b.contig[forward][i], forward_end, reverse_start, genes.low[0]);
	if (forward_end > reverse_start) distance *= -1;
	int32_t forward_length = (int32_t) b.seq_length[forward][i], reverse_length = (int32_t) b.seq_length[reverse][i];
	if (distance < -forward_length) distance = -forward_length;
	if (distance < -reverse_length) distance = -reverse_length;
	return distance;
}

// ---- stage group 2 predicates -----------------------------------------------------------------------

// boundaries of the genes of alignment `slot` (source/annotation.cpp:558-567); the ids are read from the gene-set columns, so that a slot
// chosen at run time does not force an array of sets out of the registers
AGPU_HD void gene_boundaries(const BatchView& b, const AnnotationView& ann, int slot, uint64_t i, int32_t& start, int32_t& end) {
	start = -1; end = -1;
	const uint32_t count = b.gene_count[slot][i];
	const uint32_t* ids = b.genes[slot] + i * GENE_INLINE;
	if (count > (uint32_t) GENE_INLINE) ids = b.gene_pool + ids[0];
	for (uint32_t g = 0; g < count; ++g) {
		const uint32_t gene = ids[g];
		int32_t gene_start = ann.gene_start[gene], gene_end = ann.gene_end[gene];
		if (start == -1 || start > gene_start) start = gene_start;
		if (end == -1 || end < gene_end) end = gene_end;
	}
}

AGPU_HD bool is_proximal_read_through(const BatchView& b, const AnnotationView& ann, const FilterTables& t, uint64_t i) {
	int n_aln = b.n_aln[i];
	int forward, reverse;
	if (n_aln == 2) {
		bool mate1_forward = b.abits[MATE1][i] & ABIT_STRAND;
		forward = mate1_forward ? MATE1 : MATE2; reverse = mate1_forward ? MATE2 : MATE1;
	} else {
		bool split_forward = b.abits[SPLIT_READ][i] & ABIT_STRAND;
		forward = split_forward ? SUPPLEMENTARY : SPLIT_READ; reverse = split_forward ? SPLIT_READ : SUPPLEMENTARY;
	}
	bool forward_strand = b.abits[forward][i] & ABIT_STRAND, reverse_strand = b.abits[reverse][i] & ABIT_STRAND;
	bool colinear = b.contig[forward][i] == b.contig[reverse][i] && b.end[forward][i] < b.start[reverse][i];
	if ((n_aln == 2 && forward_strand != reverse_strand && colinear) || (n_aln == 3 && forward_strand == reverse_strand && colinear)) {
		int32_t forward_gene_start, forward_gene_end, reverse_gene_start, reverse_gene_end;
		gene_boundaries(b, ann, forward, i, forward_gene_start, forward_gene_end);
		gene_boundaries(b, ann, reverse, i, reverse_gene_start, reverse_gene_end);
		if (b.end[forward][i] >= reverse_gene_start - t.min_read_through_distance || b.start[reverse][i] <= forward_gene_end + t.min_read_through_distance)
			return true;
	}
	return false;
}

AGPU_HD bool is_inconsistently_clipped(const BatchView& b, uint64_t i) {
	if (b.n_aln[i] != 3) return false;
	bool forward = b.abits[MATE1][i] & ABIT_STRAND;
	return (forward && b.end[MATE1][i] > b.end[SPLIT_READ][i] + 3) || (!forward && b.start[MATE1][i] < b.start[SPLIT_READ][i] - 3);
}

AGPU_HD bool split_read_is_spliced(const BatchView& b, const AnnotationView& ann, uint64_t i, const IdSet& genes) { // source/filter_homopolymer.cpp:7-14
	bool forward = b.abits[SPLIT_READ][i] & ABIT_STRAND;
	int32_t breakpoint = forward ? b.start[SPLIT_READ][i] : b.end[SPLIT_READ][i];
	for (uint32_t g = 0; g < genes.n; ++g)
		if (is_breakpoint_spliced(ann, genes.get(g), forward, breakpoint))
			return true;
	return false;
}

AGPU_HD bool has_homopolymer_at_breakpoint(const BatchView& b, const AnnotationView& ann, const FilterTables& t, uint64_t i, const IdSet& split_read_genes, const SequenceStage& stage) {
	if (b.n_aln[i] != 3) return false;
	SequenceRef sequence = sequence_of(b, SPLIT_READ, i, stage);
	const uint32_t* cigar = cigar_of(b, SPLIT_READ, i); uint32_t n_cigar = b.cigar_count[SPLIT_READ][i];
	uint32_t H = t.homopolymer_length, length = sequence.length;
	// the reference concatenates up to two H-mers, each followed by a blank; segments: [start, start+H)
	uint32_t segment_start[2]; int segments = 0;
	if (b.abits[SPLIT_READ][i] & ABIT_STRAND) {
		uint32_t clip = preclipping(cigar, n_cigar);
		if (clip >= H) segment_start[segments++] = clip - H;
		if (length - clip >= H) segment_start[segments++] = clip;
	} else {
		uint32_t clip = postclipping(cigar, n_cigar);
		if (clip >= H) segment_start[segments++] = length - clip;
		if (length - clip >= H) segment_start[segments++] = length - clip - H;
	}
	// runs cannot cross the blank separator, so each H-mer is checked on its own: a run of H equal characters == all H equal
	for (int s = 0; s < segments; ++s) {
		uint32_t run = 1;
		for (uint32_t c = 1; c < H; ++c) {
			if (sequence.code(segment_start[s] + c - 1) == sequence.code(segment_start[s] + c)) {
				++run;
				if (run == H && !split_read_is_spliced(b, ann, i, split_read_genes))
					return true;
			} else {
				run = 1;
			}
		}
	}
	return false;
}

AGPU_HD bool has_small_insert_size(const BatchView& b, uint64_t i, int32_t max_overhang) {
	if (b.n_aln[i] != 2) return false;
	if (((b.abits[MATE1][i] ^ b.abits[MATE2][i]) & ABIT_STRAND) && b.contig[MATE1][i] == b.contig[MATE2][i]) {
		int32_t d1 = b.start[MATE1][i] - b.start[MATE2][i]; if (d1 < 0) d1 = -d1;
		int32_t d2 = b.end[MATE1][i] - b.end[MATE2][i]; if (d2 < 0) d2 = -d2;
		// the reference compares abs(int) with an unsigned max_overhang; both are non-negative
		return d1 <= max_overhang || d2 <= max_overhang;
	}
	return false;
}

AGPU_HD bool has_long_gap(const BatchView& b, uint64_t i) {
	const int32_t min_long_gap = 700000, max_long_gap = 1500000;
	const uint32_t short_segment = 15;
	int n_aln = b.n_aln[i];
	int32_t size_of_deletion = 0;
	if (n_aln == 3 && b.contig[SPLIT_READ][i] == b.contig[SUPPLEMENTARY][i]) {
		bool split_forward = b.abits[SPLIT_READ][i] & ABIT_STRAND, supp_forward = b.abits[SUPPLEMENTARY][i] & ABIT_STRAND;
		if (!split_forward && !supp_forward) size_of_deletion = b.start[SUPPLEMENTARY][i] - b.end[SPLIT_READ][i];
		else if (split_forward && supp_forward) size_of_deletion = b.start[SPLIT_READ][i] - b.end[SUPPLEMENTARY][i];
	}
	for (int s = 0; s < n_aln; ++s) {
		const uint32_t* cigar = cigar_of(b, s, i); uint32_t n = b.cigar_count[s][i];
		for (uint32_t c = 1; c + 1 < n; ++c) {
			if ((cigar[c] & 15) == CIGAR_N && ((int32_t) (cigar[c] >> 4) >= min_long_gap || (size_of_deletion >= min_long_gap && size_of_deletion <= max_long_gap))) {
				uint32_t left = 0, right = 0;
				for (int32_t j = (int32_t) c - 1; j >= 0; --j) {
					uint32_t op = cigar[j] & 15;
					if (op == CIGAR_M || op == CIGAR_X || op == CIGAR_EQ) left += cigar[j] >> 4;
					else if (op == CIGAR_D || op == CIGAR_I || op == CIGAR_P) {}
					else break;
				}
				for (uint32_t j = c + 1; j < n; ++j) {
					uint32_t op = cigar[j] & 15;
					if (op == CIGAR_M || op == CIGAR_X || op == CIGAR_EQ) right += cigar[j] >> 4;
					else if (op == CIGAR_D || op == CIGAR_I || op == CIGAR_P) {}
					else break;
				}
				if (left <= short_segment && right <= short_segment)
					return true;
			}
		}
	}
	return false;
}

AGPU_HD bool is_same_gene_artifact(const BatchView& b, uint64_t i, const IdSet* genes) {
	AGPU_IDSET(common);
	if (b.n_aln[i] == 2) intersect_sets(genes[MATE1], genes[MATE2], common);
	else intersect_sets(genes[MATE2], genes[SUPPLEMENTARY], common);
	if (common.n == 0) return false;
	if (b.n_aln[i] == 2) {
		bool forward1 = b.abits[MATE1][i] & ABIT_STRAND, forward2 = b.abits[MATE2][i] & ABIT_STRAND;
		return (forward1 && !forward2 && b.start[MATE1][i] <= b.end[MATE2][i]) || (!forward1 && forward2 && b.end[MATE1][i] >= b.start[MATE2][i]);
	}
	bool split_forward = b.abits[SPLIT_READ][i] & ABIT_STRAND, supp_forward = b.abits[SUPPLEMENTARY][i] & ABIT_STRAND;
	return (split_forward && supp_forward && b.start[SPLIT_READ][i] >= b.end[SUPPLEMENTARY][i]) || (!split_forward && !supp_forward && b.end[SPLIT_READ][i] <= b.start[SUPPLEMENTARY][i]);
}

AGPU_HD bool breakpoint_within_aligned_segment(const BatchView& b, int slot, uint64_t i, int32_t breakpoint) { // source/filter_hairpin.cpp:7-27
	const uint32_t* cigar = cigar_of(b, slot, i); uint32_t n = b.cigar_count[slot][i];
	int32_t reference_position = b.start[slot][i];
	for (uint32_t c = 0; c < n; ++c) {
		uint32_t op = cigar[c] & 15; int32_t length = cigar[c] >> 4;
		if (op == CIGAR_N || op == CIGAR_D) reference_position += length;
		else if (op == CIGAR_M || op == CIGAR_X || op == CIGAR_EQ) {
			if (breakpoint >= reference_position && breakpoint <= reference_position + length) return true;
			reference_position += length;
		}
	}
	return false;
}

AGPU_HD bool is_hairpin(const BatchView& b, uint64_t i, const IdSet* genes) {
	AGPU_IDSET(common);
	if (b.n_aln[i] == 2) {
		intersect_sets(genes[MATE1], genes[MATE2], common);
		if (common.n == 0 && b.contig[MATE1][i] != b.contig[MATE2][i]) return false;
		int32_t breakpoint1 = (b.abits[MATE1][i] & ABIT_STRAND) ? b.end[MATE1][i] : b.start[MATE1][i];
		int32_t breakpoint2 = (b.abits[MATE2][i] & ABIT_STRAND) ? b.end[MATE2][i] : b.start[MATE2][i];
		return breakpoint_within_aligned_segment(b, MATE2, i, breakpoint1) || breakpoint_within_aligned_segment(b, MATE1, i, breakpoint2);
	}
	intersect_sets(genes[SPLIT_READ], genes[SUPPLEMENTARY], common);
	if (common.n == 0 && b.contig[SPLIT_READ][i] != b.contig[SUPPLEMENTARY][i]) return false;
	int32_t breakpoint_split = (b.abits[SPLIT_READ][i] & ABIT_STRAND) ? b.start[SPLIT_READ][i] : b.end[SPLIT_READ][i];
	int32_t breakpoint_supp = (b.abits[SUPPLEMENTARY][i] & ABIT_STRAND) ? b.end[SUPPLEMENTARY][i] : b.start[SUPPLEMENTARY][i];
	return breakpoint_within_aligned_segment(b, SUPPLEMENTARY, i, breakpoint_split) || breakpoint_within_aligned_segment(b, SPLIT_READ, i, breakpoint_supp) ||
	       breakpoint_within_aligned_segment(b, MATE1, i, breakpoint_supp);
}

// ---- sixteen bases at a time ------------------------------------------------------------------------------------------------
// The mismatch walk of source/filter_mismatches.cpp:12-52 compares read and reference base by base.  Here an aligned block (M/=/X)
// is processed in chunks of 16 bases: three sequence words and five genome words are loaded back to back (independent loads instead
// of one dependent load per four bases), the sixteen 4-bit read codes are turned into characters four at a time and XOR-ed with the
// reference bytes.

AGPU_HD uint32_t swap_nibbles(uint32_t word) { return ((word & 0x0F0F0F0Fu) << 4) | ((word >> 4) & 0x0F0F0F0Fu); } // base j of the word -> bits 4j..4j+3

// codes of the physical positions first .. first+15 of the sequence (code k at bits 4k); positions behind the last word repeat it
AGPU_HD uint64_t physical_codes16(const uint32_t* words, uint32_t n_words, uint32_t first) {
	const uint32_t w = first >> 3, shift = (first & 7) << 2;
	const uint32_t last = n_words - 1;
	const uint64_t low = (uint64_t) swap_nibbles(words[w + 1 < last ? w + 1 : last]) << 32 | swap_nibbles(words[w < last ? w : last]);
	const uint64_t high = swap_nibbles(words[w + 2 < last ? w + 2 : last]);
	return shift ? (low >> shift) | (high << (64 - shift)) : low;
}
AGPU_HD uint64_t reverse_nibbles(uint64_t x) {
	x = (x >> 32) | (x << 32);
	x = ((x & 0xFFFF0000FFFF0000ULL) >> 16) | ((x & 0x0000FFFF0000FFFFULL) << 16);
	x = ((x & 0xFF00FF00FF00FF00ULL) >> 8) | ((x & 0x00FF00FF00FF00FFULL) << 8);
	return ((x & 0xF0F0F0F0F0F0F0F0ULL) >> 4) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
}
// complement of A/C/G/T codes (1,2,4,8) = bit reversal of the nibble; 0 and 15 map to themselves.  Other ambiguity codes do not:
// the caller checks only_plain_codes() first.
AGPU_HD uint64_t complement_plain_codes(uint64_t x) {
	return ((x & 0x1111111111111111ULL) << 3) | ((x & 0x2222222222222222ULL) << 1) | ((x & 0x4444444444444444ULL) >> 1) | ((x & 0x8888888888888888ULL) >> 3);
}
AGPU_HD bool only_plain_codes(uint64_t x) { // every nibble has at most one bit set or is 15
	const uint64_t ones = 0x1111111111111111ULL;
	const uint64_t sum = (x & ones) + ((x >> 1) & ones) + ((x >> 2) & ones) + ((x >> 3) & ones); // bits set per nibble: 0..4
	return (((sum >> 1) & ones) & ~((sum >> 2) & ones)) == 0;                                      // none has 2 or 3
}
// codes of the logical positions position .. position+15 of a (possibly reverse-complemented) sequence; ok = false if the chunk holds
// an ambiguity code that the nibble tricks do not complement like the reference does
AGPU_HD uint64_t logical_codes16(const SequenceRef& sequence, uint32_t position, bool& ok) {
	const uint32_t n_words = (sequence.length + 7) >> 3;
	ok = true;
	if (!sequence.reverse_complement) return physical_codes16(sequence.words, n_words, position);
	// logical p <-> physical length-1-p: the sixteen positions end at physical `top`
	const int32_t top = (int32_t) sequence.length - 1 - (int32_t) position, first = top - 15;
	uint64_t codes = physical_codes16(sequence.words, n_words, first > 0 ? (uint32_t) first : 0u);
	if (first < 0) codes <<= (uint32_t) (-first) << 2; // the chunk starts before the sequence: its leading (logical: trailing) codes are not used
	codes = reverse_nibbles(codes);
	ok = only_plain_codes(codes);
	return complement_plain_codes(codes);
}

// four 4-bit codes (one per byte of `spread`, values 0..15) -> the four characters of seq_nt16_str
AGPU_HD uint32_t characters_of_codes(uint32_t spread) {
#if defined(__HIP_DEVICE_COMPILE__)
	const uint32_t select = spread & 0x07070707u;
	const uint32_t low = __builtin_amdgcn_perm(0x56535247u, 0x4d43413du, select);   // "=ACM" | "GRSV"
	const uint32_t high = __builtin_amdgcn_perm(0x4e42444bu, 0x48595754u, select);  // "TWYH" | "KDBN"
	const uint32_t take_high = ((spread >> 3) & 0x01010101u) * 0xFFu;
	return (high & take_high) | (low & ~take_high);
#else
	return (uint32_t) (uint8_t) base_char(spread & 15) | (uint32_t) (uint8_t) base_char(spread >> 8 & 15) << 8 | (uint32_t) (uint8_t) base_char(spread >> 16 & 15) << 16 | (uint32_t) (uint8_t) base_char(spread >> 24 & 15) << 24;
#endif
}
AGPU_HD uint32_t popcount32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
	return (uint32_t) __popc(x);
#else
	return (uint32_t) __builtin_popcount(x);
#endif
}

// `count` (1..16) bases of an aligned block: read positions read_position.., reference bytes at byte offset `reference_offset` of the genome.
// Adds the number of compared (non-N) bases and of mismatches.  Returns false if the chunk needs the base-by-base walk.
AGPU_HD bool compare_chunk(const SequenceRef& sequence, uint32_t read_position, const uint32_t* genome_words, uint64_t genome_last_word, uint64_t reference_offset, uint32_t count,
                           uint32_t& alignment_length, uint32_t& mismatches) {
	bool ok;
	const uint64_t codes = logical_codes16(sequence, read_position, ok);
	if (!ok) return false;
	const uint64_t word_index = reference_offset >> 2;
	const uint32_t shift = (uint32_t) (reference_offset & 3) << 3;
	// (the five loads do not depend on each other)
	const uint32_t g0 = genome_words[word_index];
	const uint32_t g1 = genome_words[word_index + 1 < genome_last_word ? word_index + 1 : genome_last_word];
	const uint32_t g2 = genome_words[word_index + 2 < genome_last_word ? word_index + 2 : genome_last_word];
	const uint32_t g3 = genome_words[word_index + 3 < genome_last_word ? word_index + 3 : genome_last_word];
	const uint32_t g4 = genome_words[word_index + 4 < genome_last_word ? word_index + 4 : genome_last_word];
#define AGPU_QUAD(q, LOW, HIGH) { \
		const uint32_t valid_bytes = count > 4u * q ? (count - 4u * q >= 4u ? 0xFFFFFFFFu : (1u << ((count - 4u * q) << 3)) - 1u) : 0u; \
		const uint32_t reference = (uint32_t) (((uint64_t) HIGH << 32 | LOW) >> shift); \
		const uint32_t four = (uint32_t) (codes >> (16 * q)) & 0xFFFFu; \
		uint32_t spread = (four & 0x00FFu) | (four & 0xFF00u) << 8; \
		spread = (spread | spread << 4) & 0x0F0F0F0Fu; \
		const uint32_t is_n = ((spread + 0x01010101u) & 0x10101010u) << 3;            /* 0x80 in every byte whose code is 15 */ \
		const uint32_t difference = characters_of_codes(spread) ^ reference; \
		const uint32_t differs = (((difference & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | difference) & 0x80808080u; \
		const uint32_t compared = ~is_n & 0x80808080u & valid_bytes; \
		alignment_length += popcount32(compared); \
		mismatches += popcount32(differs & compared); \
	}
	AGPU_QUAD(0, g0, g1) AGPU_QUAD(1, g1, g2) AGPU_QUAD(2, g2, g3) AGPU_QUAD(3, g3, g4)
#undef AGPU_QUAD
	return true;
}

// reference: count_mismatches + test_mismatch_probability, source/filter_mismatches.cpp:12-99.  Aligned blocks that lie inside the
// contig and the read are compared sixteen bases at a time; anything else falls back to the base-by-base walk (one aligned 32-bit
// genome word per four bases).
AGPU_HD bool has_too_many_mismatches(const BatchView& b, const GenomeView& genome, const FilterTables& t, uint64_t i, int slot, const SequenceRef& sequence, bool is_multimapper) {
	const uint32_t* cigar = cigar_of(b, slot, i); uint32_t n = b.cigar_count[slot][i];
	uint32_t contig = b.contig[slot][i];
	uint64_t contig_begin = genome.contig_offset[contig], contig_size = genome.contig_offset[contig + 1] - contig_begin;
	const uint32_t* genome_words = (const uint32_t*) genome.bases;
	bool forward = b.abits[slot][i] & ABIT_STRAND;
	uint32_t mismatches = 0, alignment_length = 0;
	int64_t reference_position = b.start[slot][i];
	uint32_t read_position = 0;
	uint64_t cached_word_index = ~0ull; uint32_t cached_word = 0;
	for (uint32_t c = 0; c < n; ++c) {
		uint32_t op = cigar[c] & 15, length = cigar[c] >> 4;
		switch (op) {
			case CIGAR_S: case CIGAR_H:
				read_position += length;
				if (!((c == 0 && !forward) || (c == n - 1 && forward))) mismatches++;
				break;
			case CIGAR_D:
				mismatches++;
				reference_position += length;
				break;
			case CIGAR_N:
				reference_position += length;
				break;
			case CIGAR_I:
				mismatches++;
				read_position += length;
				break;
			case CIGAR_M: case CIGAR_EQ: case CIGAR_X: {
				uint32_t k = 0;
				if (reference_position >= 0 && (uint64_t) reference_position + length <= contig_size && (uint64_t) read_position + length <= sequence.length) {
					const uint64_t genome_last_word = (genome.contig_offset[genome.n_contigs] - 1) >> 2;
					while (k < length) {
						const uint32_t count = length - k < 16 ? length - k : 16;
						if (!compare_chunk(sequence, read_position, genome_words, genome_last_word, contig_begin + (uint64_t) reference_position, count, alignment_length, mismatches)) break;
						reference_position += count; read_position += count; k += count;
					}
				}
				for (; k < length; ++k) {
					uint32_t code = (read_position < sequence.length) ? sequence.code(read_position) : 16;
					if (code != 15) { // 'N' bases are skipped
						char reference_base = '\0';
						if (reference_position >= 0 && (uint64_t) reference_position < contig_size) {
							uint64_t byte_index = contig_begin + (uint64_t) reference_position;
							if ((byte_index >> 2) != cached_word_index) { cached_word_index = byte_index >> 2; cached_word = genome_words[cached_word_index]; }
							reference_base = (char) (cached_word >> ((byte_index & 3) << 3));
						}
						char base = (code < 16) ? base_char(code) : '\0';
						if (base != reference_base) mismatches++;
						alignment_length++;
					}
					reference_position++;
					read_position++;
				}
				break;
			}
			default: break;
		}
	}
	if (is_multimapper) mismatches += 2;
	if (mismatches > alignment_length) return true;
	if (alignment_length > t.mismatch_max_length) return true; // unreachable for reads within the table; fail closed
	uint32_t bit = alignment_length * (alignment_length + 1) / 2 + mismatches;
	return (t.mismatch_verdict[bit >> 5] >> (bit & 31)) & 1;
}

AGPU_HD bool fails_mismatch_filter(const BatchView& b, const GenomeView& genome, const FilterTables& t, uint64_t i, const SequenceStage& stage) {
	bool multimapper = b.fbits[i] & FBIT_MULTIMAPPER;
	int other = (b.n_aln[i] == 2) ? MATE2 : SUPPLEMENTARY;
	bool viral1 = genome.contig_bits[b.contig[MATE1][i]] & CBIT_VIRAL, viral2 = genome.contig_bits[b.contig[other][i]] & CBIT_VIRAL;
	if (!viral1 && has_too_many_mismatches(b, genome, t, i, MATE1, sequence_of(b, MATE1, i, stage), multimapper && !viral2))
		return true;
	if (!viral2) {
		SequenceRef sequence;
		if (b.n_aln[i] == 2) sequence = sequence_of(b, MATE2, i, stage);
		else {
			sequence = sequence_of(b, SPLIT_READ, i, stage);
			sequence.reverse_complement = ((b.abits[SUPPLEMENTARY][i] ^ b.abits[SPLIT_READ][i]) & ABIT_STRAND) != 0;
		}
		if (has_too_many_mismatches(b, genome, t, i, other, sequence, multimapper && !viral1))
			return true;
	}
	return false;
}

// reference: kmer_to_int with k=3 (source/filter_mismappers.cpp:33-45): T=0, G=1, C=2, everything else 3
AGPU_HD uint32_t kmer_digit(uint32_t code) { return (0xFFFCFDEFu >> (code << 1)) & 3u; } // 2-bit table: code 8 (T) -> 0, 4 (G) -> 1, 2 (C) -> 2, others 3

AGPU_HD uint32_t kmer_threshold(const FilterTables& t, uint32_t length) {
	if (length < t.kmer_threshold_size) return t.kmer_threshold[length];
	return (uint32_t) ((double) ((float) length * t.max_kmer_content / 3.0f) + 0.5); // same operation order as the reference; only reached for absurd lengths
}

AGPU_HD bool looks_like_internal_tandem_duplication(const BatchView& b, const FilterTables& t, uint64_t i) { // source/filter_low_entropy.cpp:17-27
	if (b.n_aln[i] != 3) return false;
	bool split_forward = b.abits[SPLIT_READ][i] & ABIT_STRAND, supp_forward = b.abits[SUPPLEMENTARY][i] & ABIT_STRAND;
	if (split_forward != supp_forward || b.contig[SPLIT_READ][i] != b.contig[SUPPLEMENTARY][i]) return false;
	int32_t max_itd = (int32_t) t.max_itd_length;
	if (split_forward) return b.start[SPLIT_READ][i] < b.end[SUPPLEMENTARY][i] && b.start[SPLIT_READ][i] + max_itd >= b.end[SUPPLEMENTARY][i];
	return b.end[SPLIT_READ][i] > b.start[SUPPLEMENTARY][i] && b.end[SPLIT_READ][i] <= b.start[SUPPLEMENTARY][i] + max_itd;
}

// Counters of the 64 possible 3-mers, bit-sliced: plane p holds bit p of all 64 counters (counter k = bit k of every plane), so the
// counters live in sixteen 32-bit registers, need no LDS and no run-time indexing.  Adding one to counter k is a ripple carry over
// the planes (on average two planes); "is any counter >= T" is one bit-parallel comparison over all 64 counters.
const int KMER_PLANES = 8; // counts up to 255; a carry out of the last plane is remembered (it exceeds every threshold of a read of <= 1024 bases)
struct KmerCounters {
	uint64_t plane[KMER_PLANES];
	bool overflow;
	AGPU_HD void clear() {
		plane[0] = plane[1] = plane[2] = plane[3] = plane[4] = plane[5] = plane[6] = plane[7] = 0;
		overflow = false;
	}
	AGPU_HD void increment(uint64_t carry) { // carry = the counter's bit
#define AGPU_RIPPLE(p) if (carry) { const uint64_t next = plane[p] & carry; plane[p] ^= carry; carry = next; }
		AGPU_RIPPLE(0) AGPU_RIPPLE(1) AGPU_RIPPLE(2) AGPU_RIPPLE(3) AGPU_RIPPLE(4) AGPU_RIPPLE(5) AGPU_RIPPLE(6) AGPU_RIPPLE(7)
#undef AGPU_RIPPLE
		if (carry) overflow = true;
	}
	// bit k set iff counter k >= threshold
	AGPU_HD uint64_t at_least(uint32_t threshold) const {
		if (threshold >> KMER_PLANES) return overflow ? ~0ull : 0ull; // cannot be decided beyond the planes: only reachable for reads the kernel rejects
		uint64_t less = 0; // counter < threshold, decided from the least significant bit upwards
#define AGPU_COMPARE(p) less = (threshold >> p & 1u) ? (~plane[p] | less) : (less & ~plane[p]);
		AGPU_COMPARE(0) AGPU_COMPARE(1) AGPU_COMPARE(2) AGPU_COMPARE(3) AGPU_COMPARE(4) AGPU_COMPARE(5) AGPU_COMPARE(6) AGPU_COMPARE(7)
#undef AGPU_COMPARE
		return ~less;
	}
};

// reference: source/filter_low_entropy.cpp:33-101.  The reference counts an occurrence of a 3-mer only if it starts at or after the
// end of the previously counted occurrence of the same 3-mer; with k = 3 that means "not counted at either of the two preceding
// positions", so two registers replace its previous_kmer_pos array.  It tests the three counters of a 3-mer whenever that 3-mer is
// counted; counters only grow, so testing all counters of all counted 3-mers once at the end of the read gives the same verdict.
AGPU_HD bool has_low_entropy(const BatchView& b, const FilterTables& t, uint64_t i, const SequenceStage& stage) {
	const uint32_t K = 3, NONE = 64;
	for (int mate = MATE1; mate <= MATE2; ++mate) {
		SequenceRef sequence = sequence_of(b, mate, i, stage);
		uint32_t length = sequence.length;
		if (length < K) continue;
		const uint32_t* cigar = cigar_of(b, mate, i); uint32_t n = b.cigar_count[mate][i];
		uint32_t aligned_start1 = ((cigar[0] & 15) == CIGAR_S) ? cigar[0] >> 4 : 0;
		uint32_t aligned_end1 = length;
		if ((cigar[n - 1] & 15) == CIGAR_S) aligned_end1 -= cigar[n - 1] >> 4;
		uint32_t aligned_start2, aligned_end2;
		if (b.n_aln[i] == 3 && mate == SPLIT_READ) {
			const uint32_t* supp = cigar_of(b, SUPPLEMENTARY, i); uint32_t m = b.cigar_count[SUPPLEMENTARY][i];
			aligned_start2 = ((supp[0] & 15) == CIGAR_S) ? supp[0] >> 4 : 0;
			aligned_end2 = length;
			if ((supp[m - 1] & 15) == CIGAR_S) aligned_end2 -= supp[m - 1] >> 4;
			if ((b.abits[SUPPLEMENTARY][i] ^ b.abits[SPLIT_READ][i]) & ABIT_STRAND) {
				aligned_start2 = length - aligned_start2;
				aligned_end2 = length - aligned_end2;
				uint32_t swap = aligned_start2; aligned_start2 = aligned_end2; aligned_end2 = swap;
			}
		} else {
			aligned_start2 = aligned_start1;
			aligned_end2 = aligned_end1;
		}
		const uint32_t max_count = kmer_threshold(t, length);
		const uint32_t max_count_aligned1 = kmer_threshold(t, aligned_end1 - aligned_start1);
		const uint32_t max_count_aligned2 = kmer_threshold(t, aligned_end2 - aligned_start2);
		KmerCounters whole, aligned1, aligned2;
		whole.clear(); aligned1.clear(); aligned2.clear();
		uint64_t counted_kmers = 0; // 3-mers that were counted at least once
		uint32_t counted_previous = NONE, counted_before_previous = NONE; // 3-mers counted at position-1 / position-2
		uint32_t kmer = 0;
		const uint32_t last_position = length - K; // exclusive: the last k-mer is skipped, as in the reference (:77)
		const uint32_t n_words = (length + 7) >> 3;
		for (uint32_t w = 0; w < n_words; ++w) {
			uint32_t word = swap_nibbles(sequence.words[w]); // base j of this word now sits at bits 4j..4j+3
			uint32_t bases_here = (length - (w << 3)) < 8 ? (length - (w << 3)) : 8;
			for (uint32_t j = 0; j < bases_here; ++j) {
				kmer = ((kmer << 2) | kmer_digit((word >> (j << 2)) & 15)) & 63;
				uint32_t base_position = (w << 3) + j;
				if (base_position < 2) continue;
				uint32_t position = base_position - 2; // start of the 3-mer ending at this base
				if (position >= last_position) break;
				bool counted = kmer != counted_previous && kmer != counted_before_previous;
				counted_before_previous = counted_previous;
				counted_previous = counted ? kmer : NONE;
				if (!counted) continue;
				const uint64_t bit = 1ull << kmer;
				counted_kmers |= bit;
				whole.increment(bit);
				if (position + 1 >= aligned_start1 && position < aligned_end1) aligned1.increment(bit);
				if (position + 1 >= aligned_start2 && position < aligned_end2) aligned2.increment(bit);
			}
		}
		if ((whole.at_least(max_count) | aligned1.at_least(max_count_aligned1) | aligned2.at_least(max_count_aligned2)) & counted_kmers)
			return true;
		if (whole.overflow || aligned1.overflow || aligned2.overflow) return true;
	}
	return false;
}

// Stage-2 cascade without low_entropy for one fragment; `filter` is the state after stage group 1.  Returns the new filter id.
// first_hit receives the ordinal (0..7) of the stage that discarded the read, 9 if none (for the per-stage "remaining" counts).
AGPU_HD uint8_t read_filters_stage2(const BatchView& b, const AnnotationView& ann, const GenomeView& genome, const FilterTables& t, const uint8_t* enabled, uint64_t i, uint8_t filter, const SequenceStage& stage, uint32_t& first_hit) {
	first_hit = 9;
	if (filter != FILTER_none) return filter;
	AGPU_IDSET3(genes);
	int n_aln = b.n_aln[i];
	AGPU_UNROLL for (int s = 0; s < 3; ++s) { genes[s].clear(); if (s < n_aln) load_genes(b, s, i, genes[s]); }
	if (enabled[FILTER_read_through] && is_proximal_read_through(b, ann, t, i)) { filter = FILTER_read_through; first_hit = 0; }
	else if (enabled[FILTER_inconsistently_clipped] && is_inconsistently_clipped(b, i)) { filter = FILTER_inconsistently_clipped; first_hit = 1; }
	else if (enabled[FILTER_homopolymer] && has_homopolymer_at_breakpoint(b, ann, t, i, genes[SPLIT_READ], stage)) { filter = FILTER_homopolymer; first_hit = 2; }
	else if (enabled[FILTER_small_insert_size] && has_small_insert_size(b, i, 5)) { filter = FILTER_small_insert_size; first_hit = 3; }
	else if (enabled[FILTER_long_gap] && has_long_gap(b, i)) { filter = FILTER_long_gap; first_hit = 4; }
	else if (enabled[FILTER_same_gene] && is_same_gene_artifact(b, i, genes)) { filter = FILTER_same_gene; first_hit = 5; }
	else if (enabled[FILTER_hairpin] && is_hairpin(b, i, genes)) { filter = FILTER_hairpin; first_hit = 6; }
	else if (enabled[FILTER_mismatches] && fails_mismatch_filter(b, genome, t, i, stage)) { filter = FILTER_mismatches; first_hit = 7; }
	return filter;
}

// low_entropy for one fragment (the last read-level filter).  ITD-shaped reads are tested even if an earlier filter other than
// duplicates discarded them (source/filter_low_entropy.cpp:29-31).  was_unfiltered tells the caller whether the read counted as
// remaining before this filter.
AGPU_HD bool needs_low_entropy_test(const BatchView& b, const FilterTables& t, uint64_t i, uint8_t filter) {
	return filter == FILTER_none || (filter != FILTER_duplicates && looks_like_internal_tandem_duplication(b, t, i));
}

}

#endif
