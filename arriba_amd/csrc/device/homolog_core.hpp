// arriba_amd/csrc/device/homolog_core.hpp -- filter_homologs (reference: source/filter_homologs.cpp:9-141, called at source/arriba.cpp:556-560):
// a candidate between homologous genes is discarded; of two candidates that share one gene and whose other genes are homologs, the one with
// the poorer support is discarded.  The reference walks the unfiltered candidates in REVERSE iteration order of fusions_t (push_front, hazard H2)
// and compares each with all behind it, filters changing on the way: that elimination is sequential.  The homology verdicts it needs are pure
// functions of a gene pair (k-mer index, genome) and are what costs time: the device evaluates them in parallel for every gene pair the
// elimination can ask for, the elimination itself then runs over a table of verdicts.
#ifndef AGPU_HOMOLOG_CORE_HPP
#define AGPU_HOMOLOG_CORE_HPP 1

#include "mismapper_core.hpp"

namespace agpu {

const uint8_t FILTER_homologs = 37; // source/common.hpp:29-67

AGPU_HD char complement_of_base(char base) { return base == 'A' ? 'T' : base == 'T' ? 'A' : base == 'C' ? 'G' : base == 'G' ? 'C' : base; } // the genome is upper case

// reference: is_homolog (:9-66).  Gene length = end - start (source/common.hpp:126); the sequence of the smaller gene is taken from the
// assembly as substr(start, length) and reverse-complemented if the genes lie on different strands.  Split into the set-up of a gene pair, the question asked at
// one position of the smaller gene (does its k-mer have a hit in the bigger gene whose next eight bases match, too? -- independent of the other positions) and
// the reference's walk over the answers in position order with its two early exits; a thread does all three in a loop, a wavefront asks 64 positions at once.
struct HomologPair {
	const char* small_bases; const char* big_bases;
	uint64_t size, big_contig_size;      // bases of the smaller gene that exist (substr is cut at the end of the contig); size of the bigger gene's contig
	uint32_t small_length, table;
	int32_t small_start, small_end, big_start, big_end;
	bool reverse, same_contig;
};
// false: the verdict is "no" without looking at any base (the same gene, overlapping genes)
AGPU_HD bool homolog_pair_setup(const AnnotationView& ann, const GenomeView& genome, const KmerIndexView& kmers, uint32_t gene1, uint32_t gene2, HomologPair& p) {
	if (gene1 == gene2) return false;
	uint32_t small_gene = gene1, big_gene = gene2;
	if ((uint32_t) (ann.gene_end[small_gene] - ann.gene_start[small_gene]) > (uint32_t) (ann.gene_end[big_gene] - ann.gene_start[big_gene])) { small_gene = gene2; big_gene = gene1; }
	const uint32_t small_contig = ann.gene_contig[small_gene], big_contig = ann.gene_contig[big_gene];
	p.small_start = ann.gene_start[small_gene]; p.small_end = ann.gene_end[small_gene]; p.big_start = ann.gene_start[big_gene]; p.big_end = ann.gene_end[big_gene];
	if (small_contig == big_contig && ((p.small_start >= p.big_start && p.small_start <= p.big_end) || (p.small_end >= p.big_start && p.small_end <= p.big_end))) return false; // overlapping genes
	p.small_length = (uint32_t) (p.small_end - p.small_start);
	const uint64_t small_contig_size = genome.contig_offset[small_contig + 1] - genome.contig_offset[small_contig];
	p.big_contig_size = genome.contig_offset[big_contig + 1] - genome.contig_offset[big_contig];
	p.size = p.small_length; // substr() is cut at the end of the contig
	if ((uint64_t) p.small_start >= small_contig_size) p.size = 0; else if ((uint64_t) p.small_start + p.size > small_contig_size) p.size = small_contig_size - (uint64_t) p.small_start;
	p.small_bases = genome.bases + genome.contig_offset[small_contig] + p.small_start;
	p.big_bases = genome.bases + genome.contig_offset[big_contig];
	p.reverse = ((ann.gene_bits[small_gene] ^ ann.gene_bits[big_gene]) & GBIT_STRAND) != 0;
	p.same_contig = small_contig == big_contig;
	p.table = big_contig < kmers.n_contigs ? kmers.contig_table[big_contig] : NO_KMER_TABLE;
	return true;
}
// the k-mer of the smaller gene at `pos` (pos + 2 * KMER_LENGTH < p.size): a hit inside the bigger gene (outside the smaller one) whose next eight bases match as well?
AGPU_HD bool homolog_position_matches(const HomologPair& p, const KmerIndexView& kmers, uint64_t pos) {
	const uint32_t extended_kmer_length = 8;
	if (p.table == NO_KMER_TABLE) return false;
	uint32_t kmer = 0;
	for (uint32_t j = 0; j < (uint32_t) KMER_LENGTH; ++j) kmer = kmer << 2 | kmer_digit_of_char(p.reverse ? complement_of_base(p.small_bases[p.size - 1 - (pos + j)]) : p.small_bases[pos + j]);
	const uint32_t* offsets = kmers.offsets + (size_t) p.table * KMER_COUNT;
	uint32_t lo = offsets[kmer], hi = offsets[kmer + 1];
	const uint32_t bucket_end = hi;
	while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (kmers.positions[mid] < p.big_start) lo = mid + 1; else hi = mid; }
	for (uint32_t hit = lo; hit < bucket_end && kmers.positions[hit] <= p.big_end; ++hit) {
		const int32_t position = kmers.positions[hit];
		if (!p.same_contig || position < p.small_start || position > p.small_end) {
			bool extended_match = true; // strncmp over the next extended_kmer_length bases; behind the end of the contig the reference's string ends: a mismatch
			for (uint32_t j = 0; j < extended_kmer_length && extended_match; ++j) {
				const uint64_t big_at = (uint64_t) position + KMER_LENGTH + j, small_at = pos + KMER_LENGTH + j;
				const char small_base = p.reverse ? complement_of_base(p.small_bases[p.size - 1 - small_at]) : p.small_bases[small_at];
				extended_match = big_at < p.big_contig_size && p.big_bases[big_at] == small_base;
			}
			if (extended_match) return true; // (the reference counts the k-mer once and goes to the next position)
		}
	}
	return false;
}
// the reference's loop at position `pos`, given the answer there: -1 = "not homologs" (max_identity_fraction cannot be reached any more), 1 = "homologs", 0 = go on
AGPU_HD int homolog_walk(const HomologPair& p, float max_identity_fraction, uint64_t pos, bool matches, uint32_t& matching_kmers) {
	AGPU_FP_AS_WRITTEN
	if ((float) ((uint64_t) (matching_kmers * (uint32_t) KMER_LENGTH) + (p.size - pos)) < (float) p.small_length * max_identity_fraction) return -1;
	if (matches) {
		matching_kmers++;
		if ((float) (matching_kmers * (uint32_t) KMER_LENGTH) >= (float) p.small_length * max_identity_fraction) return 1;
	}
	return 0;
}
AGPU_HD bool genes_are_homologs(const AnnotationView& ann, const GenomeView& genome, const KmerIndexView& kmers, uint32_t gene1, uint32_t gene2, float max_identity_fraction) {
	AGPU_FP_AS_WRITTEN
	HomologPair p;
	if (!homolog_pair_setup(ann, genome, kmers, gene1, gene2, p)) return false;
	uint32_t matching_kmers = 0;
	for (uint64_t pos = 0; pos + 2 * KMER_LENGTH < p.size; pos += KMER_LENGTH) {
		// (the exit "cannot be reached any more" comes before the look-up in the reference: no look-up behind it here either)
		if ((float) ((uint64_t) (matching_kmers * (uint32_t) KMER_LENGTH) + (p.size - pos)) < (float) p.small_length * max_identity_fraction) return false;
		const int verdict = homolog_walk(p, max_identity_fraction, pos, homolog_position_matches(p, kmers, pos), matching_kmers);
		if (verdict != 0) return verdict > 0;
	}
	return false;
}

// which genes of two candidates that share a gene have to be homologs for one of them to go? (:88-104)  false = no gene in common
AGPU_HD bool homolog_partners(const CandidateTable& t, uint32_t fusion, uint32_t other, uint32_t& homolog1, uint32_t& homolog2) {
	if (t.gene1[fusion] == t.gene1[other] && t.breakpoint2[fusion] != t.breakpoint2[other]) { homolog1 = t.gene2[fusion]; homolog2 = t.gene2[other]; return true; }
	if (t.gene1[fusion] == t.gene2[other] && t.breakpoint2[fusion] != t.breakpoint1[other]) { homolog1 = t.gene2[fusion]; homolog2 = t.gene1[other]; return true; }
	if (t.gene2[fusion] == t.gene1[other] && t.breakpoint1[fusion] != t.breakpoint2[other]) { homolog1 = t.gene1[fusion]; homolog2 = t.gene2[other]; return true; }
	if (t.gene2[fusion] == t.gene2[other] && t.breakpoint1[fusion] != t.breakpoint1[other]) { homolog1 = t.gene1[fusion]; homolog2 = t.gene1[other]; return true; }
	return false;
}
AGPU_HD uint32_t homolog_anchor_count(const CandidateTable& t, uint32_t c) { return (t.split_reads1[c] > 0) + (t.split_reads2[c] > 0) + (t.discordant_mates[c] > 0); }

// of two candidates that share a gene and whose other genes are homologs: does `fusion` (earlier in the list) stay? (:106-127) more kinds of
// supporting reads, then more reads, then the better e-value
AGPU_HD bool homolog_candidate_prevails(const CandidateTable& t, const float* evalue, uint32_t fusion, uint32_t other) {
	const uint32_t anchor1 = homolog_anchor_count(t, fusion), anchor2 = homolog_anchor_count(t, other);
	const uint32_t support1 = t.split_reads1[fusion] + t.split_reads2[fusion] + t.discordant_mates[fusion], support2 = t.split_reads1[other] + t.split_reads2[other] + t.discordant_mates[other];
	return anchor1 > anchor2 || (anchor1 == anchor2 && support1 > support2) || (anchor1 == anchor2 && support1 == support2 && evalue[fusion] <= evalue[other]);
}

}

#endif
