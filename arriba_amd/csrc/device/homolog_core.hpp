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
// assembly as substr(start, length) and reverse-complemented if the genes lie on different strands.
AGPU_HD bool genes_are_homologs(const AnnotationView& ann, const GenomeView& genome, const KmerIndexView& kmers, uint32_t gene1, uint32_t gene2, float max_identity_fraction) {
	AGPU_FP_AS_WRITTEN
	const uint32_t extended_kmer_length = 8;
	if (gene1 == gene2) return false;
	uint32_t small_gene = gene1, big_gene = gene2;
	if ((uint32_t) (ann.gene_end[small_gene] - ann.gene_start[small_gene]) > (uint32_t) (ann.gene_end[big_gene] - ann.gene_start[big_gene])) { small_gene = gene2; big_gene = gene1; }
	const uint32_t small_contig = ann.gene_contig[small_gene], big_contig = ann.gene_contig[big_gene];
	const int32_t small_start = ann.gene_start[small_gene], small_end = ann.gene_end[small_gene], big_start = ann.gene_start[big_gene], big_end = ann.gene_end[big_gene];
	if (small_contig == big_contig && ((small_start >= big_start && small_start <= big_end) || (small_end >= big_start && small_end <= big_end))) return false; // overlapping genes
	const uint32_t small_length = (uint32_t) (small_end - small_start);
	const uint64_t small_contig_size = genome.contig_offset[small_contig + 1] - genome.contig_offset[small_contig], big_contig_size = genome.contig_offset[big_contig + 1] - genome.contig_offset[big_contig];
	uint64_t size = small_length; // substr() is cut at the end of the contig
	if ((uint64_t) small_start >= small_contig_size) size = 0; else if ((uint64_t) small_start + size > small_contig_size) size = small_contig_size - (uint64_t) small_start;
	const char* small_bases = genome.bases + genome.contig_offset[small_contig] + small_start;
	const char* big_bases = genome.bases + genome.contig_offset[big_contig];
	const bool reverse = ((ann.gene_bits[small_gene] ^ ann.gene_bits[big_gene]) & GBIT_STRAND) != 0;
	const uint32_t table = big_contig < kmers.n_contigs ? kmers.contig_table[big_contig] : NO_KMER_TABLE;
	uint32_t matching_kmers = 0;
	for (uint64_t pos = 0; pos + 2 * KMER_LENGTH < size; pos += KMER_LENGTH) {
		if ((float) ((uint64_t) (matching_kmers * (uint32_t) KMER_LENGTH) + (size - pos)) < (float) small_length * max_identity_fraction) return false; // max_identity_fraction cannot be reached any more
		if (table == NO_KMER_TABLE) continue;
		uint32_t kmer = 0;
		for (uint32_t j = 0; j < (uint32_t) KMER_LENGTH; ++j) kmer = kmer << 2 | kmer_digit_of_char(reverse ? complement_of_base(small_bases[size - 1 - (pos + j)]) : small_bases[pos + j]);
		const uint32_t* offsets = kmers.offsets + (size_t) table * KMER_COUNT;
		uint32_t lo = offsets[kmer], hi = offsets[kmer + 1];
		const uint32_t bucket_end = hi;
		while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (kmers.positions[mid] < big_start) lo = mid + 1; else hi = mid; }
		for (uint32_t hit = lo; hit < bucket_end && kmers.positions[hit] <= big_end; ++hit) {
			const int32_t position = kmers.positions[hit];
			if (small_contig != big_contig || position < small_start || position > small_end) {
				bool extended_match = true; // strncmp over the next extended_kmer_length bases; behind the end of the contig the reference's string ends: a mismatch
				for (uint32_t j = 0; j < extended_kmer_length && extended_match; ++j) {
					const uint64_t big_at = (uint64_t) position + KMER_LENGTH + j, small_at = pos + KMER_LENGTH + j;
					const char small_base = reverse ? complement_of_base(small_bases[size - 1 - small_at]) : small_bases[small_at];
					extended_match = big_at < big_contig_size && big_bases[big_at] == small_base;
				}
				if (extended_match) {
					matching_kmers++;
					if ((float) (matching_kmers * (uint32_t) KMER_LENGTH) >= (float) small_length * max_identity_fraction) return true;
					break;
				}
			}
		}
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
