// arriba_amd/csrc/device/in_vitro_host.hpp -- host side of filter_in_vitro: the expression threshold (reference: find_top_expressed_genes,
// source/filter_in_vitro.cpp:60-80).  Host code only, shared by the device library and the host stepping harness.
#ifndef AGPU_IN_VITRO_HOST_HPP
#define AGPU_IN_VITRO_HOST_HPP 1

#include <algorithm>
#include <cmath>
#include <stdint.h>
#include <vector>

namespace agpu {

// The reference puts the genes with at least one chimeric read into a vector, takes position floor(quantile * size) (float arithmetic,
// clamped to the last element) and partially sorts by (reads, gene id): the threshold is the count at that position of the ascending order.
inline uint32_t high_expression_threshold(const std::vector<uint32_t>& gene_read_count, float high_expression_quantile) {
	std::vector<uint32_t> counts;
	for (size_t gene = 0; gene < gene_read_count.size(); ++gene)
		if (gene_read_count[gene] > 0) counts.push_back(gene_read_count[gene]);
	if (counts.empty()) return 0;
	unsigned int quantile = static_cast<int>(floor(high_expression_quantile * counts.size()));
	if (quantile >= counts.size()) quantile = counts.size() - 1;
	std::nth_element(counts.begin(), counts.begin() + quantile, counts.end());
	return counts[quantile];
}

// recover_both_spliced (source/recover_both_spliced.cpp:152-170): histogram[r] = candidates with r supporting reads that would be recovered; the
// smallest number of supporting reads is raised until fewer than max_fusions_to_recover candidates with more reads than that are recovered
inline uint32_t both_spliced_min_supporting_reads(const std::vector<uint32_t>& histogram, uint32_t max_fusions_to_recover) {
	uint32_t would_be_recovered = 0;
	for (size_t reads = histogram.size(); reads-- > 0; ) {
		if (histogram[reads] == 0) continue;
		would_be_recovered += histogram[reads];
		if (would_be_recovered >= max_fusions_to_recover) return (uint32_t) reads + 1;
	}
	return 1;
}
const uint32_t BOTH_SPLICED_HISTOGRAM_BINS = 3 * 32768; // three 15-bit counters

}

#endif
