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

}

#endif
