// arriba_amd/csrc/device/index_bins.hpp -- host-side construction of the coarse directory over the keys of a flattened interval index
// (FlatIndexView::bins, views.hpp): per contig, for every bin start j << shift up to the bin behind the last key, the index of the first
// key >= it.  index_lower_bound (annotate_core.hpp) then bisects only the keys of one bin.
#ifndef AGPU_INDEX_BINS_HPP
#define AGPU_INDEX_BINS_HPP 1

#include <stdint.h>
#include <vector>
#include "views.hpp"

namespace agpu {

// bins[0 .. n_contigs] = offset of the contig's directory inside `bins`, then the directories
inline void build_index_bins(uint32_t n_contigs, const uint32_t* contig_offset, const int32_t* keys, std::vector<uint32_t>& bins) {
	bins.assign((size_t) n_contigs + 1, 0);
	for (uint32_t contig = 0; contig < n_contigs; ++contig) {
		const uint32_t begin = contig_offset[contig], end = contig_offset[contig + 1];
		bins[contig] = (uint32_t) bins.size();
		if (end > begin) {
			const int32_t last_key = keys[end - 1];
			const uint32_t n_bins = last_key < 0 ? 1 : ((uint32_t) last_key >> INDEX_BIN_SHIFT) + 2;
			uint32_t k = begin;
			for (uint32_t j = 0; j < n_bins; ++j) {
				const int64_t bin_start = (int64_t) j << INDEX_BIN_SHIFT;
				while (k < end && (int64_t) keys[k] < bin_start) ++k;
				bins.push_back(k);
			}
		}
	}
	bins[n_contigs] = (uint32_t) bins.size();
}

}

#endif
