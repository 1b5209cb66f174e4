// arriba_amd/csrc/device/range_rule_host.hpp -- the genome-bin index over the lines of a blacklist / known-fusions file, built on the host
// (reference: source/filter_blacklisted_ranges.cpp:268-274 -- bins of both items, each widened by max_mate_gap --, source/recover_known_fusions.cpp:32-36 -- not widened).
#ifndef AGPU_RANGE_RULE_HOST_HPP
#define AGPU_RANGE_RULE_HOST_HPP 1

#include <algorithm>
#include <utility>
#include <vector>
#include "range_rule_core.hpp"

namespace agpu {

struct RangeRuleBins { std::vector<uint64_t> bin_keys; std::vector<uint32_t> bin_offset, bin_rules; };

inline void build_range_rule_bins(const agpu_range_rule* rules, uint32_t n_rules, int32_t padding, RangeRuleBins& out) {
	std::vector<std::pair<uint64_t, uint32_t> > entries;
	for (uint32_t r = 0; r < n_rules; ++r) {
		const agpu_range_item* items[2] = { &rules[r].first, &rules[r].second };
		for (int k = 0; k < 2; ++k)
			if (items[k]->type == AGPU_RULE_POSITION || items[k]->type == AGPU_RULE_RANGE || items[k]->type == AGPU_RULE_GENE)
				for (int32_t bin = genome_bin_first(items[k]->start - padding); bin <= genome_bin_last(items[k]->end + padding); ++bin)
					entries.push_back(std::make_pair(genome_bin_key(items[k]->contig, bin), r));
	}
	std::sort(entries.begin(), entries.end());
	entries.erase(std::unique(entries.begin(), entries.end()), entries.end());
	out.bin_keys.clear(); out.bin_offset.clear(); out.bin_rules.clear();
	for (size_t e = 0; e < entries.size(); ++e) {
		if (e == 0 || entries[e].first != entries[e - 1].first) { out.bin_keys.push_back(entries[e].first); out.bin_offset.push_back((uint32_t) e); }
		out.bin_rules.push_back(entries[e].second);
	}
	out.bin_offset.push_back((uint32_t) entries.size());
}

}

#endif
