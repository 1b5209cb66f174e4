// arriba_amd/csrc/device/homolog_host.hpp -- the host side of filter_homologs (reference: source/filter_homologs.cpp:68-141): the list of
// unfiltered candidates in the reference's order, the gene pairs whose homology the elimination can ask for, and the elimination over a table
// of verdicts.  The verdicts themselves (homolog_core.hpp: genes_are_homologs) are computed by the caller, one thread per pair on the device.
#ifndef AGPU_HOMOLOG_HOST_HPP
#define AGPU_HOMOLOG_HOST_HPP 1

#include <algorithm>
#include <cstring>
#include <unordered_map>
#include <vector>
#include "homolog_core.hpp"

namespace agpu {

struct RemainingCandidate { uint32_t candidate, iteration_rank, gene1, gene2; int32_t breakpoint1, breakpoint2; uint32_t split_reads1, split_reads2, discordant_mates; float evalue; };

struct HomologElimination {
	std::vector<uint32_t> candidates;  // index into the device's candidate table, list order
	std::vector<uint8_t> filter;       // result: FILTER_none or FILTER_homologs per entry of `candidates`
	std::vector<uint64_t> pairs;       // gene pairs (gene_a << 32 | gene_b) to evaluate
	std::vector<uint32_t> gene1, gene2, split_reads1, split_reads2, discordant_mates, order;
	std::vector<int32_t> breakpoint1, breakpoint2;
	std::vector<float> evalue;
	CandidateTable compact;
	std::unordered_map<uint32_t, std::vector<uint32_t> > by_gene; // gene -> candidates of the list with that gene, ascending list position

	// `list` = the unfiltered candidates in any order
	void prepare(std::vector<RemainingCandidate>& list) {
		// the reference pushes to the front of its list while iterating fusions_t: descending iteration rank (hazard H2)
		std::sort(list.begin(), list.end(), [](const RemainingCandidate& x, const RemainingCandidate& y) { return x.iteration_rank > y.iteration_rank; });
		const uint32_t n = (uint32_t) list.size();
		candidates.resize(n); filter.assign(n, FILTER_none); gene1.resize(n); gene2.resize(n); split_reads1.resize(n); split_reads2.resize(n); discordant_mates.resize(n); order.resize(n);
		breakpoint1.resize(n); breakpoint2.resize(n); evalue.resize(n);
		for (uint32_t k = 0; k < n; ++k) {
			candidates[k] = list[k].candidate; gene1[k] = list[k].gene1; gene2[k] = list[k].gene2; breakpoint1[k] = list[k].breakpoint1; breakpoint2[k] = list[k].breakpoint2;
			split_reads1[k] = list[k].split_reads1; split_reads2[k] = list[k].split_reads2; discordant_mates[k] = list[k].discordant_mates; evalue[k] = list[k].evalue; order[k] = k;
		}
		memset(&compact, 0, sizeof(compact));
		compact.n = n; compact.gene1 = gene1.data(); compact.gene2 = gene2.data(); compact.breakpoint1 = breakpoint1.data(); compact.breakpoint2 = breakpoint2.data();
		compact.split_reads1 = split_reads1.data(); compact.split_reads2 = split_reads2.data(); compact.discordant_mates = discordant_mates.data(); compact.filter = filter.data();
		// every gene pair the elimination can ask about: a candidate's own genes, and the other genes of two candidates that have a gene in common.
		// Candidates are bucketed by gene so that only those sharing one are compared.
		by_gene.clear();
		for (uint32_t k = 0; k < n; ++k) { by_gene[gene1[k]].push_back(k); if (gene2[k] != gene1[k]) by_gene[gene2[k]].push_back(k); }
		std::unordered_map<uint64_t, uint32_t> seen;
		pairs.clear();
		for (uint32_t k = 0; k < n; ++k) add_pair(seen, gene1[k], gene2[k]);
		for (std::unordered_map<uint32_t, std::vector<uint32_t> >::const_iterator bucket = by_gene.begin(); bucket != by_gene.end(); ++bucket)
			for (size_t x = 0; x < bucket->second.size(); ++x)
				for (size_t y = 0; y < bucket->second.size(); ++y) {
					const uint32_t i = bucket->second[x], j = bucket->second[y];
					uint32_t homolog1, homolog2;
					if (i < j && homolog_partners(compact, i, j, homolog1, homolog2)) add_pair(seen, homolog1, homolog2);
				}
	}
	// `verdicts[k]` answers pairs[k]; returns the number of candidates still unfiltered, `filter` holds the outcome
	uint32_t run(const std::vector<uint8_t>& verdicts) {
		std::unordered_map<uint64_t, uint8_t> table;
		table.reserve(pairs.size() * 2);
		for (size_t k = 0; k < pairs.size(); ++k) table[pairs[k]] = verdicts[k];
		Lookup lookup = { table };
		// The reference compares every candidate with all behind it in the list (quadratic: 20 min for 31 k candidates); only candidates that share a
		// gene can be affected (homolog_partners), so the walk behind candidate i goes over the two gene buckets of i merged in list order instead.
		const uint32_t n = compact.n;
		for (uint32_t i = 0; i < n; ++i) {
			if (filter[i] != FILTER_none) continue;
			if (lookup(gene1[i], gene2[i])) { filter[i] = FILTER_homologs; continue; }
			const std::vector<uint32_t>& bucket_a = by_gene[gene1[i]];
			const std::vector<uint32_t>& bucket_b = gene2[i] != gene1[i] ? by_gene[gene2[i]] : bucket_a;
			size_t a = std::upper_bound(bucket_a.begin(), bucket_a.end(), i) - bucket_a.begin(), b = &bucket_b != &bucket_a ? std::upper_bound(bucket_b.begin(), bucket_b.end(), i) - bucket_b.begin() : bucket_b.size();
			while (a < bucket_a.size() || b < bucket_b.size()) {
				uint32_t j;
				if (b >= bucket_b.size() || (a < bucket_a.size() && bucket_a[a] <= bucket_b[b])) { j = bucket_a[a++]; if (b < bucket_b.size() && bucket_b[b] == j) ++b; } else j = bucket_b[b++];
				if (filter[j] != FILTER_none) continue;
				uint32_t homolog1, homolog2;
				if (!homolog_partners(compact, i, j, homolog1, homolog2) || !lookup(homolog1, homolog2)) continue;
				if (homolog_candidate_prevails(compact, evalue.data(), i, j)) filter[j] = FILTER_homologs;
				else { filter[i] = FILTER_homologs; break; }
			}
		}
		uint32_t kept = 0;
		for (uint32_t i = 0; i < n; ++i) kept += filter[i] == FILTER_none;
		return kept;
	}
private:
	struct Lookup {
		const std::unordered_map<uint64_t, uint8_t>& table;
		bool operator()(uint32_t gene_a, uint32_t gene_b) const { return table.at((uint64_t) gene_a << 32 | gene_b) != 0; } // at(): a pair that was not prepared is a bug, not a "no"
	};
	void add_pair(std::unordered_map<uint64_t, uint32_t>& seen, uint32_t gene_a, uint32_t gene_b) {
		const uint64_t key = (uint64_t) gene_a << 32 | gene_b;
		if (seen.emplace(key, (uint32_t) pairs.size()).second) pairs.push_back(key);
	}
};

}

#endif
