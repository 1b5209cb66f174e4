// arriba_amd/csrc/device/evalue_host.hpp -- host side of the e-value stage: the sample-wide covariates with the reference's
// small-sample fallbacks and the pow() factor tables (reference: source/filter_relative_support.cpp:59-176).  Host code only;
// the tables are computed with the host's libm so that they are bit-identical to what the reference computes on the same machine.
#ifndef AGPU_EVALUE_HOST_HPP
#define AGPU_EVALUE_HOST_HPP 1

#include <algorithm>
#include <cmath>
#include <vector>
#include "evalue_core.hpp"

namespace agpu {

// counters[EG_*] as counted over the candidates; genes_with_(read_through_)fusions = sizes of the two gene sets of :114-126
inline EvalueGlobals make_evalue_globals(const unsigned int* counters, unsigned int genes_with_fusions, unsigned int genes_with_read_through_fusions) {
	EvalueGlobals g;
	unsigned int spliced_breakpoints = counters[EG_SPLICED], exonic_breakpoints = counters[EG_EXONIC], intronic_breakpoints = counters[EG_INTRONIC], exonic_intronic_breakpoints = counters[EG_MIXED];
	if (spliced_breakpoints + exonic_breakpoints + intronic_breakpoints + exonic_intronic_breakpoints < 100 ||
	    spliced_breakpoints == 0 || exonic_breakpoints == 0 || intronic_breakpoints == 0 || exonic_intronic_breakpoints == 0) {
		spliced_breakpoints = 10; exonic_breakpoints = 65; intronic_breakpoints = 10; exonic_intronic_breakpoints = 15;
	}
	unsigned int intragenic_duplications = counters[EG_DUPLICATIONS], intragenic_inversions = counters[EG_INVERSIONS];
	if (intragenic_inversions + intragenic_duplications < 100) { intragenic_inversions = 1; intragenic_duplications = 1; }
	unsigned int spliced_events_in_same_gene = counters[EG_SPLICED_SAME_GENE], spliced_events_in_different_genes = counters[EG_SPLICED_DIFFERENT_GENES];
	if (spliced_events_in_same_gene + spliced_events_in_different_genes < 100) { spliced_events_in_same_gene = 0; spliced_events_in_different_genes = 100; }
	float fraction_of_genes_with_read_through_fusions = (genes_with_fusions == 0) ? 0 : 1.0 * genes_with_read_through_fusions / genes_with_fusions;
	g.spliced_breakpoints = spliced_breakpoints; g.exonic_breakpoints = exonic_breakpoints; g.intronic_breakpoints = intronic_breakpoints; g.exonic_intronic_breakpoints = exonic_intronic_breakpoints;
	g.intragenic_duplications = intragenic_duplications; g.intragenic_inversions = intragenic_inversions;
	g.intragenic_scale = 2.0 / (intragenic_duplications + intragenic_inversions);
	g.intragenic_excess = std::max(1.0, spliced_events_in_same_gene / 0.25 / spliced_events_in_different_genes);
	g.location_scale = 4.0 / (spliced_breakpoints + exonic_breakpoints + intronic_breakpoints + exonic_intronic_breakpoints);
	g.read_through_penalty_applies = fraction_of_genes_with_read_through_fusions > 0.25;
	g.read_through_penalty = 1 + pow((fraction_of_genes_with_read_through_fusions - 0.25) * 20, 2);
	return g;
}

struct EvalueHostTables {
	std::vector<double> support_scale, intragenic_support, intergenic_support; // indexed by supporting reads, 0..max_support
	std::vector<double> distances;                                             // [1000 | 400 | 400000 | 400000]
	static const size_t DISTANCE_TABLE_SIZE = 1000 + 400 + 2 * 400000;

	void build_support_tables(unsigned long int mapped_reads, unsigned int max_support) { // :143, :161, :176
		support_scale.resize(max_support + 1); intragenic_support.resize(max_support + 1); intergenic_support.resize(max_support + 1);
		for (unsigned int supporting_reads = 0; supporting_reads <= max_support; ++supporting_reads) {
			support_scale[supporting_reads] = std::max(1.0, mapped_reads / 20000000.0 * pow(0.02, supporting_reads - 2)); // unsigned wrap for 0 and 1, as in the reference (hazard H14)
			intragenic_support[supporting_reads] = pow(supporting_reads - 0.42, -2.11) * pow(10, -1.11);
			intergenic_support[supporting_reads] = pow(supporting_reads - 0.73, -2.28) * pow(10, -1.75);
		}
	}
	void build_distance_tables() { // :164-167, :180-183
		distances.assign(DISTANCE_TABLE_SIZE, 0.0);
		double* distance_1000 = distances.data(), *distance_400 = distance_1000 + 1000, *read_through = distance_400 + 400, *proximal = read_through + 400000;
		for (int d = 400; d < 1000; ++d) distance_1000[d] = pow(d / 1000.0, -2);
		for (int d = 1; d < 400; ++d) distance_400[d] = pow(d / 400.0, -4.58);
		for (int d = 1; d < 400000; ++d) { read_through[d] = pow(d / 400000.0, -0.63); proximal[d] = pow(d / 400000.0, -1.53); }
	}
};

inline EvalueTables evalue_table_view(const double* support_scale, const double* intragenic_support, const double* intergenic_support, unsigned int max_support, const double* distances) {
	EvalueTables t;
	t.support_scale = support_scale; t.intragenic_support = intragenic_support; t.intergenic_support = intergenic_support; t.max_support = max_support;
	t.distance_1000 = distances; t.distance_400 = t.distance_1000 + 1000; t.read_through_distance = t.distance_400 + 400; t.proximal_distance = t.read_through_distance + 400000;
	return t;
}

}

#endif
