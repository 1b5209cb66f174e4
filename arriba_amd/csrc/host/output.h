// arriba_amd/csrc/host/output.h -- the candidate table as the output writer reads it (filled from the device's result getters)
#ifndef ARRIBA_HOST_OUTPUT_H
#define ARRIBA_HOST_OUTPUT_H 1

#include "arriba_host.h"

namespace arriba {

struct FusionTable {
	uint32_t n_candidates;
	const uint32_t* gene1; const uint32_t* gene2; const uint32_t* contigs; const int32_t* breakpoint1; const int32_t* breakpoint2; const uint32_t* flags; const uint8_t* filter;
	const uint32_t* split_reads1; const uint32_t* split_reads2; const uint32_t* discordant_mates;
	const uint64_t* list_offset; const uint32_t* read_lists;
	const float* evalue; const uint8_t* confidence; const uint32_t* iteration_rank;
	const uint8_t* read_filter;
	const int32_t* closest_genomic_breakpoint1; const int32_t* closest_genomic_breakpoint2; // NULL = none
	uint32_t n_genes; const uint16_t* gene_contig; const int32_t* gene_start; const int32_t* gene_end; // GTF genes + dummy genes (agpu_get_gene_table)
};
struct OutputExtras {
	const Tags* tags; const std::vector<ProteinDomain>* protein_domains; const FlatIndex* protein_domain_index; int max_mate_gap; bool fill_sequence_gaps; // -t, -p (NULL = not given), -I
	// one sample over several ranks: the rows part, part + parts, ... of the file are formatted and appended to *text_of_part (the header line in front of them
	// for part 0); nothing is written to `path`.  The rows are independent of each other; whoever holds the texts of all parts interleaves them.
	unsigned part = 0, parts = 1; std::string* text_of_part = NULL;
};
// reference: write_fusions_to_file (source/output_fusions.cpp:1043-1261)
void write_fusions_to_file(const Annotation& annotation, const FlatIndex& exon_index, const Contigs& contigs, const Assembly& assembly, const Coverage& coverage, const Batch* batch, const FusionTable& table,
                           const std::string& path, bool write_discarded, bool print_extra_info, unsigned max_itd_length, const OutputExtras& extras);

}

#endif
