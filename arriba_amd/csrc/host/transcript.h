// arriba_amd/csrc/host/transcript.h -- fusion transcript, best fitting annotated transcripts, peptide and reading frame (see transcript.cpp)
#ifndef ARRIBA_HOST_TRANSCRIPT_H
#define ARRIBA_HOST_TRANSCRIPT_H 1

#include "arriba_host.h"
#include <atomic>

namespace arriba {

extern std::atomic<long long> transcript_profile_ns[4]; // ARRIBA_WRITER_PROFILE: nanoseconds of all threads in the pile-ups, their columns, the consensus (with the columns), the rest
extern thread_local std::string* transcript_warnings; // where the warnings of the functions below go while a row is formatted (NULL: stderr)

struct TranscriptInput { const Batch& batch; const uint8_t* read_filter; const Assembly& assembly; const Annotation& annotation; const FlatIndex& exon_index; };
struct FusionEvent { // one candidate with its read lists
	contig_t contig_of_gene1, contig_of_gene2; // gene->contig: where the reference bases of the pileups are looked up
	position_t breakpoint1, breakpoint2;
	bool upstream1, upstream2, predicted_strand1, predicted_strand2, strands_ambiguous, transcript_start_gene1, transcript_start_ambiguous;
	const uint32_t* split_read1_list; const uint32_t* split_read2_list; const uint32_t* discordant_mate_list;
	uint32_t n_split_reads1, n_split_reads2, n_discordant_mates;
};
struct PeptideGenes { contig_t contig_5, contig_3; bool forward_5, forward_3, dummy_5, dummy_3, predicted_strand_3; };

void fusion_transcript_sequence(const TranscriptInput& in, const FusionEvent& fusion, std::string& sequence, std::vector<position_t>& positions);
void best_fitting_transcripts(const TranscriptInput& in, const std::string& sequence, const std::vector<position_t>& transcribed_bases, int gene, bool gene_is_dummy, contig_t gene_contig, bool gene_forward,
                              bool strand, bool strand_ambiguous, int which_end, std::vector<int>& best);
void fill_gaps_in_fusion_transcript(const TranscriptInput& in, std::string& sequence, std::vector<position_t>& positions, int transcript_5, int transcript_3, bool strand_5, bool strand_3, bool is_internal_tandem_duplication);
std::string fusion_peptide_sequence(const TranscriptInput& in, const std::string& sequence, const std::vector<position_t>& positions, const PeptideGenes& genes, int transcript_5, int transcript_3);
std::string reading_frame_verdict(const std::string& peptide);

}

#endif
