// arriba_amd/csrc/host/arriba_host.h -- host side of the MI355X-native fusion caller.
//
// The host driver reads the reference data (FASTA, GTF) and the STAR BAM once, restates the
// record classification of the reference's ingest (reference: source/read_chimeric_alignments.cpp)
// and packs everything into structure-of-arrays column buffers that are handed to the HIP
// kernels through the C ABI in include/arriba_gpu.h.  Nothing here does per-read filtering or
// candidate clustering -- that is the device's job.
#ifndef ARRIBA_HOST_H
#define ARRIBA_HOST_H 1

#include <cstdint>
#include "../../../include/arriba_host.h"
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace arriba {

// The processors this process may actually use at once: the hardware threads it sees, less what its CPU affinity and the CPU quota of its container (cgroup cpu.max /
// cfs_quota_us) allow.  A host with 256 hardware threads behind a quota of 16 CPUs runs 128 busy threads for 12 ms of every 100 ms and stalls them for the rest
// (profiles/r03g_probe.txt): every pool of worker threads of the host library sizes itself by this number.
unsigned int cpu_budget();
void limit_threads_of_this_thread(unsigned int n); // (0: no limit)
// called by a thread that only reads a file or formats rows: nice 10 (ARRIBA_WORKER_NICE, 0 = as its parent).  In a session with two lanes a dozen of these run beside the one
// thread that launches the kernels of the stages and waits for their words; with all at one priority that thread took 1.10 s for stages it does in 0.75 s alone, with four
// readers instead of eight 0.96 s (profiles/r04q_*) -- it should not queue behind memcpy.
void worker_thread_starts();

typedef int32_t position_t;
typedef uint16_t contig_t;

// filter ids: the position in the reference's registry (source/common.hpp:29-67) is the id
enum : uint8_t {
	FILTER_none = 0, FILTER_duplicates, FILTER_inconsistently_clipped, FILTER_homopolymer, FILTER_read_through, FILTER_same_gene,
	FILTER_small_insert_size, FILTER_long_gap, FILTER_hairpin, FILTER_multimappers, FILTER_mismatches, FILTER_mismappers,
	FILTER_relative_support, FILTER_intronic, FILTER_non_coding_neighbors, FILTER_intragenic_exonic, FILTER_internal_tandem_duplication,
	FILTER_min_support, FILTER_known_fusions, FILTER_spliced, FILTER_blacklist, FILTER_end_to_end, FILTER_in_vitro, FILTER_merge_adjacent,
	FILTER_select_best, FILTER_marginal_read_through, FILTER_short_anchor, FILTER_no_coverage, FILTER_many_spliced, FILTER_no_genomic_support,
	FILTER_uninteresting_contigs, FILTER_viral_contigs, FILTER_top_expressed_viral_contigs, FILTER_low_coverage_viral_contigs,
	FILTER_genomic_support, FILTER_isoforms, FILTER_low_entropy, FILTER_homologs, FILTER_COUNT
};
extern const char* const FILTER_NAMES[FILTER_COUNT];

const int COVERAGE_RESOLUTION = 20;         // reference: source/read_stats.hpp:15
const int MAX_SPLICE_SITE_DISTANCE = 2;     // reference: source/annotation.hpp:14

// CIGAR encoding is BAM's: length << 4 | op
enum { CIGAR_M = 0, CIGAR_I = 1, CIGAR_D = 2, CIGAR_N = 3, CIGAR_S = 4, CIGAR_H = 5, CIGAR_P = 6, CIGAR_EQ = 7, CIGAR_X = 8 };
inline uint32_t cigar_op(uint32_t c) { return c & 15; }
inline uint32_t cigar_len(uint32_t c) { return c >> 4; }
inline uint32_t cigar_make(uint32_t length, uint32_t op) { return length << 4 | op; }
inline bool cigar_consumes_query(uint32_t op) { return (0x3C1A7 >> (op << 1)) & 1; }
inline bool cigar_consumes_reference(uint32_t op) { return (0x3C1A7 >> (op << 1)) & 2; }

std::string remove_chr(std::string contig);                                                  // reference: source/common.hpp:74-80
bool is_interesting_contig(std::string contig, const std::string& interesting_contigs);     // reference: source/common.hpp:82-110

// ---- reference data ---------------------------------------------------------------------------

struct Contigs {
	std::map<std::string, contig_t> by_name;   // ordered like the reference's contigs_t (source/common.hpp:72)
	std::vector<std::string> original_names;
	contig_t add(const std::string& original_name); // returns the id (existing or new)
	size_t size() const { return by_name.size(); }
};

struct Assembly {
	std::vector<std::string> sequence;          // per contig id; empty = not loaded (uninteresting)
	bool has(contig_t contig) const { return contig < sequence.size() && !sequence[contig].empty(); }
};
// reference: source/assembly.cpp:28-58
void load_assembly(Assembly& assembly, const std::string& fasta_path, Contigs& contigs, const std::string& interesting_contigs);

struct GeneRecord {
	contig_t contig; position_t start, end; bool strand; // strand: true = forward
	std::string gene_id, name;
	int exonic_length;
	bool is_dummy, is_protein_coding;
};
struct ExonRecord {
	contig_t contig; position_t start, end; bool strand;
	int gene, transcript;
	int previous_exon, next_exon;               // exon ids, -1 = none
	position_t coding_region_start, coding_region_end;
};
struct TranscriptRecord {
	unsigned int id; std::string name;
	int first_exon, last_exon;
	unsigned int coding_length;
};
struct Annotation {
	std::vector<GeneRecord> genes;              // index == the reference's gene->id (list order, dummy genes appended last)
	std::vector<ExonRecord> exons;              // index == allocation order == canonical order inside exon sets (hazard H1)
	std::vector<TranscriptRecord> transcripts;
	std::unordered_map<std::string, int> gene_by_name;
	size_t real_genes = 0;                      // genes[0..real_genes) come from the GTF
};
// reference: source/annotation.cpp:161-377
// blacklist (keywords allowed in the second column) or known-fusions file -> rules; reference: source/filter_blacklisted_ranges.cpp:17-118, :244-264
void load_range_rules(const std::string& path, const Contigs& contigs, const Annotation& annotation, bool allow_keyword_in_second_column, std::vector<agpu_range_rule>& rules);
// tags (-t; reference: source/annotate_tags.cpp:11-44): lines of two items (no keywords) and a tag, looked up through 100 kb genome bins
struct TagRule { agpu_range_item first, second; std::string tag; };
struct Tags { std::vector<TagRule> rules; std::map<uint64_t, std::vector<uint32_t> > by_bin; bool empty() const { return by_bin.empty(); } };
void load_tags(const std::string& path, const Contigs& contigs, const Annotation& annotation, Tags& tags);
// protein domains (-p; reference: source/annotate_protein_domains.cpp:33-121): GFF3 records mapped to genes by id or name, indexed like the exons
struct ProteinDomain { contig_t contig; position_t start, end; bool strand; int gene; std::string name; };
void read_annotation_gtf(Annotation& annotation, const std::string& gtf_path, const std::string& gtf_features, Contigs& contigs, const Assembly& assembly);

// Flattened interval index (reference: source/annotation.t.hpp:25-45): per contig a sorted array of
// boundary keys (every feature.end and feature.start-1); bucket k lists the features that contain
// position keys[k], in ascending feature id.
struct FlatIndex {
	std::vector<uint32_t> contig_offset;        // [n_contigs+1] into keys
	std::vector<position_t> keys;
	std::vector<uint32_t> member_offset;        // [n_keys+1] into members
	std::vector<uint32_t> members;
	size_t n_contigs() const { return contig_offset.empty() ? 0 : contig_offset.size() - 1; }
	// first key >= position on the contig; returns the global key index or contig end
	uint32_t lower_bound(contig_t contig, position_t position) const;
	// (size_t, not contig_t: the index has as many slots as there are FEATURES -- the reference sizes it so, source/annotation.t.hpp:26 -- and a walk over all of them, as
	//  compute_exonic_length makes it, must not wrap at 65 536: with the 550 666 exons of a GENCODE-size annotation every real contig was visited nine times and every exonic length
	//  came out nine-fold -- found by the test on reference data of the size of hg38, round 6)
	uint32_t contig_begin(size_t contig) const { return contig_offset[contig]; }
	uint32_t contig_end(size_t contig) const { return contig_offset[contig + 1]; }
};
template <class Feature> void make_flat_index(const std::vector<Feature>& features, size_t n_contigs, FlatIndex& index);
// reference: source/arriba.cpp:166-184
void compute_exonic_length(Annotation& annotation, const FlatIndex& exon_index);

// structural variants from WGS (-d): Arriba's four-column format or VCF (BND, INV, DEL, DUP); reference: source/filter_genomic_support.cpp:15-165
void load_genomic_breakpoints(const std::string& path, const Contigs& contigs, std::vector<agpu_genomic_breakpoint>& variants);
void load_protein_domains(const std::string& path, const Contigs& contigs, const Annotation& annotation, std::vector<ProteinDomain>& domains, FlatIndex& index);

// host-side queries on the flat index (used by ingest and by host-only stages)
void get_annotation_by_coordinate(contig_t contig, position_t start, position_t end, std::vector<uint32_t>& result, const FlatIndex& index); // reference: source/annotation.t.hpp:55-101
bool is_breakpoint_spliced(int gene, bool direction_upstream, position_t breakpoint, const Annotation& annotation, const FlatIndex& exon_index); // reference: source/annotation.cpp:379-429
int get_spliced_distance(contig_t contig, position_t position1, position_t position2, int gene, const Annotation& annotation, const FlatIndex& exon_index); // reference: source/annotation.cpp:570-618

// ---- coverage -----------------------------------------------------------------------------------

struct Coverage { // reference: source/read_stats.hpp:17-27
	std::vector<std::vector<uint16_t> > coverage;
	std::vector<std::vector<uint8_t> > fragment_starts, fragment_ends;
	void resize(const Contigs& contigs, const Assembly& assembly);
	bool fragment_starts_here(contig_t contig, position_t start, position_t end) const;
	bool fragment_ends_here(contig_t contig, position_t start, position_t end) const;
	int get_coverage(contig_t contig, position_t position, bool direction_upstream) const;
};

// ---- packed batch -------------------------------------------------------------------------------

const unsigned MATE1 = 0, MATE2 = 1, SPLIT_READ = 1, SUPPLEMENTARY = 2;

// alignment bits (one byte per alignment slot)
enum : uint8_t { ABIT_STRAND = 1, ABIT_FIRST_IN_PAIR = 2, ABIT_SUPPLEMENTARY = 4, ABIT_EXONIC = 8, ABIT_PREDICTED_STRAND = 16, ABIT_PREDICTED_STRAND_AMBIGUOUS = 32 };
// fragment bits
enum : uint8_t { FBIT_SINGLE_END = 1, FBIT_MULTIMAPPER = 2, FBIT_DUPLICATE = 4 };

// Structure-of-arrays table of chimeric fragments in name order (index == name rank, hazard H3).
// Slot-major columns: column[slot][fragment].  Sequences are BAM 4-bit codes, two bases per byte,
// each sequence starting on a 4-byte boundary; only slots 0 and 1 carry a sequence.
struct Batch {
	size_t n = 0;
	std::vector<uint8_t> n_aln;                 // 2 = discordant mates, 3 = split read
	std::vector<uint8_t> fbits;
	std::vector<uint8_t> filter;
	std::vector<uint32_t> group;                // fragments sharing a read name (multimapper group) share a group id
	std::vector<contig_t> contig[3];
	std::vector<position_t> start[3], end[3];
	std::vector<uint8_t> abits[3];
	std::vector<uint32_t> cigar_offset[3];      // into cigar_pool
	std::vector<uint16_t> cigar_count[3];
	std::vector<uint32_t> cigar_pool;
	std::vector<uint32_t> seq_offset[2];        // in units of 4 bytes into seq_pool
	std::vector<uint32_t> seq_length[2];        // in bases
	std::vector<uint8_t> seq_pool;
	std::vector<uint32_t> name_offset;          // [n+1] into names ("QNAME,HI" as the reference keys its map)
	std::string names;
	std::string name(size_t i) const { return names.substr(name_offset[i], name_offset[i + 1] - name_offset[i]); }
	std::vector<uint32_t> cigar(unsigned slot, size_t i) const { return std::vector<uint32_t>(cigar_pool.begin() + cigar_offset[slot][i], cigar_pool.begin() + cigar_offset[slot][i] + cigar_count[slot][i]); }
	std::string sequence(unsigned slot, size_t i) const;
	void sequence_into(unsigned slot, size_t i, std::string& out) const; // the same into a string the caller keeps (no allocation once it has grown)
};

struct IngestOptions {
	std::string interesting_contigs = "1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 X Y AC_* NC_*";
	std::string viral_contigs = "AC_* NC_*";
	bool external_duplicate_marking = false;
	unsigned int max_itd_length = 100;
};

struct IngestResult {
	Batch batch;
	Coverage coverage;
	uint64_t mapped_reads = 0;
	std::vector<uint64_t> mapped_viral_reads_by_contig;
	unsigned int malformed_count = 0, missing_hi_tag = 0;
	uint64_t records = 0;
};

// Streaming source of uncompressed BAM bytes; returns the number of bytes delivered (0 = end of data).
struct ByteSource { virtual size_t read(uint8_t* buffer, size_t capacity) = 0; virtual ~ByteSource() {} };
ByteSource* open_bam_file(const std::string& path);               // BGZF/gzip or raw BAM, file or /dev/stdin
ByteSource* open_memory_source(const uint8_t* data, size_t size); // raw (already inflated) BAM stream

// The file side of the device ingest (agpu_ingest_*): container recognised from the one open stream, BAM header parsed, bytes handed on in pieces.
class BamFeed;
BamFeed* open_bam_feed(const std::string& path);
void close_bam_feed(BamFeed* feed);
uint64_t bam_feed_header(BamFeed* feed, std::vector<std::string>& target_names); // returns the size of the header = offset of the first record in the uncompressed stream
uint64_t bam_feed_take_part(BamFeed* feed, uint32_t part, uint32_t parts);            // the feed delivers only part `part` of `parts` of the records from now on; returns the offset of its first record in the stream
uint64_t bam_feed_size_hint(BamFeed* feed);                                       // expected size of the uncompressed stream, 0 = unknown
bool bam_feed_next(BamFeed* feed, uint8_t* buffer, size_t capacity, agpu_bgzf_block* blocks, uint32_t block_capacity, ahost_bam_piece& piece);

// reference: source/read_chimeric_alignments.cpp:560-773 with separate_chimeric_bam_file=false, is_rna_bam_file=true
// gene_index must be the index over the GTF genes (before dummy genes are added).
void read_chimeric_alignments(ByteSource& source, const Assembly& assembly, Contigs& contigs, const Annotation& annotation, const FlatIndex& gene_index, const IngestOptions& options, IngestResult& result);

}

#endif
