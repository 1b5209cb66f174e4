// tools/gen_synth.hpp -- synthetic benchmark/test input generator (test + bench tooling, not product code).
//
// Produces the inputs SURVEY.md section 8(d) specifies for BASELINE.json's synthetic configs:
// a toy/scaled genome (FASTA), a GENCODE-like annotation (GTF) and a STAR-"WithinBAM"-like
// alignment stream (BAM): split-read triplets (primary with SA tag + mate + hard-clipped
// supplementary), discordant pairs, read-through pairs, ordinary proper pairs, PCR duplicates,
// multimappers (HI tag), low-complexity / homopolymer breakpoints, paralogous genes (for the
// mismapper re-alignment), internal tandem duplications, viral and uninteresting contigs.
// The record layout follows the SAM/BAM specification; what the reference consumes from each
// record is listed in SURVEY.md appendix A.2/A.7.
#ifndef GEN_SYNTH_HPP
#define GEN_SYNTH_HPP 1

#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace synth {

struct Rng;

struct Config {
	uint64_t seed = 1;
	uint64_t read_seed = 0; // non-zero: the reads are drawn from this seed while genome and annotation come from `seed` (shards of one sample)
	int contigs = 6;                 // main contigs named 1..N
	int contig_length = 400000;
	double genes_per_mb = 60;
	int gene_stack = 0;              // > 0: 20 % of the genes get 2..gene_stack overlapping copies on the same locus (large gene sets)
	int read_length = 100;
	long fragments = 20000;          // chimeric fragments (split-read triplets, discordant pairs, read-through)
	double normal_multiplier = 1.0;  // ordinary proper pairs per chimeric fragment
	int junctions = 200;             // "true" fusion junctions with Zipf(1.2) support
	double frac_split = 0.55, frac_discordant = 0.35, frac_read_through = 0.10;
	double frac_noise = 0.35;        // of split/discordant fragments: random (non-recurrent) junctions
	double frac_same_gene = 0.06;    // of noise: both sides in the same gene
	double frac_duplicates = 0.30;
	double frac_multimappers = 0.03;
	double frac_low_complexity_junctions = 0.04;
	double frac_paralog_genes = 0.05;
	double frac_itd = 0.01;          // of ordinary pairs: internal tandem duplication reads
	int itd_hotspots = 0;            // > 0: that many recurrent internal tandem duplications (fixed position and length inside a coding exon), each
	double frac_itd_hotspot = 0.02;  //      hit by this fraction of the ordinary pairs divided among them (no random numbers are drawn when 0)
	int homolog_families = 0;        // > 0: that many gene pairs with one locus copied over the other, with junctions to a common partner and between them
	double frac_malformed = 0.003;
	double error_rate = 0.005, high_error_fraction = 0.01, high_error_rate = 0.08;
	int clip_min = 12, clip_max = 60;
	double frac_clip_from_partner = 0.0; // mismapper stress: clipped segment copied from the partner gene
	double frac_indels = 0.0;        // of the mates / discordant mates: a 1-3 base insertion or deletion inside the first long aligned segment (no random numbers are drawn when 0)
	double frac_non_template = 0.0;  // of the junctions: 1-3 bases that belong to neither gene between the two parts of every split read (decided by the breakpoints: no random numbers are drawn)
	bool soft_clip_supplementary = false; // supplementary alignments with soft clips and the whole read sequence (STAR --chimOutType WithinBAM SoftClip) instead of hard clips
	double frac_n_bases = 0.0;       // of the bases of every read: N (drawn behind the sequencing errors; no random numbers are drawn when 0)
	int name_length = 0;           // read names padded to this many characters (11 without: "r%010d")
	int bgzf_level = 0;            // 0: stored BGZF blocks (STAR --outBAMcompression 0); 1-9: deflated at that zlib level (write_bam_segmented)
	double frac_missing_hi = 0.0;    // of the multi-mapping fragments: the secondary alignments are written without the HI tag (STAR without --outSAMattributes HI; no random numbers are drawn when 0)
	bool single_end = false;         // a single-end library: of every fragment only the records of one read (a split read keeps its supplementary alignment), no pairing flags
	bool shuffle_names = false;      // emit records so that BAM order != sorted name order
	bool separate_mates = false;     // put other records between the two mates of a pair
	bool viral = true;
	bool stranded = false;           // library strandedness (read1 == transcript strand)
};

struct Exon { int start, end; }; // 0-based inclusive
struct Transcript {
	std::string id;
	std::vector<Exon> exons;     // ascending by coordinate
	int cds_start = -1, cds_end = -1; // genomic, inclusive; -1 = non-coding
};
struct Gene {
	int contig;
	int start, end;
	bool plus;
	std::string id, name;
	std::vector<Transcript> transcripts;
};

// sink receives the uncompressed BAM byte stream (header first, then records)
typedef std::function<void(const uint8_t*, size_t)> ByteSink;

class Generator {
public:
	explicit Generator(const Config& config);
	void build_reference();                       // genome + genes + junction table
	void write_fasta(const std::string& path) const;
	void write_gtf(const std::string& path) const;
	void write_rule_files(const std::string& blacklist_path, const std::string& known_fusions_path) const; // a blacklist and a known-fusions file derived from the junction table
	void write_bam(const std::string& path);      // BGZF with stored (uncompressed) blocks, like STAR --outBAMcompression 0
	void stream_bam(const ByteSink& sink);        // raw (un-BGZF'd) BAM stream for in-memory consumers
	void write_bam_segmented(const std::string& path, unsigned int n_threads, bool bgzf); // large samples: segments of the fragments made by several threads
	const std::vector<std::string>& contig_names() const { return contig_names_; }
	const std::vector<std::string>& contig_sequences() const { return contig_sequences_; }
	const std::vector<Gene>& genes() const { return genes_; }
	long records_written() const { return records_written_; }
	struct Impl;
private:
	void stream_records(Rng& rng, uint64_t first_serial, long fragments, const ByteSink& sink, long& records_written) const;
	Config config_;
	std::vector<std::string> contig_names_;
	std::vector<std::string> contig_sequences_;
	std::vector<Gene> genes_;
	long records_written_ = 0;
	Impl* impl_;
};

}

#endif
