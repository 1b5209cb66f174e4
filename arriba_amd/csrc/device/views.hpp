// arriba_amd/csrc/device/views.hpp -- device-resident views (plain pointers into HBM) shared by all kernels.
#ifndef AGPU_VIEWS_HPP
#define AGPU_VIEWS_HPP 1

#include <stdint.h>

#if defined(__HIPCC__)
#define AGPU_HD __host__ __device__ __forceinline__
#define AGPU_UNROLL _Pragma("unroll")
#define AGPU_NOUNROLL _Pragma("nounroll")
// put at the top of a function body whose floating-point expressions restate the reference's: no fused multiply-add (the host compiler of the
// reference does not contract on baseline x86-64), so that a*b+c rounds twice as it does there
#define AGPU_FP_AS_WRITTEN _Pragma("clang fp contract(off)")
#else
#define AGPU_HD inline
#define AGPU_UNROLL
#define AGPU_NOUNROLL
#define AGPU_FP_AS_WRITTEN
#endif

namespace agpu {

enum : uint8_t { ABIT_STRAND = 1, ABIT_FIRST_IN_PAIR = 2, ABIT_SUPPLEMENTARY = 4, ABIT_EXONIC = 8, ABIT_PREDICTED_STRAND = 16, ABIT_PREDICTED_STRAND_AMBIGUOUS = 32 };
enum : uint8_t { FBIT_SINGLE_END = 1, FBIT_MULTIMAPPER = 2, FBIT_DUPLICATE = 4 };
enum : uint8_t { GBIT_STRAND = 1, GBIT_DUMMY = 2, GBIT_PROTEIN_CODING = 4 };
enum : uint8_t { CBIT_INTERESTING = 1, CBIT_VIRAL = 2 };
enum { MATE1 = 0, MATE2 = 1, SPLIT_READ = 1, SUPPLEMENTARY = 2 };
enum { CIGAR_M = 0, CIGAR_I = 1, CIGAR_D = 2, CIGAR_N = 3, CIGAR_S = 4, CIGAR_H = 5, CIGAR_P = 6, CIGAR_EQ = 7, CIGAR_X = 8 };

// filter ids: position in the reference's registry (source/common.hpp:29-67)
enum : uint8_t {
	FILTER_none = 0, FILTER_duplicates = 1, FILTER_inconsistently_clipped = 2, FILTER_homopolymer = 3, FILTER_read_through = 4, FILTER_same_gene = 5,
	FILTER_small_insert_size = 6, FILTER_long_gap = 7, FILTER_hairpin = 8, FILTER_multimappers = 9, FILTER_mismatches = 10, FILTER_mismappers = 11,
	FILTER_relative_support = 12, FILTER_uninteresting_contigs = 30, FILTER_viral_contigs = 31, FILTER_top_expressed_viral_contigs = 32,
	FILTER_low_coverage_viral_contigs = 33, FILTER_low_entropy = 36
};

const int MAX_SPLICE_SITE_DISTANCE = 2; // source/annotation.hpp:14
const int GENE_INLINE = 2;              // gene ids stored inline per alignment; larger sets live in the overflow pool

struct FlatIndexView {
	uint32_t n_contigs;
	const uint32_t* contig_offset;
	const int32_t* keys;
	const uint32_t* member_offset;
	const uint32_t* members;
	// optional coarse directory over the keys: bins[bins[contig] + j] = lower bound of position j << INDEX_BIN_SHIFT (the first n_contigs + 1
	// words are the per-contig offsets into the array itself).  It replaces the ~12 upper levels of a binary search (dependent loads that
	// miss the L2 in their lower half) by one load of two neighbouring words.  One pointer only: the kernels are short of scalar registers.
	const uint32_t* bins = nullptr;
};
const uint32_t INDEX_BIN_SHIFT = 10; // 1 kb bins: about one exon boundary per bin in a GENCODE-scale annotation, 12 MB of directory for a 3.1 Gb genome

struct AnnotationView {
	uint32_t n_genes;          // GTF genes
	uint32_t n_dummy;          // dummy genes appended after annotate
	const uint16_t* gene_contig; const int32_t* gene_start; const int32_t* gene_end; const uint8_t* gene_bits; const int32_t* gene_exonic_length;
	uint32_t n_exons;
	const int32_t* exon_start; const int32_t* exon_end; const uint32_t* exon_gene;
	const int32_t* exon_previous; const int32_t* exon_next; const int32_t* exon_cds_start; const int32_t* exon_cds_end;
	FlatIndexView exon_index;
	FlatIndexView gene_index;
	// dummy genes sorted by (contig, position): composite keys contig << 32 | coordinate
	const uint64_t* dummy_start_key; const uint64_t* dummy_end_key;
};

struct GenomeView {
	uint32_t n_contigs;
	const uint64_t* contig_offset;
	const uint8_t* contig_bits;
	const char* bases;
};

struct BatchView {
	uint64_t n;
	uint64_t first_rank;       // global name rank of fragment 0 (non-zero when the context holds one shard of the sample)
	const uint8_t* n_aln;
	uint8_t* fbits;
	uint8_t* filter;
	const uint32_t* group;
	const uint16_t* contig[3];
	const int32_t* start[3];
	const int32_t* end[3];
	uint8_t* abits[3];
	const uint32_t* cigar_offset[3];
	const uint16_t* cigar_count[3];
	const uint32_t* cigar_pool;
	const uint32_t* seq_offset[2];
	const uint32_t* seq_length[2];
	const uint8_t* seq_pool;
	// annotation results
	uint8_t* gene_count[3];
	uint32_t* genes[3];        // [n * GENE_INLINE]; if count > GENE_INLINE, genes[..][0] is the offset into gene_pool
	uint32_t* gene_pool;
	uint32_t* gene_pool_used;  // single counter
	uint32_t gene_pool_capacity;
	// null, or one byte per fragment with what the walks over read lists ask of a read (WALK_*: agpu_events.hip makes it in front of filter_both_intronic and recover_both_spliced):
	// 10^8 bytes that stay in the last-level cache, where the walks used to drag a line of each of two or four byte columns per list entry
	const uint8_t* walk = nullptr;
};
enum : uint8_t { WALK_UNFILTERED = 1, WALK_MULTIMAPPER = 2, WALK_EXONIC = 4 };

}

#endif
