/*
 * oracle/standin/sam.h -- TEST INFRASTRUCTURE, not product code.
 *
 * Declaration-compatible stand-in for the small subset of htslib's <sam.h>
 * that the reference (suhrig/arriba v2.5.1, /root/reference/source) consumes.
 * htslib 1.22.1 is downloaded by the reference's Makefile (Makefile:37-39) and
 * is not available offline, so the oracle build (oracle/Makefile) compiles the
 * reference's UNMODIFIED sources against this header and oracle/standin/hts_standin.cpp.
 * Only container decoding lives here (BAM over gzip/BGZF through zlib); all
 * arithmetic of the hot path stays in the reference's own sources.
 *
 * Call sites served: source/read_chimeric_alignments.cpp:19-91,197-336,511-773,
 * source/read_stats.cpp:161-266, source/common.hpp:185-207.
 * Semantics follow the SAM/BAM specification (SAMv1 section 4) and the public
 * htslib API documentation; nothing is copied from htslib.
 */
#ifndef ORACLE_STANDIN_SAM_H
#define ORACLE_STANDIN_SAM_H 1

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int64_t hts_pos_t;

/* CIGAR operations (SAMv1 4.2) */
#define BAM_CMATCH      0
#define BAM_CINS        1
#define BAM_CDEL        2
#define BAM_CREF_SKIP   3
#define BAM_CSOFT_CLIP  4
#define BAM_CHARD_CLIP  5
#define BAM_CPAD        6
#define BAM_CEQUAL      7
#define BAM_CDIFF       8
#define BAM_CBACK       9

#define BAM_CIGAR_SHIFT 4
#define BAM_CIGAR_MASK  0xf
/* bit0: consumes query, bit1: consumes reference; two bits per op, MIDNSHP=XB */
#define BAM_CIGAR_TYPE  0x3C1A7

#define bam_cigar_op(c)     ((c) & BAM_CIGAR_MASK)
#define bam_cigar_oplen(c)  ((c) >> BAM_CIGAR_SHIFT)
#define bam_cigar_gen(l, o) ((l) << BAM_CIGAR_SHIFT | (o))
#define bam_cigar_type(o)   (BAM_CIGAR_TYPE >> ((o) << 1) & 3)

/* FLAG bits (SAMv1 1.4) */
#define BAM_FPAIRED         1
#define BAM_FPROPER_PAIR    2
#define BAM_FUNMAP          4
#define BAM_FMUNMAP         8
#define BAM_FREVERSE       16
#define BAM_FMREVERSE      32
#define BAM_FREAD1         64
#define BAM_FREAD2        128
#define BAM_FSECONDARY    256
#define BAM_FQCFAIL       512
#define BAM_FDUP         1024
#define BAM_FSUPPLEMENTARY 2048

typedef struct bam1_core_t {
	hts_pos_t pos;
	int32_t tid;
	uint16_t bin;
	uint8_t qual;
	uint8_t l_extranul;
	uint16_t flag;
	uint16_t l_qname;     /* including terminating NUL(s) */
	uint32_t n_cigar;
	int32_t l_qseq;
	int32_t mtid;
	hts_pos_t mpos;
	hts_pos_t isize;
} bam1_core_t;

typedef struct bam1_t {
	bam1_core_t core;
	uint64_t id;
	uint8_t *data;        /* qname | cigar | seq(4bit) | qual | aux */
	int l_data;
	uint32_t m_data;
	uint32_t mempolicy;
} bam1_t;

#define bam_get_qname(b) ((char*)(b)->data)
#define bam_get_cigar(b) ((uint32_t*)((b)->data + (b)->core.l_qname))
#define bam_get_seq(b)   ((b)->data + ((b)->core.n_cigar << 2) + (b)->core.l_qname)
#define bam_get_qual(b)  ((b)->data + ((b)->core.n_cigar << 2) + (b)->core.l_qname + (((b)->core.l_qseq + 1) >> 1))
#define bam_get_aux(b)   ((b)->data + ((b)->core.n_cigar << 2) + (b)->core.l_qname + (((b)->core.l_qseq + 1) >> 1) + (b)->core.l_qseq)
#define bam_get_l_aux(b) ((b)->l_data - ((b)->core.n_cigar << 2) - (b)->core.l_qname - (b)->core.l_qseq - (((b)->core.l_qseq + 1) >> 1))
#define bam_seqi(s, i)   ((s)[(i) >> 1] >> ((~(i) & 1) << 2) & 0xf)

extern const char seq_nt16_str[];

typedef struct sam_hdr_t {
	int32_t n_targets;
	uint32_t *target_len;
	char **target_name;
	char *text;
	size_t l_text;
} sam_hdr_t;
typedef sam_hdr_t bam_hdr_t;

struct cram_fd;
typedef struct htsFile {
	uint32_t is_bin:1, is_write:1, is_be:1, is_cram:1, is_bgzf:1, dummy:27;
	union {
		void *bgzf;
		struct cram_fd *cram;
		void *hfile;
	} fp;
	void *standin_state;
} htsFile;
typedef htsFile samFile;

samFile *sam_open(const char *path, const char *mode);
int sam_close(samFile *fp);
int hts_set_threads(htsFile *fp, int n);
sam_hdr_t *sam_hdr_read(samFile *fp);
void bam_hdr_destroy(sam_hdr_t *h);
int sam_read1(samFile *fp, sam_hdr_t *h, bam1_t *b); /* >=0 ok, -1 EOF, <-1 error */

bam1_t *bam_init1(void);
void bam_destroy1(bam1_t *b);

hts_pos_t bam_cigar2qlen(int n_cigar, const uint32_t *cigar);
hts_pos_t bam_cigar2rlen(int n_cigar, const uint32_t *cigar);
hts_pos_t bam_endpos(const bam1_t *b);

uint8_t *bam_aux_get(const bam1_t *b, const char tag[2]);
int64_t bam_aux2i(const uint8_t *s);

#ifdef __cplusplus
}
#endif

#endif /* ORACLE_STANDIN_SAM_H */
