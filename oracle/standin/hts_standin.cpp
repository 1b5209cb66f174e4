/*
 * oracle/standin/hts_standin.cpp -- TEST INFRASTRUCTURE, not product code.
 *
 * BAM-only implementation of the htslib entry points declared in
 * oracle/standin/{sam,bgzf,cram}.h, written from the SAM/BAM specification
 * (SAMv1 section 4: BGZF = concatenated gzip members; BAM = magic, header text,
 * reference dictionary, then length-prefixed alignment records). zlib's gz*
 * reader walks concatenated gzip members, so it decodes BGZF (stored or
 * deflated blocks) as well as plain gzip; that is all the reference needs for
 * -x <bam>, and for .gz GTF/FASTA files. SAM text and CRAM are not supported.
 *
 * Semantics the reference depends on:
 *   bam_endpos  = pos + reference length of the CIGAR (at least 1)
 *   bam_aux2i   = integer aux value for types c C s S i I
 *   in-memory qname is NUL-padded to a multiple of 4 so the CIGAR is aligned
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <zlib.h>
#include "sam.h"
#include "bgzf.h"
#include "cram.h"

extern "C" {

const char seq_nt16_str[] = "=ACMGRSVTWYHKDBN";

struct standin_state_t {
	gzFile gz;
};

static inline uint32_t le32(const uint8_t* p) { return (uint32_t) p[0] | (uint32_t) p[1] << 8 | (uint32_t) p[2] << 16 | (uint32_t) p[3] << 24; }
static inline uint16_t le16(const uint8_t* p) { return (uint16_t) (p[0] | p[1] << 8); }

static bool read_fully(gzFile gz, void* buffer, size_t length, bool& clean_eof) {
	clean_eof = false;
	size_t done = 0;
	while (done < length) {
		unsigned int chunk = (length - done > (1u << 30)) ? (1u << 30) : (unsigned int) (length - done);
		int got = gzread(gz, (char*) buffer + done, chunk);
		if (got < 0)
			return false;
		if (got == 0) {
			clean_eof = (done == 0);
			return false;
		}
		done += got;
	}
	return true;
}

samFile* sam_open(const char* path, const char* mode) {
	if (mode == NULL || mode[0] != 'r')
		return NULL;
	gzFile gz = gzopen(path, "rb");
	if (gz == NULL)
		return NULL;
	gzbuffer(gz, 4u << 20);
	samFile* fp = (samFile*) calloc(1, sizeof(samFile));
	standin_state_t* state = (standin_state_t*) calloc(1, sizeof(standin_state_t));
	state->gz = gz;
	fp->is_bin = 1;
	fp->is_bgzf = 1;
	fp->is_cram = 0;
	fp->standin_state = state;
	return fp;
}

int sam_close(samFile* fp) {
	if (fp == NULL)
		return -1;
	standin_state_t* state = (standin_state_t*) fp->standin_state;
	int status = gzclose(state->gz);
	free(state);
	free(fp);
	return (status == Z_OK) ? 0 : -1;
}

int hts_set_threads(htsFile*, int) { return 0; }

int cram_set_option(struct cram_fd*, enum hts_fmt_option, ...) { return 0; }

sam_hdr_t* sam_hdr_read(samFile* fp) {
	standin_state_t* state = (standin_state_t*) fp->standin_state;
	bool eof;
	uint8_t word[4];
	if (!read_fully(state->gz, word, 4, eof) || memcmp(word, "BAM\1", 4) != 0)
		return NULL;
	if (!read_fully(state->gz, word, 4, eof))
		return NULL;
	sam_hdr_t* h = (sam_hdr_t*) calloc(1, sizeof(sam_hdr_t));
	h->l_text = le32(word);
	h->text = (char*) malloc(h->l_text + 1);
	if (h->l_text > 0 && !read_fully(state->gz, h->text, h->l_text, eof))
		return NULL;
	h->text[h->l_text] = '\0';
	if (!read_fully(state->gz, word, 4, eof))
		return NULL;
	h->n_targets = (int32_t) le32(word);
	h->target_name = (char**) calloc(h->n_targets > 0 ? h->n_targets : 1, sizeof(char*));
	h->target_len = (uint32_t*) calloc(h->n_targets > 0 ? h->n_targets : 1, sizeof(uint32_t));
	for (int32_t i = 0; i < h->n_targets; ++i) {
		if (!read_fully(state->gz, word, 4, eof))
			return NULL;
		uint32_t l_name = le32(word);
		h->target_name[i] = (char*) malloc(l_name + 1);
		if (!read_fully(state->gz, h->target_name[i], l_name, eof))
			return NULL;
		h->target_name[i][l_name] = '\0';
		if (!read_fully(state->gz, word, 4, eof))
			return NULL;
		h->target_len[i] = le32(word);
	}
	return h;
}

void bam_hdr_destroy(sam_hdr_t* h) {
	if (h == NULL)
		return;
	for (int32_t i = 0; i < h->n_targets; ++i)
		free(h->target_name[i]);
	free(h->target_name);
	free(h->target_len);
	free(h->text);
	free(h);
}

bam1_t* bam_init1(void) {
	return (bam1_t*) calloc(1, sizeof(bam1_t));
}

void bam_destroy1(bam1_t* b) {
	if (b == NULL)
		return;
	free(b->data);
	free(b);
}

int sam_read1(samFile* fp, sam_hdr_t*, bam1_t* b) {
	standin_state_t* state = (standin_state_t*) fp->standin_state;
	bool eof;
	uint8_t fixed[36];
	if (!read_fully(state->gz, fixed, 4, eof))
		return eof ? -1 : -2;
	uint32_t block_size = le32(fixed);
	if (block_size < 32)
		return -3;
	if (!read_fully(state->gz, fixed + 4, 32, eof))
		return -2;
	bam1_core_t& c = b->core;
	c.tid = (int32_t) le32(fixed + 4);
	c.pos = (int32_t) le32(fixed + 8);
	uint32_t l_read_name = fixed[12];
	c.qual = fixed[13];
	c.bin = le16(fixed + 14);
	c.n_cigar = le16(fixed + 16);
	c.flag = le16(fixed + 18);
	c.l_qseq = (int32_t) le32(fixed + 20);
	c.mtid = (int32_t) le32(fixed + 24);
	c.mpos = (int32_t) le32(fixed + 28);
	c.isize = (int32_t) le32(fixed + 32);
	uint32_t payload = block_size - 32;
	uint32_t extranul = (l_read_name % 4 != 0) ? 4 - l_read_name % 4 : 0;
	c.l_extranul = extranul;
	c.l_qname = l_read_name + extranul;
	uint32_t needed = payload + extranul;
	if (needed > b->m_data) {
		uint32_t capacity = needed + (needed >> 1) + 32;
		b->data = (uint8_t*) realloc(b->data, capacity);
		b->m_data = capacity;
	}
	if (payload < l_read_name)
		return -3;
	if (!read_fully(state->gz, b->data, l_read_name, eof))
		return -2;
	memset(b->data + l_read_name, 0, extranul);
	if (!read_fully(state->gz, b->data + c.l_qname, payload - l_read_name, eof) && payload != l_read_name)
		return -2;
	b->l_data = needed;
	return (int) block_size;
}

hts_pos_t bam_cigar2qlen(int n_cigar, const uint32_t* cigar) {
	hts_pos_t length = 0;
	for (int i = 0; i < n_cigar; ++i)
		if (bam_cigar_type(bam_cigar_op(cigar[i])) & 1)
			length += bam_cigar_oplen(cigar[i]);
	return length;
}

hts_pos_t bam_cigar2rlen(int n_cigar, const uint32_t* cigar) {
	hts_pos_t length = 0;
	for (int i = 0; i < n_cigar; ++i)
		if (bam_cigar_type(bam_cigar_op(cigar[i])) & 2)
			length += bam_cigar_oplen(cigar[i]);
	return length;
}

hts_pos_t bam_endpos(const bam1_t* b) {
	hts_pos_t reference_length = 1;
	if (!(b->core.flag & BAM_FUNMAP) && b->core.n_cigar > 0) {
		reference_length = bam_cigar2rlen(b->core.n_cigar, bam_get_cigar(b));
		if (reference_length == 0)
			reference_length = 1;
	}
	return b->core.pos + reference_length;
}

/* size in bytes of one aux value of the given type starting at s (s points at the type byte); 0 = malformed */
static size_t aux_value_size(const uint8_t* s, const uint8_t* end) {
	switch (*s) {
		case 'A': case 'c': case 'C': return 2;
		case 's': case 'S': return 3;
		case 'i': case 'I': case 'f': return 5;
		case 'd': return 9;
		case 'Z': case 'H': {
			const uint8_t* p = s + 1;
			while (p < end && *p != 0)
				++p;
			return (p < end) ? (size_t) (p - s) + 1 : 0;
		}
		case 'B': {
			if (s + 6 > end)
				return 0;
			size_t element;
			switch (s[1]) {
				case 'c': case 'C': element = 1; break;
				case 's': case 'S': element = 2; break;
				case 'i': case 'I': case 'f': element = 4; break;
				default: return 0;
			}
			return 6 + element * le32(s + 2);
		}
		default: return 0;
	}
}

uint8_t* bam_aux_get(const bam1_t* b, const char tag[2]) {
	uint8_t* s = bam_get_aux(b);
	uint8_t* end = b->data + b->l_data;
	while (s + 3 <= end) {
		size_t size = aux_value_size(s + 2, end);
		if (size == 0 || s + 2 + size > end)
			return NULL;
		if (s[0] == (uint8_t) tag[0] && s[1] == (uint8_t) tag[1])
			return s + 2;
		s += 2 + size;
	}
	return NULL;
}

int64_t bam_aux2i(const uint8_t* s) {
	switch (*s) {
		case 'c': return (int8_t) s[1];
		case 'C': return s[1];
		case 's': return (int16_t) le16(s + 1);
		case 'S': return le16(s + 1);
		case 'i': return (int32_t) le32(s + 1);
		case 'I': return le32(s + 1);
		default: return 0;
	}
}

struct BGZF {
	gzFile gz;
};

BGZF* bgzf_open(const char* path, const char* mode) {
	if (mode == NULL || mode[0] != 'r')
		return NULL;
	gzFile gz = gzopen(path, "rb");
	if (gz == NULL)
		return NULL;
	gzbuffer(gz, 1u << 20);
	BGZF* fp = (BGZF*) calloc(1, sizeof(BGZF));
	fp->gz = gz;
	return fp;
}

ssize_t bgzf_read(BGZF* fp, void* data, size_t length) {
	size_t done = 0;
	while (done < length) {
		int got = gzread(fp->gz, (char*) data + done, (unsigned int) (length - done));
		if (got < 0)
			return -1;
		if (got == 0)
			break;
		done += got;
	}
	return (ssize_t) done;
}

int bgzf_close(BGZF* fp) {
	int status = gzclose(fp->gz);
	free(fp);
	return (status == Z_OK) ? 0 : -1;
}

} /* extern "C" */
