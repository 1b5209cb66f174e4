/*
 * oracle/standin/cram.h -- TEST INFRASTRUCTURE, not product code.
 * CRAM input is not needed by any benchmark configuration; the reference only
 * calls cram_set_option when htsFile::is_cram is set
 * (source/read_chimeric_alignments.cpp:567-568), which the stand-in never sets.
 */
#ifndef ORACLE_STANDIN_CRAM_H
#define ORACLE_STANDIN_CRAM_H 1
#ifdef __cplusplus
extern "C" {
#endif
struct cram_fd;
enum hts_fmt_option { CRAM_OPT_REFERENCE = 5 };
int cram_set_option(struct cram_fd *fd, enum hts_fmt_option opt, ...);
#ifdef __cplusplus
}
#endif
#endif
