/*
 * oracle/standin/bgzf.h -- TEST INFRASTRUCTURE, not product code.
 * Stand-in for the three htslib BGZF entry points the reference uses to read
 * .gz annotation/assembly files (source/read_compressed_file.cpp:22-37).
 * BGZF is a series of concatenated gzip members, so zlib's gz* reader decodes
 * it (and plain gzip) directly.
 */
#ifndef ORACLE_STANDIN_BGZF_H
#define ORACLE_STANDIN_BGZF_H 1
#include <stddef.h>
#include <sys/types.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct BGZF BGZF;
BGZF *bgzf_open(const char *path, const char *mode);
ssize_t bgzf_read(BGZF *fp, void *data, size_t length);
int bgzf_close(BGZF *fp);
#ifdef __cplusplus
}
#endif
#endif
