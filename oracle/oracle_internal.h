/* oracle/oracle_internal.h -- TEST INFRASTRUCTURE ONLY: pieces shared between oracle.c and oracle_cse.c. */
#ifndef REGTOOLS_ORACLE_INTERNAL_H
#define REGTOOLS_ORACLE_INTERNAL_H
#include <stddef.h>
#include <stdint.h>

typedef struct { char *name; int64_t len, offset; int line_blen, line_len; } fa_seq;
typedef struct { uint8_t *data; size_t dlen; fa_seq *seq; int n; } fasta;

fasta *fasta_load(const char *path);
void   fasta_free(fasta *fa);
/* fai_fetch("name:beg1-end1") semantics (faidx.c:341-413); -1 when the contig is missing */
int    fasta_fetch(const fasta *fa, const char *name, int64_t beg1, int64_t end1, char *out, int cap);
void   orc_rev_comp(char *s, int n);
uint8_t *orc_slurp(const char *path, size_t *len);
#endif
