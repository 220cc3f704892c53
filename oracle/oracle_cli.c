/*
 * oracle_cli.c -- command-line driver of the CPU ORACLE (test infrastructure).
 *
 * Restates main.cpp:83-239 + argument_parser.hpp + FastaSplitter/FastqSplitter
 * record handling + ReadOutput.hpp:37-50 so the oracle can be pinned end to
 * end on the reference's example (README.md:63-69): same flags, ssv on
 * stdout, FASTQ to -o/-p.  The FASTA/FASTQ reader restates the parse rules of
 * kseq.h:177-218 (name = up to first whitespace; multi-line sequence; quality
 * read until it is as long as the sequence).
 */
#define _GNU_SOURCE
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <zlib.h>

#include "shark_oracle.h"

/* ---- reader (kseq.h semantics) ----------------------------------------- */
typedef struct { char *s; size_t l, m; } str_t;
static void str_push(str_t *s, int c)
{
  if (s->l + 2 > s->m) { s->m = s->m ? s->m * 2 : 256; s->s = (char *)realloc(s->s, s->m); }
  s->s[s->l++] = (char)c; s->s[s->l] = 0;
}
static void str_clear(str_t *s) { s->l = 0; if (!s->s) { s->m = 256; s->s = (char *)malloc(256); } s->s[0] = 0; }

typedef struct {
  gzFile f;
  unsigned char buf[16384]; /* kseq.h:228 */
  int begin, end, eof;
  int last_char;
  str_t name, comment, seq, qual;
} reader;

static int rd_getc(reader *r) /* kseq.h:67-79 */
{
  if (r->eof && r->begin >= r->end) return -1;
  if (r->begin >= r->end) {
    r->begin = 0;
    r->end = gzread(r->f, r->buf, sizeof(r->buf));
    if (r->end <= 0) { r->eof = 1; r->end = 0; return -1; }
  }
  return (int)r->buf[r->begin++];
}

/* append the rest of the current line to s (kseq.h:94-140 with KS_SEP_LINE, append=1);
 * returns -1 when nothing could be read because of EOF */
static int rd_line(reader *r, str_t *s)
{
  int got = 0, c;
  if (r->eof && r->begin >= r->end) return -1;
  while ((c = rd_getc(r)) >= 0) {
    got = 1;
    if (c == '\n') break;
    str_push(s, c);
  }
  if (!got && c < 0) return -1;
  if (s->l > 1 && s->s[s->l - 1] == '\r') { s->l--; s->s[s->l] = 0; } /* kseq.h:135 */
  return (int)s->l;
}

static int rd_read(reader *r) /* kseq.h:177-218 */
{
  int c;
  if (r->last_char == 0) {
    while ((c = rd_getc(r)) >= 0 && c != '>' && c != '@') {}
    if (c < 0) return -1;
    r->last_char = c;
  }
  str_clear(&r->comment); str_clear(&r->seq); str_clear(&r->qual); str_clear(&r->name);
  /* name: up to the first whitespace (kseq.h:188) */
  int got = 0;
  while ((c = rd_getc(r)) >= 0) {
    got = 1;
    if (isspace(c)) break;
    str_push(&r->name, c);
  }
  if (!got) return -1;
  if (c >= 0 && c != '\n') rd_line(r, &r->comment);
  while ((c = rd_getc(r)) >= 0 && c != '>' && c != '+' && c != '@') { /* kseq.h:194-198 */
    if (c == '\n') continue;
    str_push(&r->seq, c);
    rd_line(r, &r->seq);
  }
  if (c == '>' || c == '@') r->last_char = c;
  if (c != '+') return (int)r->seq.l;
  while ((c = rd_getc(r)) >= 0 && c != '\n') {}
  if (c < 0) return -2;
  while (rd_line(r, &r->qual) >= 0 && r->qual.l < r->seq.l) {}
  r->last_char = 0;
  if (r->seq.l != r->qual.l) return -2;
  return (int)r->seq.l;
}

static reader *rd_open(const char *path)
{
  reader *r = (reader *)calloc(1, sizeof(reader));
  r->f = gzopen(path, "r");
  if (!r->f) { free(r); return NULL; }
  return r;
}
static void rd_close(reader *r)
{
  if (!r) return;
  gzclose(r->f);
  free(r->name.s); free(r->comment.s); free(r->seq.s); free(r->qual.s);
  free(r);
}

/* ---- main --------------------------------------------------------------- */
typedef struct { char *id, *seq, *qual; } rec_t;

int main(int argc, char **argv)
{
  /* argument_parser.hpp:49-63 defaults */
  const char *fasta = NULL, *s1p = NULL, *s2p = NULL, *o1p = NULL, *o2p = NULL;
  unsigned k = 17; double cth = 0.6; uint64_t bf_size = (uint64_t)1 << 33;
  int min_quality = 0, single = 0, threads = 1;
  static const struct option lo[] = {
    {"reference", required_argument, 0, 'r'}, {"threads", required_argument, 0, 't'},
    {"sample1", required_argument, 0, '1'}, {"sample2", required_argument, 0, '2'},
    {"out1", required_argument, 0, 'o'}, {"out2", required_argument, 0, 'p'},
    {"kmer-size", required_argument, 0, 'k'}, {"confidence", required_argument, 0, 'c'},
    {"bf-size", required_argument, 0, 'b'}, {"min-base-quality", required_argument, 0, 'q'},
    {"single", no_argument, 0, 's'}, {"verbose", no_argument, 0, 'v'}, {"help", no_argument, 0, 'h'},
    {"bf-bits", required_argument, 0, 'B'}, /* oracle-only: exact filter size in bits (tests) */
    {0, 0, 0, 0}};
  int ch;
  while ((ch = getopt_long(argc, argv, "t:r:1:2:o:p:k:c:b:q:svhB:", lo, NULL)) != -1) {
    switch (ch) {
    case 'r': fasta = optarg; break;
    case 't': threads = atoi(optarg); if (threads <= 0) return EXIT_FAILURE; break;
    case '1': s1p = optarg; break;
    case '2': s2p = optarg; break;
    case 'o': o1p = optarg; break;
    case 'p': o2p = optarg; break;
    case 'k': k = (unsigned)atoi(optarg); if (k == 0 || k > 31) return EXIT_FAILURE; break;
    case 'c': cth = atof(optarg); if (cth < 0 || cth > 1) return EXIT_FAILURE; break;
    case 'b': bf_size = strtoull(optarg, NULL, 10) * ((uint64_t)1 << 33); break; /* :130-134 */
    case 'B': bf_size = strtoull(optarg, NULL, 10); break;
    case 'q': min_quality = atoi(optarg); if (min_quality < 0) return EXIT_FAILURE; break;
    case 's': single = 1; break;
    case 'v': break;
    case 'h': return EXIT_SUCCESS;
    default: return EXIT_FAILURE;
    }
  }
  if (!fasta || !s1p) { fprintf(stderr, "shark : missing required files\n"); return EXIT_FAILURE; }
  if (!o1p) o1p = "sharked_sample.1";                       /* argument_parser.hpp:168-170 */
  if (!o2p && s2p) o2p = "sharked_sample.2";                /* :171-173 */

  /* FastaSplitter.hpp:42-54: legend_ID in file order; records kept for both passes */
  reader *fr = rd_open(fasta);
  if (!fr) { fprintf(stderr, "cannot open %s\n", fasta); return EXIT_FAILURE; }
  size_t nrec = 0, caprec = 0;
  char **names = NULL, **seqs = NULL; uint64_t *lens = NULL;
  int l;
  while ((l = rd_read(fr)) >= 0) {
    if (nrec == caprec) {
      caprec = caprec ? caprec * 2 : 128;
      names = (char **)realloc(names, caprec * sizeof(char *));
      seqs = (char **)realloc(seqs, caprec * sizeof(char *));
      lens = (uint64_t *)realloc(lens, caprec * sizeof(uint64_t));
    }
    names[nrec] = strdup(fr->name.s);
    /* main.cpp:164 passes seq->seq.s (a C string) to build_kmer(const string&): stops at NUL */
    seqs[nrec] = strdup(fr->seq.s);
    lens[nrec] = strlen(seqs[nrec]);
    nrec++;
  }
  rd_close(fr);

  so_shark *sh = so_shark_new(k, cth, bf_size, min_quality, single);
  if (!sh) { fprintf(stderr, "cannot allocate filter\n"); return EXIT_FAILURE; }
  so_shark_build(sh, (const char *const *)seqs, lens, nrec);

  reader *r1 = rd_open(s1p), *r2 = s2p ? rd_open(s2p) : NULL;
  if (!r1 || (s2p && !r2)) { fprintf(stderr, "cannot open sample\n"); return EXIT_FAILURE; }
  FILE *out1 = fopen(o1p, "w"), *out2 = (s2p && o2p) ? fopen(o2p, "w") : NULL;

  /* main.cpp:66-77 with -t 1 semantics (deterministic order); the threaded
   * batch path is exercised through so_classify_batch instead. */
  (void)threads;
  int gcap = 64; int *genes = (int *)malloc(gcap * sizeof(int));
  char *joined = NULL; size_t jcap = 0;
  char *previd = strdup("");                                /* ReadOutput.hpp:39 */
  uint64_t in_batch = 0;
  for (;;) {
    if (in_batch == 50000) {                                /* main.cpp:215: new batch => new ro() call */
      in_batch = 0; free(previd); previd = strdup("");
    }
    ++in_batch;
    int l1 = rd_read(r1);
    if (l1 < 0) break;                                      /* FastqSplitter.hpp:53/:60 */
    int l2 = 0;
    if (r2) { l2 = rd_read(r2); if (l2 < 0) break; }
    size_t need = (size_t)l1 + (size_t)l2 + 2;
    if (need > jcap) { jcap = need * 2; joined = (char *)realloc(joined, jcap); }
    /* the reference builds std::string from C strings: lengths are strlen */
    size_t sl1 = strlen(r1->seq.s), sl2 = r2 ? strlen(r2->seq.s) : 0;
    size_t n = so_join_mask(r1->seq.s, sl1, r1->qual.s, r2 ? r2->seq.s : NULL, sl2,
                            r2 ? r2->qual.s : NULL, r2 != NULL, (char)min_quality, joined);
    int ng = so_analyze_read(sh, joined, n, genes, gcap, NULL, NULL, NULL);
    if (ng > gcap) {
      gcap = ng; genes = (int *)realloc(genes, gcap * sizeof(int));
      ng = so_analyze_read(sh, joined, n, genes, gcap, NULL, NULL, NULL);
    }
    /* ReadOutput.hpp:40-48; previd suppresses the FASTQ record for the 2nd.. gene of a read */
    for (int g = 0; g < ng; ++g) {
      printf("%s %s\n", r1->name.s, names[genes[g]]);
      if (strcmp(previd, r1->name.s) != 0) {                /* :44-47 */
        if (out1) fprintf(out1, "@%s\n%s\n+\n%s\n", r1->name.s, r1->seq.s, r1->qual.s);
        if (out2) fprintf(out2, "@%s\n%s\n+\n%s\n", r2->name.s, r2->seq.s, r2->qual.s);
      }
      free(previd); previd = strdup(r1->name.s);            /* :48 */
    }
  }
  free(previd);
  if (out1) fclose(out1);
  if (out2) fclose(out2);
  rd_close(r1); rd_close(r2);
  so_shark_free(sh);
  for (size_t i = 0; i < nrec; ++i) { free(names[i]); free(seqs[i]); }
  free(names); free(seqs); free(lens); free(genes); free(joined);
  return 0;
}
