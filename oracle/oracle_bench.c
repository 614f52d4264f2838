/*
 * oracle/oracle_bench.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU baseline driver for bench.py's "cpu_baseline" leg: N worker threads, each
 * owning one workspace (fqo_ctx) and pulling whole blocks, exactly the shape of
 * the reference's processReads()/processArchiveParts() thread pool
 * (src/process.cpp:46-68, 93-104).  Only encodeChunk/decodeChunk-equivalent work
 * is inside the timed region (workspaces are built before the start barrier).
 */
#define _POSIX_C_SOURCE 200809L
#include "fqc_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
  const fqo_seq_ft *sft;
  const fqo_qual_ft *qft;
  int n_blocks;
  uint8_t **raws;
  const fqo_rec **recs;
  const size_t *n_recs;
  const size_t *n_bases;
  int decode; /* 0: encode timed; 1: encode + decode timed; 2: decode alone timed (streams prepared before) */
  uint8_t **enc_seq, **enc_qual; /* decode == 2: every block's streams and N tables */
  uint16_t **enc_nc, **enc_np;
  size_t *enc_sl, *enc_ql, *enc_npn;
  int next2;
  pthread_barrier_t start2;
  int next; /* block dispenser, guarded by mu (reader mutex in the reference) */
  pthread_mutex_t mu;
  pthread_barrier_t start;
  int rc;
} job_t;

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *worker(void *arg) {
  job_t *j = (job_t *)arg;
  fqo_ctx *c = fqo_ctx_create(j->sft, j->qft);
  size_t max_bases = 0, max_recs = 0;
  int b;
  for (b = 0; b < j->n_blocks; b++) {
    if (j->n_bases[b] > max_bases) max_bases = j->n_bases[b];
    if (j->n_recs[b] > max_recs) max_recs = j->n_recs[b];
  }
  {
    const size_t scap = fqo_bound_seq(max_bases), qcap = fqo_bound_qual(max_bases);
    uint8_t *seq = (uint8_t *)malloc(scap), *qual = (uint8_t *)malloc(qcap);
    uint16_t *rl = (uint16_t *)malloc(2 * (max_recs + 1));
    uint16_t *nc = (uint16_t *)malloc(2 * (max_recs + 1));
    uint16_t *np = (uint16_t *)malloc(2 * (max_bases + 1));
    pthread_barrier_wait(&j->start);
    if (j->decode == 2) {
      /* untimed: code every block once and keep its streams; timed (behind start2): decodeChunk only */
      for (;;) {
        size_t sl = 0, ql = 0, npn = 0;
        int rc;
        pthread_mutex_lock(&j->mu);
        b = j->next++;
        pthread_mutex_unlock(&j->mu);
        if (b >= j->n_blocks) break;
        rc = fqo_encode_block(c, j->raws[b], j->recs[b], j->n_recs[b], seq, scap, &sl, qual, qcap, &ql, rl, nc, np, &npn);
        if (rc != 0) { j->rc = rc; continue; }
        j->enc_seq[b] = (uint8_t *)malloc(sl + 8); memcpy(j->enc_seq[b], seq, sl);
        j->enc_qual[b] = (uint8_t *)malloc(ql + 8); memcpy(j->enc_qual[b], qual, ql);
        j->enc_nc[b] = (uint16_t *)malloc(2 * (j->n_recs[b] + 1)); memcpy(j->enc_nc[b], nc, 2 * j->n_recs[b]);
        j->enc_np[b] = (uint16_t *)malloc(2 * (npn + 1)); memcpy(j->enc_np[b], np, 2 * npn);
        j->enc_sl[b] = sl; j->enc_ql[b] = ql; j->enc_npn[b] = npn;
      }
      pthread_barrier_wait(&j->start2);
      for (;;) {
        int rc;
        pthread_mutex_lock(&j->mu);
        b = j->next2++;
        pthread_mutex_unlock(&j->mu);
        if (b >= j->n_blocks) break;
        if (!j->enc_seq[b]) continue;
        rc = fqo_decode_block(c, j->enc_seq[b], j->enc_sl[b], j->enc_qual[b], j->enc_ql[b], j->enc_nc[b], j->n_recs[b],
                              j->enc_np[b], j->enc_npn[b], j->recs[b], j->n_recs[b], j->raws[b]);
        if (rc != 0) j->rc = rc;
      }
    } else
    for (;;) {
      size_t sl = 0, ql = 0, npn = 0;
      int rc;
      pthread_mutex_lock(&j->mu);
      b = j->next++;
      pthread_mutex_unlock(&j->mu);
      if (b >= j->n_blocks) break;
      rc = fqo_encode_block(c, j->raws[b], j->recs[b], j->n_recs[b], seq, scap, &sl, qual, qcap,
                            &ql, rl, nc, np, &npn);
      if (rc == 0 && j->decode)
        rc = fqo_decode_block(c, seq, sl, qual, ql, nc, j->n_recs[b], np, npn, j->recs[b],
                              j->n_recs[b], j->raws[b]);
      if (rc != 0) j->rc = rc;
    }
    free(seq); free(qual); free(rl); free(nc); free(np);
  }
  fqo_ctx_destroy(c);
  return NULL;
}

/* Encodes (and optionally decodes back in place) every block once with
 * n_threads workers; returns wall seconds of the coding region, <0 on error.
 * decode == 2 times the decode alone (SequenceDecoder/QualityDecoder::decodeRecord loops,
 * src/fse_sequence.cpp:114-143, src/fse_quality.cpp:55-67). */
double fqo_bench_blocks(const fqo_seq_ft *sft, const fqo_qual_ft *qft, int n_threads, int n_blocks,
                        uint8_t **raws, const fqo_rec **recs, const size_t *n_recs,
                        const size_t *n_bases, int decode) {
  job_t j;
  pthread_t *th;
  double t0, t1;
  int i;
  memset(&j, 0, sizeof(j));
  j.sft = sft; j.qft = qft; j.n_blocks = n_blocks; j.raws = raws; j.recs = recs;
  j.n_recs = n_recs; j.n_bases = n_bases; j.decode = decode;
  pthread_mutex_init(&j.mu, NULL);
  pthread_barrier_init(&j.start, NULL, (unsigned)n_threads + 1);
  pthread_barrier_init(&j.start2, NULL, (unsigned)n_threads + 1);
  if (decode == 2) {
    j.enc_seq = (uint8_t **)calloc((size_t)n_blocks, sizeof(void *));
    j.enc_qual = (uint8_t **)calloc((size_t)n_blocks, sizeof(void *));
    j.enc_nc = (uint16_t **)calloc((size_t)n_blocks, sizeof(void *));
    j.enc_np = (uint16_t **)calloc((size_t)n_blocks, sizeof(void *));
    j.enc_sl = (size_t *)calloc((size_t)n_blocks, sizeof(size_t));
    j.enc_ql = (size_t *)calloc((size_t)n_blocks, sizeof(size_t));
    j.enc_npn = (size_t *)calloc((size_t)n_blocks, sizeof(size_t));
  }
  th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
  for (i = 0; i < n_threads; i++) pthread_create(&th[i], NULL, worker, &j);
  pthread_barrier_wait(&j.start);
  if (decode == 2) pthread_barrier_wait(&j.start2);
  t0 = now_s();
  for (i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
  t1 = now_s();
  free(th);
  if (decode == 2) {
    for (i = 0; i < n_blocks; i++) { free(j.enc_seq[i]); free(j.enc_qual[i]); free(j.enc_nc[i]); free(j.enc_np[i]); }
    free(j.enc_seq); free(j.enc_qual); free(j.enc_nc); free(j.enc_np); free(j.enc_sl); free(j.enc_ql); free(j.enc_npn);
  }
  pthread_barrier_destroy(&j.start2);
  pthread_barrier_destroy(&j.start);
  pthread_mutex_destroy(&j.mu);
  return j.rc ? -1.0 : t1 - t0;
}
