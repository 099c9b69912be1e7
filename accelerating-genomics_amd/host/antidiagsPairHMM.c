/*
 * Drop-in for the reference command line `antidiagsPairHMM <input_file_r> <output_file>`
 * (pairHMM/antidiagsPairHMM.c:307-500): same arguments, same input format, same output file
 * (one "%f\n" per (read, haplotype), read-major inside a batch), same stdout chatter
 * (`#batch: k` at the top of every loop turn, including the one that meets EOF, and every
 * value), same failure behaviour on a truncated batch.  pairHMM() (:120-267) runs on the GPU
 * through libagx (include/agx.h); there is no CPU path.
 * Built a second time as `pairHMMmatrix` (-DAGX_PHMM_MATRIX_STDOUT): the reference's row-major
 * program pairHMM/pairHMMmatrix.c has the same command line and output file but prints only the
 * `#batch:` lines on stdout (:171 vs its fprintf at :258).
 *   AGX_PHMM_PRECISION = f64 (default; raw sums bit-identical to the reference) | f64fma | f32 | f32fma
 *   AGX_NUM_DEVICES = n   shard whole batches over GPUs 0..n-1 (default 1, 0 = all visible)
 *   AGX_DEVICES = 0,0,1   explicit list, one shard per entry (an ordinal may repeat)
 *   AGX_CLI_CHUNK_PAIRS   pairs per pipeline step, whole batches (default 65536)
 *
 * Streaming, the reference's batch loop (:371-433, :484-489) as a pipeline of three
 * threads: the parser reads batches k+1.. while the main thread has batch k on the device and the
 * printer formats and writes batch k-1.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "agx.h"
#include "agx_fmt.h"
#include "agx_pipe.h"

static double seconds(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct {
    agx_phmm_reader *reader;
    int64_t chunk_pairs;
    agx_pipe q;
    int rc;
    char err[512];
    double t_parse; /* AGX_TRACE_CLI: time inside the reader */
} parse_stage;

static void *parser_main(void *arg)
{
    parse_stage *s = (parse_stage *)arg;
    while (!agx_phmm_reader_done(s->reader)) {
        agx_phmm_text *t = NULL;
        const double ta = seconds();
        const int rc = agx_phmm_reader_next(s->reader, s->chunk_pairs, &t);
        s->t_parse += seconds() - ta;
        if (rc != AGX_OK) {
            s->rc = rc;
            snprintf(s->err, sizeof s->err, "%s", agx_last_error());
            break;
        }
        agx_pipe_push(&s->q, t);
    }
    agx_pipe_close(&s->q);
    return NULL;
}

typedef struct {
    agx_phmm_text *text;
    double *lh;
} scored_chunk;

typedef struct {
    const int *devices; /* NULL: devices 0 .. n-1 */
    int n;
    double t_warm;
} warm_t;

static void *warm_main(void *arg)
{
    warm_t *w = (warm_t *)arg;
    const double ta = seconds();
    (void)agx_warmup_devices(w->devices, w->n); /* HIP start-up beside the parsing; a failure shows at the first batch */
    w->t_warm = seconds() - ta;
    return NULL;
}

typedef struct {
    agx_pipe q;
    FILE *out;
    uint32_t batches; /* `#batch:` lines printed so far */
    double t_print;   /* AGX_TRACE_CLI: time spent formatting and writing */
} print_stage;

static void *printer_main(void *arg)
{
    print_stage *s = (print_stage *)arg;
    for (;;) {
        scored_chunk *c = (scored_chunk *)agx_pipe_pop(&s->q);
        if (!c) break;
        const double ta = seconds();
        const agx_phmm_desc *d = &c->text->desc;
        int64_t k = 0;
        /* every value is formatted once (agx_fmt.h: printf's "%f\n" byte for byte, at a tenth of its cost) and written
         * twice, a block of lines at a time */
        static char block[(size_t)1 << 16];
        for (uint32_t g = 0; g < d->n_regions; g++) {
            printf("#batch: %u\n", ++s->batches); /* :372 */
            const int64_t n = (int64_t)(d->region_read[g + 1] - d->region_read[g]) * (d->region_hap[g + 1] - d->region_hap[g]);
            size_t fill = 0;
            for (int64_t i = 0; i < n; i++, k++) {
                fill += (size_t)agx_fmt_f6_line(block + fill, c->lh[k]);
                if (fill + AGX_FMT_F6_MAX > sizeof block || i + 1 == n) {
#ifndef AGX_PHMM_MATRIX_STDOUT
                    fwrite(block, 1, fill, stdout); /* :459 */
#endif
                    fwrite(block, 1, fill, s->out); /* :461 */
                    fill = 0;
                }
            }
        }
        free(c->lh);
        agx_phmm_text_free(c->text);
        free(c);
        s->t_print += seconds() - ta;
    }
    return NULL;
}

int main(int argc, const char *argv[])
{
    if (argc != 3) {
        fprintf(stderr, "Usage: %s <input_file_r> <output_file>\n", argv[0]); /* :314-317 */
        return EXIT_FAILURE;
    }
    const int trace = getenv("AGX_TRACE_CLI") != NULL; /* stage times on stderr, like the Smith-Waterman command line */
    const double tr0 = seconds();
    parse_stage ps;
    memset(&ps, 0, sizeof ps);
    int rc = agx_phmm_reader_open(argv[1], &ps.reader);
    if (rc != AGX_OK) {
        fprintf(stderr, "%s\n", agx_last_error()); /* perror("Error opening input file_r"), :320-324 */
        return EXIT_FAILURE;
    }
    print_stage pr;
    memset(&pr, 0, sizeof pr);
    pr.out = fopen(argv[2], "w");
    if (!pr.out) {
        perror("Error opening output file"); /* :333-339 */
        return EXIT_FAILURE;
    }
    int precision = AGX_PHMM_F64;
    const char *pe = getenv("AGX_PHMM_PRECISION");
    if (pe && strcmp(pe, "f32") == 0) precision = AGX_PHMM_F32;
    else if (pe && strcmp(pe, "f32fma") == 0) precision = AGX_PHMM_F32_FMA;
    else if (pe && strcmp(pe, "f64fma") == 0) precision = AGX_PHMM_F64_FMA;
    int devices[64];
    const int n_dev = agx_parse_devices(getenv("AGX_DEVICES"), devices, 64);
    const char *nd = getenv("AGX_NUM_DEVICES");
    const int n_multi = nd ? atoi(nd) : 1;
    const char *cp = getenv("AGX_CLI_CHUNK_PAIRS");
    ps.chunk_pairs = cp && atoll(cp) > 0 ? atoll(cp) : 65536;
    agx_pipe_init(&ps.q, 2);
    agx_pipe_init(&pr.q, 2);
    pthread_t parser, printer, warmer;
    warm_t warm;
    memset(&warm, 0, sizeof warm);
    warm.devices = n_dev ? devices : NULL;
    warm.n = n_dev ? n_dev : n_multi;
    if (pthread_create(&parser, NULL, parser_main, &ps) || pthread_create(&printer, NULL, printer_main, &pr) ||
        pthread_create(&warmer, NULL, warm_main, &warm)) {
        fprintf(stderr, "antidiagsPairHMM: cannot start threads\n");
        return EXIT_FAILURE;
    }
    int status = EXIT_SUCCESS, truncated = 0, warm_joined = 0;
    double t_wait = 0, t_warm_wait = 0, t_score = 0, t_push = 0;
    long n_chunks = 0;
    for (;;) {
        double ta = seconds();
        agx_phmm_text *t = (agx_phmm_text *)agx_pipe_pop(&ps.q);
        t_wait += seconds() - ta;
        if (!t) break;
        if (!warm_joined) {
            ta = seconds();
            pthread_join(warmer, NULL);
            t_warm_wait = seconds() - ta;
            warm_joined = 1;
        }
        ta = seconds();
        if (t->truncated) truncated = t->truncated;
        scored_chunk *c = (scored_chunk *)calloc(1, sizeof *c);
        double *lh = (double *)malloc(sizeof(double) * (size_t)(t->n_pairs ? t->n_pairs : 1));
        if (!c || !lh) {
            fprintf(stderr, "Error allocating memory for matrices.\n");
            status = EXIT_FAILURE;
        }
        if (status == EXIT_SUCCESS && t->n_pairs) {
            rc = n_dev ? agx_phmm_forward_devices(devices, n_dev, &t->desc, precision, lh)
                       : agx_phmm_forward_multi(n_multi, &t->desc, precision, lh);
            if (rc != AGX_OK) {
                fprintf(stderr, "antidiagsPairHMM: %s\n", agx_last_error());
                status = EXIT_FAILURE;
            }
        }
        if (status != EXIT_SUCCESS) { /* drain the parser so it can finish, print nothing more */
            free(lh);
            free(c);
            agx_phmm_text_free(t);
            while ((t = (agx_phmm_text *)agx_pipe_pop(&ps.q)) != NULL) agx_phmm_text_free(t);
            break;
        }
        t_score += seconds() - ta;
        c->text = t;
        c->lh = lh;
        ta = seconds();
        agx_pipe_push(&pr.q, c);
        t_push += seconds() - ta;
        n_chunks++;
    }
    pthread_join(parser, NULL);
    agx_pipe_close(&pr.q);
    pthread_join(printer, NULL);
    if (status == EXIT_SUCCESS && ps.rc != AGX_OK) {
        fprintf(stderr, "antidiagsPairHMM: %s\n", ps.err);
        status = EXIT_FAILURE;
    }
    if (status == EXIT_SUCCESS) {
        printf("#batch: %u\n", pr.batches + 1); /* the turn that meets EOF, or the truncated batch */
        if (truncated) {
            fprintf(stderr, truncated == 2 ? "Memory allocation failed for haplotypes array\n" /* :381-385, a negative count */
                                           : "Error reading haplotypes.\n");                   /* :394-398 */
            status = EXIT_FAILURE;
        }
    }
    if (!warm_joined) pthread_join(warmer, NULL);
    if (trace)
        fprintf(stderr,
                "[cli] %ld chunk(s): HIP start-up %.3f s (waited %.3f s more for it behind the first chunk), waited for the parser %.3f s "
                "(parser busy %.3f s), device %.3f s, waited for the printer's queue %.3f s (printer busy %.3f s), total %.3f s\n",
                n_chunks, warm.t_warm, t_warm_wait, t_wait, ps.t_parse, t_score, t_push, pr.t_print, seconds() - tr0);
    fclose(pr.out);
    agx_phmm_reader_close(ps.reader);
    /* everything is written: leave without tearing the HIP runtime down */
    fflush(NULL);
    _exit(status);
}
