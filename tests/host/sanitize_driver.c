/*
 * Host-only driver for the sanitizer build (tests/test_host_sanitizers.py): exercises the text
 * readers and the plan-only batch constructors of libagx under AddressSanitizer + UBSan.
 * No device is touched (ctx == NULL everywhere).  usage: sanitize_driver <golden dir>
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "agx.h"

static int fails = 0;
#define EXPECT(c)                                                   \
    do {                                                            \
        if (!(c)) {                                                 \
            fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, #c); \
            fails++;                                                \
        }                                                           \
    } while (0)

static void sw_file(const char *dir, const char *name, int line_buf)
{
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    agx_sw_text *t = NULL;
    int rc = agx_sw_text_read(path, line_buf, &t);
    EXPECT(rc == AGX_OK && t);
    if (!t) return;
    agx_sw_batch *b = NULL;
    rc = agx_sw_batch_create(NULL, t->bases, t->off, t->len, t->n_pairs, &b);
    EXPECT(rc == AGX_OK && b);
    agx_sw_info info;
    if (b) {
        EXPECT(agx_sw_batch_info(b, &info) == AGX_OK && info.n_pairs == t->n_pairs && info.padded_cells >= info.cells);
        EXPECT(agx_sw_batch_launch(b) == AGX_E_NODEVICE);
        agx_sw_batch_destroy(b);
    }
    agx_sw_scoring sc = {3, -2, -6, -1};
    rc = agx_sw_batch_create_scored(NULL, &sc, t->bases, t->off, t->len, t->n_pairs, &b);
    EXPECT(rc == AGX_OK);
    agx_sw_batch_destroy(b);
    b = NULL;
    /* substitution-matrix mode over every byte the file holds (newline included) */
    agx_sw_matrix m;
    memset(&m, 0, sizeof m);
    m.n_symbols = 6;
    m.gap_open = -4;
    m.gap_extend = -1;
    memset(m.code, 0xff, sizeof m.code);
    const char *alpha = "ACGTN\n";
    for (int k = 0; k < 6; k++) m.code[(unsigned char)alpha[k]] = (uint8_t)k;
    for (int a = 0; a < 6; a++)
        for (int c = 0; c < 6; c++) m.score[a][c] = (int8_t)(a == c ? 3 : -2);
    rc = agx_sw_batch_create_matrix(NULL, &m, t->bases, t->off, t->len, t->n_pairs, &b);
    EXPECT(rc == AGX_OK || rc == AGX_E_SYMBOL || rc == AGX_E_LIMIT); /* split lines / other bytes / long lines */
    agx_sw_batch_destroy(b);
    /* the same file through the chunked reader: same number of pairs */
    agx_sw_reader *r = NULL;
    EXPECT(agx_sw_reader_open(path, line_buf, &r) == AGX_OK && r);
    int64_t n = 0;
    while (r && !agx_sw_reader_done(r)) {
        agx_sw_text *c = NULL;
        EXPECT(agx_sw_reader_next(r, 3, &c) == AGX_OK && c);
        if (!c) break;
        EXPECT(c->n_pairs <= 3 && c->line_num == t->line_num);
        n += c->n_pairs;
        agx_sw_text_free(c);
    }
    EXPECT(n == t->n_pairs);
    agx_sw_reader_close(r);
    agx_sw_text_free(t);
}

static void phmm_file(const char *dir, const char *name)
{
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    agx_phmm_text *t = NULL;
    int rc = agx_phmm_text_read(path, &t);
    EXPECT(rc == AGX_OK && t);
    if (!t) return;
    for (int prec = 0; prec < 4; prec++) {
        agx_phmm_batch *b = NULL;
        rc = agx_phmm_batch_create(NULL, &t->desc, prec | (prec == 1 ? AGX_PHMM_GATK_PRIOR : 0), &b);
        EXPECT(rc == AGX_OK && b);
        agx_phmm_info info;
        if (b) {
            EXPECT(agx_phmm_batch_info(b, &info) == AGX_OK && info.n_pairs == t->n_pairs);
            agx_phmm_batch_destroy(b);
        }
    }
    agx_phmm_text_free(t);
}

/* Synthetic PairHMM batches large enough for the planner's threaded passes (shape counts in the dense table / through
 * the map, waves filled on several pieces of the pair list): plan-only, every precision. */
static uint32_t lcg(uint32_t *s) { return *s = *s * 1664525u + 1013904223u; }
static void phmm_synthetic(uint32_t n_regions, uint32_t reads_per, uint32_t haps_per, uint32_t r_lo, uint32_t r_hi, uint32_t h_lo,
                           uint32_t h_hi, int other_letters)
{
    uint32_t seed = 12345u + n_regions + r_hi + h_hi;
    const uint32_t nr = n_regions * reads_per, nh = n_regions * haps_per;
    uint64_t *roff = calloc(nr + 1, sizeof *roff), *hoff = calloc(nh + 1, sizeof *hoff);
    uint32_t *rreg = calloc(n_regions + 1, sizeof *rreg), *hreg = calloc(n_regions + 1, sizeof *hreg);
    for (uint32_t r = 0; r < nr; r++) roff[r + 1] = roff[r] + r_lo + lcg(&seed) % (r_hi - r_lo + 1);
    for (uint32_t h = 0; h < nh; h++) hoff[h + 1] = hoff[h] + h_lo + lcg(&seed) % (h_hi - h_lo + 1);
    for (uint32_t g = 0; g <= n_regions; g++) {
        rreg[g] = g * reads_per;
        hreg[g] = g * haps_per;
    }
    uint8_t *rb = malloc(roff[nr] + 1), *q = malloc(roff[nr] + 1), *hb = malloc(hoff[nh] + 1);
    for (uint64_t k = 0; k < roff[nr]; k++) {
        rb[k] = (uint8_t)"ACGTN"[lcg(&seed) % 5u];
        q[k] = (uint8_t)(33 + 10 + lcg(&seed) % 30u);
    }
    for (uint64_t k = 0; k < hoff[nh]; k++) hb[k] = (uint8_t)(other_letters ? "ACGTNacgt"[lcg(&seed) % 9u] : "ACGT"[lcg(&seed) % 4u]);
    agx_phmm_desc d;
    memset(&d, 0, sizeof d);
    d.read_bases = rb, d.q_base = q, d.q_ins = q, d.q_del = q, d.q_gcp = q;
    d.read_off = roff, d.n_reads = nr;
    d.hap_bases = hb, d.hap_off = hoff, d.n_haps = nh;
    d.region_read = rreg, d.region_hap = hreg, d.n_regions = n_regions;
    for (int prec = 0; prec < 4; prec++) {
        agx_phmm_batch *b = NULL;
        int rc = agx_phmm_batch_create(NULL, &d, prec, &b);
        EXPECT(rc == AGX_OK && b);
        agx_phmm_info info;
        if (b) {
            EXPECT(agx_phmm_batch_info(b, &info) == AGX_OK && info.n_pairs == (int64_t)n_regions * reads_per * haps_per &&
                   info.padded_cells >= info.cells);
            EXPECT(agx_phmm_batch_launch(b) == AGX_E_NODEVICE);
            agx_phmm_batch_destroy(b);
        }
    }
    free(rb), free(q), free(hb), free(roff), free(hoff), free(rreg), free(hreg);
}

int main(int argc, char **argv)
{
    if (argc != 2) return 2;
    const char *sw[] = {"sw_kat.in", "sw_nofinalnl.in", "sw_150.in", "sw_mixed.in", "sw_long.in", "sw_short.in",
                        "sw_hdr_half.in", "sw_hdr_odd.in", "sw_hdr_big.in", "sw_oddlines.in"};
    for (size_t i = 0; i < sizeof sw / sizeof sw[0]; i++) {
        sw_file(argv[1], sw[i], 0);
        sw_file(argv[1], sw[i], 10000);
        sw_file(argv[1], sw[i], 16); /* tiny buffer: every line splits many times */
    }
    const char *ph[] = {"phmm_test.in", "phmm_10s.in", "phmm_synth.in", "phmm_far.in", "phmm_long.in"};
    for (size_t i = 0; i < sizeof ph / sizeof ph[0]; i++) phmm_file(argv[1], ph[i]);
    phmm_synthetic(40, 48, 14, 50, 150, 280, 380, 0);   /* 26 880 mixed pairs: dense shape table, several pieces */
    phmm_synthetic(40, 48, 14, 50, 150, 280, 380, 1);   /* ... with letters outside ACGT: the other double fill's tables */
    phmm_synthetic(64, 32, 16, 100, 100, 300, 300, 0);  /* 32 768 pairs of one shape */
    phmm_synthetic(30, 30, 30, 1, 1000, 1, 1900, 0);    /* 27 000 pairs over a wide window of lengths: the map */
    /* wrong formats fed to each reader must fail or parse without touching invalid memory */
    agx_phmm_text *pt = NULL;
    char path[1024];
    snprintf(path, sizeof path, "%s/sw_kat.in", argv[1]);
    int rc = agx_phmm_text_read(path, &pt);
    EXPECT(rc == AGX_OK || rc == AGX_E_IO);
    agx_phmm_text_free(pt);
    agx_sw_text *st = NULL;
    snprintf(path, sizeof path, "%s/phmm_10s.in", argv[1]);
    rc = agx_sw_text_read(path, 0, &st);
    EXPECT(rc == AGX_OK);
    if (st) {
        agx_sw_batch *b = NULL;
        rc = agx_sw_batch_create(NULL, st->bases, st->off, st->len, st->n_pairs, &b);
        EXPECT(rc == AGX_OK);
        agx_sw_batch_destroy(b);
    }
    agx_sw_text_free(st);
    EXPECT(agx_sw_text_read("/nonexistent/file", 0, &st) == AGX_E_IO);
    EXPECT(agx_sw_batch_create(NULL, NULL, NULL, NULL, -1, NULL) == AGX_E_ARG);
    printf(fails ? "SANITIZE_DRIVER_FAILED %d\n" : "SANITIZE_DRIVER_OK\n", fails);
    return fails ? 1 : 0;
}
