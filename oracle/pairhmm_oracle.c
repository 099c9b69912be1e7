/*
 * oracle/pairhmm_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's PairHMM forward recurrence
 * (reference: pairHMM/pairHMMmatrix.c:41-66 = the numerical oracle named in
 * BASELINE.json config 5, and pairHMM/antidiagsPairHMM.c:120-267 = the
 * anti-diagonal program the product replaces).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this
 * file; the product never links or loads it.
 *
 * Parity status: PINNED against
 *   - the reference's own known-answer fixture pairHMM/test_set/test.in ->
 *     test.out (-4.485565), kept as tests/golden/phmm_test.in/.out;
 *   - tests/golden/phmm_10s.f.out / phmm_10s.g17.out: output of
 *     oracle/_ref/phmm_matrix_ref (`gcc -O3` of the unmodified reference
 *     source; the .g17 build differs only in the output format string, see
 *     oracle/Makefile) on the reference's corpus test_set/10s.in, 3550 pairs;
 *     tests/test_oracle_pairhmm.py requires bit-equality of all 3550 doubles.
 *
 * The reference is knowingly not GATK-exact (mismatch prior is
 * Qr, not Qr/3); this file is bug-compatible on purpose.
 *
 * Arithmetic order is the reference's, with no FMA contraction (build with
 * -ffp-contract=off, as oracle/Makefile does; plain x86-64 gcc -O3 has no FMA
 * either):  M = p * (mm*Mdiag + (1-Qg)*(Xdiag+Ydiag));  X = Mup*Qi + Xup*Qg;
 * Y = Mleft*Qd + Yleft*Qg;  sum += (M[R][j] + X[R][j]) for j = 1..H.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* quality byte -> probability, reference partition_read (pairHMMmatrix.c:24-29) */
double oracle_phred_to_prob(int c) { return pow(10.0, -(c - 33.0) * 0.1); }

/* prior, reference p() (pairHMMmatrix.c:33-35) */
static inline double prior_d(unsigned char r, unsigned char h, double q)
{
    return (r == h || r == 'N' || h == 'N') ? 1 - q : q;
}
static inline float prior_f(unsigned char r, unsigned char h, float q)
{
    return (r == h || r == 'N' || h == 'N') ? 1 - q : q;
}

/*
 * Row-major, two rolling rows.  Returns the raw sum over the last row
 * (the argument of log10 in likelihood(), pairHMMmatrix.c:59-66).
 */
double oracle_pairhmm_sum_f64(const unsigned char *R, int rl, const unsigned char *H, int hl,
                              const double *Qr, const double *Qi, const double *Qd, const double *Qg)
{
    size_t w = (size_t)hl + 1;
    double *buf = (double *)calloc(6 * w, sizeof(double));
    if (!buf) return NAN;
    double *Mp = buf, *Xp = buf + w, *Yp = buf + 2 * w, *Mc = buf + 3 * w, *Xc = buf + 4 * w, *Yc = buf + 5 * w;
    double init = DBL_MAX / 16 / (double)hl; /* pairHMMmatrix.c:43-46 */
    for (int j = 0; j <= hl; j++) Yp[j] = init;
    for (int i = 1; i <= rl; i++) {
        Mc[0] = Xc[0] = Yc[0] = 0;
        double qr = Qr[i - 1], qi = Qi[i - 1], qd = Qd[i - 1], qg = Qg[i - 1];
        for (int j = 1; j <= hl; j++) {
            Mc[j] = prior_d(R[i - 1], H[j - 1], qr) * ((1 - (qi + qd)) * Mp[j - 1] + (1 - qg) * (Xp[j - 1] + Yp[j - 1]));
            Xc[j] = Mp[j] * qi + Xp[j] * qg;
            Yc[j] = Mc[j - 1] * qd + Yc[j - 1] * qg;
        }
        double *t;
        t = Mp; Mp = Mc; Mc = t;
        t = Xp; Xp = Xc; Xc = t;
        t = Yp; Yp = Yc; Yc = t;
    }
    double l = 0;
    for (int j = 1; j <= hl; j++) l += (Mp[j] + Xp[j]);
    free(buf);
    return l;
}

/*
 * Same recurrence with GATK's mismatch prior Qr/3.  NOT a behaviour of the
 * reference ("parity unpinned"): it only pins the product's AGX_PHMM_GATK_PRIOR option against
 * this restatement.
 */
double oracle_pairhmm_sum_f64_gatk(const unsigned char *R, int rl, const unsigned char *H, int hl,
                                   const double *Qr, const double *Qi, const double *Qd, const double *Qg)
{
    size_t w = (size_t)hl + 1;
    double *buf = (double *)calloc(6 * w, sizeof(double));
    if (!buf) return NAN;
    double *Mp = buf, *Xp = buf + w, *Yp = buf + 2 * w, *Mc = buf + 3 * w, *Xc = buf + 4 * w, *Yc = buf + 5 * w;
    double init = DBL_MAX / 16 / (double)hl;
    for (int j = 0; j <= hl; j++) Yp[j] = init;
    for (int i = 1; i <= rl; i++) {
        Mc[0] = Xc[0] = Yc[0] = 0;
        double qr = Qr[i - 1], qi = Qi[i - 1], qd = Qd[i - 1], qg = Qg[i - 1];
        for (int j = 1; j <= hl; j++) {
            unsigned char r = R[i - 1], h = H[j - 1];
            double prior = (r == h || r == 'N' || h == 'N') ? 1 - qr : qr / 3.0;
            Mc[j] = prior * ((1 - (qi + qd)) * Mp[j - 1] + (1 - qg) * (Xp[j - 1] + Yp[j - 1]));
            Xc[j] = Mp[j] * qi + Xp[j] * qg;
            Yc[j] = Mc[j - 1] * qd + Yc[j - 1] * qg;
        }
        double *t;
        t = Mp; Mp = Mc; Mc = t;
        t = Xp; Xp = Xc; Xc = t;
        t = Yp; Yp = Yc; Yc = t;
    }
    double l = 0;
    for (int j = 1; j <= hl; j++) l += (Mp[j] + Xp[j]);
    free(buf);
    return l;
}

/* log10 likelihood exactly as likelihood() returns it (pairHMMmatrix.c:65) */
double oracle_pairhmm_log10_f64(const unsigned char *R, int rl, const unsigned char *H, int hl,
                                const double *Qr, const double *Qi, const double *Qd, const double *Qg)
{
    return log10(oracle_pairhmm_sum_f64(R, rl, H, hl, Qr, Qi, Qd, Qg)) - log10(DBL_MAX / 16);
}

/*
 * Same recurrence in the anti-diagonal order and storage of
 * antidiagsPairHMM.c:120-267: three live anti-diagonals per matrix, slot =
 * (i+j)%3, position = j while i+j < rl+1 else rl-i (:41-62).  The accumulator
 * starts from 0 (the reference starts from the previous pair's log10, Q8 in
 * SURVEY.md; bitwise equal on the whole corpus).  "port" for cpu_baseline.
 */
static inline int ad_pos(int i, int j, int rl, int w)
{
    int d = i + j;
    return (d % 3) * w + (d >= rl + 1 ? rl - i : j);
}

double oracle_pairhmm_sum_f64_antidiag(const unsigned char *R, int rl, const unsigned char *H, int hl,
                                       const double *Qr, const double *Qi, const double *Qd, const double *Qg)
{
    int minl = rl < hl ? rl : hl;
    int w = minl + 1;
    double *buf = (double *)calloc((size_t)9 * w, sizeof(double)); /* ref main :444-455 */
    if (!buf) return NAN;
    double *M = buf, *X = buf + 3 * w, *Y = buf + 6 * w;
    double init = DBL_MAX / 16 / (double)hl; /* :135 */
    double l = 0;
    int nd = rl + hl + 1;
    for (int d = 0; d < nd; d++) {
        int i = d < rl ? d : rl;
        int j = d - i;
        for (; i >= 0 && j <= hl; i--, j++) {
            int here = ad_pos(i, j, rl, w);
            if (i == 0) { M[here] = 0; X[here] = 0; Y[here] = init; }      /* :157-167 */
            else if (j == 0) { M[here] = 0; X[here] = 0; Y[here] = 0; }    /* :168-178 */
            else {
                int dg = ad_pos(i - 1, j - 1, rl, w), up = ad_pos(i - 1, j, rl, w), lf = ad_pos(i, j - 1, rl, w);
                double qr = Qr[i - 1], qi = Qi[i - 1], qd = Qd[i - 1], qg = Qg[i - 1];
                double m = prior_d(R[i - 1], H[j - 1], qr) * ((1 - (qi + qd)) * M[dg] + (1 - qg) * (X[dg] + Y[dg]));
                double x = M[up] * qi + X[up] * qg;
                double y = M[lf] * qd + Y[lf] * qg;
                M[here] = m; X[here] = x; Y[here] = y;
            }
            if (i == rl) l += M[here] + X[here]; /* :206-212 */
        }
    }
    free(buf);
    return l;
}

/*
 * fp32 restatement for BASELINE config 3.  The reference has no fp32 code
 * (SURVEY.md Q12); this follows the same expression order in float with the
 * initial constant FLT_MAX/16; the last row is summed in double (a float
 * running sum over thousands of columns alone would exceed 1e-6) and the final
 * log10 is taken in double.
 * Its tolerance vs the fp64 oracle is what tests assert (1e-6 relative).
 */
double oracle_pairhmm_sum_f32(const unsigned char *R, int rl, const unsigned char *H, int hl,
                             const double *Qr, const double *Qi, const double *Qd, const double *Qg)
{
    size_t w = (size_t)hl + 1;
    float *buf = (float *)calloc(6 * w, sizeof(float));
    if (!buf) return NAN;
    float *Mp = buf, *Xp = buf + w, *Yp = buf + 2 * w, *Mc = buf + 3 * w, *Xc = buf + 4 * w, *Yc = buf + 5 * w;
    float init = FLT_MAX / 16 / (float)hl;
    for (int j = 0; j <= hl; j++) Yp[j] = init;
    for (int i = 1; i <= rl; i++) {
        Mc[0] = Xc[0] = Yc[0] = 0;
        float qr = (float)Qr[i - 1], qi = (float)Qi[i - 1], qd = (float)Qd[i - 1], qg = (float)Qg[i - 1];
        for (int j = 1; j <= hl; j++) {
            Mc[j] = prior_f(R[i - 1], H[j - 1], qr) * ((1 - (qi + qd)) * Mp[j - 1] + (1 - qg) * (Xp[j - 1] + Yp[j - 1]));
            Xc[j] = Mp[j] * qi + Xp[j] * qg;
            Yc[j] = Mc[j - 1] * qd + Yc[j - 1] * qg;
        }
        float *t;
        t = Mp; Mp = Mc; Mc = t;
        t = Xp; Xp = Xc; Xc = t;
        t = Yp; Yp = Yc; Yc = t;
    }
    double l = 0;
    for (int j = 1; j <= hl; j++) l += (double)(Mp[j] + Xp[j]);
    free(buf);
    return l;
}

/*
 * Flat batch form (same layout the product's C-ABI takes, see include/agx.h):
 * reads r: bases at rb+roff[r], quality tracks qb/qi/qd/qg at the same offsets;
 * haps h: hb+hoff[h]; regions g: reads [rreg[g], rreg[g+1]) x haps
 * [hreg[g], hreg[g+1]); output read-major, hap-minor inside a region, regions
 * in order (antidiagsPairHMM.c:411,440).  variant: 0 row-major f64,
 * 1 antidiag f64, 2 row-major f32, 3 row-major f64 with the GATK mismatch prior.  out_sum receives raw sums (may be NULL),
 * out_log10 the printed quantity.
 */
int oracle_pairhmm_batch(const unsigned char *rb, const unsigned char *qb, const unsigned char *qi_,
                         const unsigned char *qd_, const unsigned char *qg_, const uint64_t *roff,
                         const unsigned char *hb, const uint64_t *hoff, const uint32_t *rreg,
                         const uint32_t *hreg, int n_regions, double *out_sum, double *out_log10, int variant)
{
    double lut[256];
    /* the reference holds quality bytes in plain `char`, signed on x86-64 (antidiagsPairHMM.c:99-107): byte 200 is -56 */
    for (int c = 0; c < 256; c++) lut[c] = oracle_phred_to_prob((signed char)c);
    size_t k = 0;
    for (int g = 0; g < n_regions; g++) {
        for (uint32_t r = rreg[g]; r < rreg[g + 1]; r++) {
            int rl = (int)(roff[r + 1] - roff[r]);
            double *q = (double *)malloc((size_t)(rl > 0 ? rl : 1) * 4 * sizeof(double));
            if (!q) return -1;
            double *Qr = q, *Qi = q + rl, *Qd = q + 2 * rl, *Qg = q + 3 * rl;
            for (int i = 0; i < rl; i++) {
                Qr[i] = lut[qb[roff[r] + i]];
                Qi[i] = lut[qi_[roff[r] + i]];
                Qd[i] = lut[qd_[roff[r] + i]];
                Qg[i] = lut[qg_[roff[r] + i]];
            }
            for (uint32_t h = hreg[g]; h < hreg[g + 1]; h++) {
                int hl = (int)(hoff[h + 1] - hoff[h]);
                double s;
                double c = log10(DBL_MAX / 16);
                if (variant == 0) s = oracle_pairhmm_sum_f64(rb + roff[r], rl, hb + hoff[h], hl, Qr, Qi, Qd, Qg);
                else if (variant == 1) s = oracle_pairhmm_sum_f64_antidiag(rb + roff[r], rl, hb + hoff[h], hl, Qr, Qi, Qd, Qg);
                else if (variant == 3) s = oracle_pairhmm_sum_f64_gatk(rb + roff[r], rl, hb + hoff[h], hl, Qr, Qi, Qd, Qg);
                else { s = (double)oracle_pairhmm_sum_f32(rb + roff[r], rl, hb + hoff[h], hl, Qr, Qi, Qd, Qg); c = log10((double)(FLT_MAX / 16)); }
                if (out_sum) out_sum[k] = s;
                if (out_log10) out_log10[k] = log10(s) - c;
                k++;
            }
            free(q);
        }
    }
    return 0;
}

/*
 * File front end restating the reference main()'s reading rule
 * (pairHMMmatrix.c:160-310 / antidiagsPairHMM.c:371-489): repeated regions
 * "nr nh", nr read lines `bases quals ins del gcp`, nh haplotype lines;
 * read length = (strlen(line)-4)/5 after stripping the newline.
 * Writes up to max_out log10 values; returns the number of pairs, or -1.
 */
#define ORACLE_PHMM_LINE 5001
long oracle_pairhmm_file(const char *path, double *out_log10, double *out_sum, long max_out, int variant)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    char *line = (char *)malloc(ORACLE_PHMM_LINE);
    long k = 0;
    double c = variant == 2 ? log10((double)(FLT_MAX / 16)) : log10(DBL_MAX / 16);
    while (fgets(line, ORACLE_PHMM_LINE, f)) {
        int nr = 0, nh = 0;
        sscanf(line, "%d %d", &nr, &nh);
        char **rl_ = (char **)calloc((size_t)nr + 1, sizeof(char *));
        char **hl_ = (char **)calloc((size_t)nh + 1, sizeof(char *));
        int ok = 1;
        for (int i = 0; i < nr && ok; i++) {
            if (!fgets(line, ORACLE_PHMM_LINE, f)) { ok = 0; break; }
            line[strcspn(line, "\n")] = 0;
            rl_[i] = strdup(line);
        }
        for (int i = 0; i < nh && ok; i++) {
            if (!fgets(line, ORACLE_PHMM_LINE, f)) { ok = 0; break; }
            line[strcspn(line, "\n")] = 0;
            hl_[i] = strdup(line);
        }
        for (int i = 0; i < nr && ok; i++) {
            int len = ((int)strlen(rl_[i]) - 4) / 5;
            if (len < 0) len = 0;
            char *bases = (char *)malloc(5 * ((size_t)strlen(rl_[i]) + 2));
            char *t0 = bases + strlen(rl_[i]) + 1, *t1 = t0 + strlen(rl_[i]) + 1, *t2 = t1 + strlen(rl_[i]) + 1,
                 *t3 = t2 + strlen(rl_[i]) + 1;
            bases[0] = t0[0] = t1[0] = t2[0] = t3[0] = 0;
            sscanf(rl_[i], "%s %s %s %s %s", bases, t0, t1, t2, t3);
            double *q = (double *)malloc((size_t)(len + 1) * 4 * sizeof(double));
            double *Qr = q, *Qi = q + len, *Qd = q + 2 * len, *Qg = q + 3 * len;
            for (int b = 0; b < len; b++) {
                Qr[b] = oracle_phred_to_prob(t0[b]);
                Qi[b] = oracle_phred_to_prob(t1[b]);
                Qd[b] = oracle_phred_to_prob(t2[b]);
                Qg[b] = oracle_phred_to_prob(t3[b]);
            }
            for (int h = 0; h < nh; h++) {
                int hl = (int)strlen(hl_[h]);
                double s;
                if (variant == 0) s = oracle_pairhmm_sum_f64((unsigned char *)bases, len, (unsigned char *)hl_[h], hl, Qr, Qi, Qd, Qg);
                else if (variant == 1) s = oracle_pairhmm_sum_f64_antidiag((unsigned char *)bases, len, (unsigned char *)hl_[h], hl, Qr, Qi, Qd, Qg);
                else s = (double)oracle_pairhmm_sum_f32((unsigned char *)bases, len, (unsigned char *)hl_[h], hl, Qr, Qi, Qd, Qg);
                if (k < max_out) {
                    if (out_log10) out_log10[k] = log10(s) - c;
                    if (out_sum) out_sum[k] = s;
                }
                k++;
            }
            free(q);
            free(bases);
        }
        for (int i = 0; i < nr; i++) free(rl_[i]);
        for (int i = 0; i < nh; i++) free(hl_[i]);
        free(rl_);
        free(hl_);
        if (!ok) break;
    }
    free(line);
    fclose(f);
    return k;
}
