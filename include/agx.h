/*
 * agx.h -- C-ABI of libagx.so: the MI355X (gfx950) drop-in for the two
 * anti-diagonal DP hot paths of AnteMarusic/Accelerating-Genomics:
 *
 *   - Smith-Waterman affine-gap score-only fill
 *       replaces the per-pair loop body of smithWaterman/antidiagonalSmithWaterman.c:254-348
 *       (and the kernel launch of smithWaterman/hipvers.cpp:470-483)
 *   - PairHMM forward recurrence
 *       replaces pairHMM() of pairHMM/antidiagsPairHMM.c:120-267 (fp64 semantics of
 *       pairHMM/pairHMMmatrix.c:41-66) as called from the batch loop :411-461
 *
 * The reference exposes no library API: its surface is two command lines and
 * one function seam.  This header is what a C host binds
 * instead; the drop-in command lines in accelerating-genomics_amd/host/ are
 * built on exactly these entry points.  Plain C99, plain pointers and sizes,
 * caller-owned buffers, int status (0 = AGX_OK, < 0 = error, text via
 * agx_last_error()).  A context and everything created from it belong to one
 * host thread at a time; distinct contexts may be used from distinct threads.
 *
 * There is no CPU fallback: every compute entry point fails with
 * AGX_E_NODEVICE when no HIP device is usable.
 */
#ifndef AGX_H
#define AGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGX_OK 0
#define AGX_E_ARG (-1)      /* bad argument (NULL pointer, negative count, inconsistent offsets) */
#define AGX_E_NODEVICE (-2) /* no usable HIP device / ordinal out of range */
#define AGX_E_HIP (-3)      /* a HIP runtime call failed; agx_last_error() has the HIP text */
#define AGX_E_NOMEM (-4)    /* host or device allocation failed */
#define AGX_E_SYMBOL (-5)   /* a sequence contains byte 0x00, reserved as the padding symbol */
#define AGX_E_LIMIT (-6)    /* a length exceeds what the kernels support (see AGX_*_MAX_*) */
#define AGX_E_IO (-7)       /* parser: file cannot be opened / is malformed */

/* Smith-Waterman: the shorter sequence of a pair is laid across lanes, at most
 * 64 lanes x AGX_SW_MAX_COLS_PER_LANE columns; the longer one streams.  (The
 * reference CLI cannot produce lines of 1000 bytes or more,
 * antidiagonalSmithWaterman.c:44; hipvers.cpp:40 allows 10000.)  Batches whose
 * shorter sides all fit 64 x 40 = 2560 columns run the packed kernel; a batch
 * with a longer one runs the int32 kernel with its wide classes. */
#define AGX_SW_MAX_COLS_PER_LANE 160
#define AGX_SW_MAX_SHORT_LEN (64 * AGX_SW_MAX_COLS_PER_LANE)
/* PairHMM: haplotype across lanes, read streams.  Up to 64 lanes x AGX_PHMM_MAX_COLS_PER_LANE
 * columns a pair is filled in one pass; longer haplotypes (the reference's line buffer allows
 * 5000, antidiagsPairHMM.c:8,353) are filled in stripes of 1536 columns by one wavefront. */
#define AGX_PHMM_MAX_COLS_PER_LANE 32
#define AGX_PHMM_MAX_HAP_LEN 16384
#define AGX_PHMM_MAX_READ_LEN 4096

/* ------------------------------------------------------------------ runtime */

typedef struct agx_ctx agx_ctx; /* one device + one HIP stream + reusable workspaces */

const char *agx_version(void);
const char *agx_last_error(void); /* thread-local, never NULL */
int agx_device_count(void);       /* >= 0; 0 when HIP reports no device */
/* Marketing name of a device, as hipvers.cpp:388-391 prints it; empty string on failure. */
int agx_device_name(int device, char *buf, size_t buf_len);

int agx_ctx_create(int device, agx_ctx **out);
/* Batches keep their context alive: a context may be destroyed before the batches created from it
 * (its streams and memory pools go with the last of them). */
void agx_ctx_destroy(agx_ctx *ctx);
int agx_ctx_device(const agx_ctx *ctx);
/* The hipStream_t all launches of this context go to (as void*).  A host that
 * already owns a stream (e.g. PyTorch's current stream) may install it. */
void *agx_ctx_stream(const agx_ctx *ctx);
int agx_ctx_set_stream(agx_ctx *ctx, void *hip_stream);
int agx_ctx_sync(agx_ctx *ctx);
/* Options.  The library reads no environment variable; whatever a host wants changed it sets here. */
#define AGX_OPT_SW_KERNEL 1 /* which Smith-Waterman fill runs; all give identical scores */
#define AGX_SW_KERNEL_AUTO 0          /* packed int16 x 2 when the batch's value range allows, else int32 */
#define AGX_SW_KERNEL_INT32 1         /* one pair per lane group, int32 state (BASELINE config 2 as worded) */
#define AGX_SW_KERNEL_PACKED_SIGNED 2 /* two pairs per lane group, signed int16 halves */
#define AGX_SW_KERNEL_PACKED_BIASED 3 /* two pairs per lane group, biased unsigned halves (the default where it fits) */
#define AGX_OPT_SW_PLANNER 2 /* where the Smith-Waterman planner's per-pair passes run; same records, same scores */
#define AGX_SW_PLANNER_AUTO 0   /* on the device for large mixed batches of the packed biased fill, else on the host */
#define AGX_SW_PLANNER_HOST 1   /* always on the host (threaded) */
#define AGX_SW_PLANNER_DEVICE 2 /* on the device whenever the batch-level rules allow it, whatever the batch size */
#define AGX_OPT_PHMM_TRAINS 3 /* read trains in the packed float PairHMM fill (AGX_PHMM_F32_FMA); same results bit for bit */
#define AGX_PHMM_TRAINS_AUTO 0 /* two reads of a region share their lane groups where the batch is large enough for it to pay */
#define AGX_PHMM_TRAINS_OFF 1  /* every read drains before the next enters (the schedule of rounds 1 and 2) */
#define AGX_PHMM_TRAINS_ON 2   /* wherever two reads can share a group, whatever the batch size */
int agx_ctx_set_option(agx_ctx *ctx, int key, int64_t value);
/* Brings the HIP runtime and the process-wide contexts of these devices up (the ones agx_*_devices / agx_*_multi /
 * agx_pairHMM use) without computing anything: about 0.2 s that a host can spend on another thread while it
 * parses its input (both drop-in command lines do).  devices == NULL: devices 0 .. n_devices-1, n_devices <= 0: all. */
int agx_warmup_devices(const int *devices, int n_devices);
/* Page-locked host memory.  Batches built from buffers allocated here are uploaded by DMA straight from
 * the caller's memory (about twice the rate of pageable memory, and asynchronously).  Optional: every
 * entry point accepts ordinary malloc'ed buffers. */
void *agx_host_alloc(size_t bytes); /* NULL on failure */
void agx_host_free(void *p);
/* HIP-event stopwatch on the context's stream (used by bench.py for the roofline figures). */
int agx_ctx_timer_start(agx_ctx *ctx);
int agx_ctx_timer_stop(agx_ctx *ctx, float *elapsed_ms); /* second event + wait + elapsed */
/* The same stopwatch in two halves, for timing a launch inside a longer host-timed step without an extra
 * wait: mark() records the second event and returns at once; elapsed() (after whatever synchronised the
 * stream, e.g. agx_sw_batch_scores) reads the time between the two events. */
int agx_ctx_timer_mark(agx_ctx *ctx);
int agx_ctx_timer_elapsed(agx_ctx *ctx, float *elapsed_ms);

/* ----------------------------------------------------------- Smith-Waterman */

/*
 * Input layout (host memory): sequence k is bases[off[k] .. off[k]+len[k]);
 * pair p aligns sequences 2p and 2p+1 -- the file order of
 * antidiagonalSmithWaterman.c:216-227.  Sequences are raw symbols: the caller
 * decides whether a trailing '\n' belongs to them (the reference CLI keeps it,
 * antidiagonalSmithWaterman.c:229-247; agx_sw_text does the same).  Scoring is
 * the reference's compile-time constants (:40-43): match +1, mismatch -1,
 * gap open -3, gap extend -1.  scores[p] = max over the matrix, >= 0, int32,
 * bit-exact with the reference.
 */
typedef struct agx_sw_batch agx_sw_batch; /* a scheduled batch resident in HBM */

typedef struct agx_sw_info {
    int64_t n_pairs;
    int64_t cells;        /* sum len_a*len_b over pairs, as given (sentinels included) */
    int64_t padded_cells; /* lane-steps x columns actually issued (>= cells) */
    int64_t input_bytes;  /* bytes of the packed device image the kernels read */
    int32_t n_launches;   /* kernel launches per agx_sw_batch_launch() */
    int32_t n_waves;      /* wavefronts over all launches */
    int32_t planned_on_device; /* 1 = the per-pair passes of the planner ran as kernels (AGX_OPT_SW_PLANNER) */
    int32_t reserved;
} agx_sw_info;

/* Scoring of the fill (8f n3: the reference's GPU variants carry these as kernel arguments but
 * ignore them, hipvers.cpp:214).  Values are ADDED to the score, as in the reference's macros
 * (antidiagonalSmithWaterman.c:40-43): the first cell of a gap costs gap_open + gap_extend, every
 * further one gap_extend.  Limits: 1 <= match <= 12, match - 128 <= mismatch <= 0,
 * -1000 <= gap_open, gap_extend <= 0 (scores then fit the kernel's int16 lanes: 12 * 2560 < 32767). */
typedef struct agx_sw_scoring {
    int32_t match, mismatch, gap_open, gap_extend;
} agx_sw_scoring;
/* The reference's compile-time constants: +1, -1, -3, -1. */
#define AGX_SW_SCORING_REFERENCE {1, -1, -3, -1}

/* Validate, pick the lane tiling per pair, pack and copy to the device.  Blocking.
 * ctx may be NULL: the batch is then only planned on the host (no device needed); it answers
 * agx_sw_batch_info() and every other call on it fails with AGX_E_NODEVICE. */
int agx_sw_batch_create(agx_ctx *ctx, const uint8_t *bases, const uint64_t *off, const uint32_t *len,
                        int64_t n_pairs, agx_sw_batch **out);
/* Same with caller-chosen scoring (NULL = the reference's).  Only the reference scoring is pinned
 * against the reference program; other settings are checked against the oracle's parametrised Gotoh. */
int agx_sw_batch_create_scored(agx_ctx *ctx, const agx_sw_scoring *scoring, const uint8_t *bases, const uint64_t *off,
                               const uint32_t *len, int64_t n_pairs, agx_sw_batch **out);
/* Substitution matrix (the reference has no such mode -- its kernel's scoring
 * arguments are ignored, hipvers.cpp:214 -- so results are checked against the oracle's Gotoh with the
 * same matrix: "parity unpinned").  code[] maps an input byte to a symbol number 0..n_symbols-1, or
 * 0xff for bytes outside the alphabet (such input fails with AGX_E_SYMBOL); score[a][b] is added on
 * the diagonal move and must be symmetric (the shorter sequence of a pair is laid across the lanes
 * whichever came first).  Gaps as in agx_sw_scoring.  Shorter side <= 2560 in this mode. */
#define AGX_SW_MATRIX_MAX_SYMBOLS 32
typedef struct agx_sw_matrix {
    int32_t n_symbols; /* 1..32 */
    int32_t gap_open, gap_extend; /* -1000..0 each */
    uint8_t code[256];
    int8_t score[AGX_SW_MATRIX_MAX_SYMBOLS][AGX_SW_MATRIX_MAX_SYMBOLS];
} agx_sw_matrix;
int agx_sw_batch_create_matrix(agx_ctx *ctx, const agx_sw_matrix *matrix, const uint8_t *bases, const uint64_t *off,
                               const uint32_t *len, int64_t n_pairs, agx_sw_batch **out);
/* Enqueue the fill on the context's stream; scores stay in HBM.  Asynchronous. */
int agx_sw_batch_launch(agx_sw_batch *b);
/* Wait for the stream and copy the scores out in the caller's pair order. */
int agx_sw_batch_scores(agx_sw_batch *b, int32_t *scores);
/* Optional: name the page-locked array (agx_host_alloc, n_pairs ints) the scores are wanted in BEFORE launching.  A batch
 * whose records are in file order (a uniform batch: fixed-length reads) then lets every launch write its scores there
 * itself -- one PCIe write per wavefront, no copy kernel behind the fill -- and agx_sw_batch_scores(b, the same pointer)
 * only waits.  Other batches keep the copy; scores == NULL unbinds.  The array must stay valid while it is bound. */
int agx_sw_batch_bind_scores(agx_sw_batch *b, int32_t *scores);
int agx_sw_batch_info(const agx_sw_batch *b, agx_sw_info *info);
void agx_sw_batch_destroy(agx_sw_batch *b);

/* One-shot: create + launch + scores + destroy. */
int agx_sw_score(agx_ctx *ctx, const uint8_t *bases, const uint64_t *off, const uint32_t *len, int64_t n_pairs,
                 int32_t *scores);
/* Same, sharded by cells over `n_devices` devices (<= 0: all visible), one host
 * thread and context per device, no collective.  The per-device contexts and
 * their memory pools are created on first use and kept for later calls. */
int agx_sw_score_multi(int n_devices, const uint8_t *bases, const uint64_t *off, const uint32_t *len,
                       int64_t n_pairs, int32_t *scores);
/* Same with an explicit device list: shard k runs on devices[k].  An ordinal may appear several times
 * (several shards then share that GPU, each with its own context and stream).  Callable from several host
 * threads at once: calls take turns on each (device, shard slot) context. */
int agx_sw_score_devices(const int *devices, int n_devices, const uint8_t *bases, const uint64_t *off, const uint32_t *len,
                         int64_t n_pairs, int32_t *scores);
/* The cut rule of the two calls above, on the host alone: cut[0..n_shards] with shard k = pairs
 * [cut[k], cut[k+1]), contiguous and balanced by sum(len_a * len_b + 1). */
int agx_sw_shard_cuts(const uint32_t *len, int64_t n_pairs, int n_shards, int64_t *cut);

/* ------------------------------------------------------------------ PairHMM */

/*
 * Input layout (host memory), mirroring the reference's regions ("batches",
 * antidiagsPairHMM.c:371-461): read r has length read_off[r+1]-read_off[r]
 * and five byte tracks at that offset (bases, base/insertion/deletion/gcp
 * qualities, Phred+33 characters exactly as in the text file); haplotype h is
 * hap_bases[hap_off[h] .. hap_off[h+1]).  Region g pairs reads
 * [region_read[g], region_read[g+1]) with haplotypes
 * [region_hap[g], region_hap[g+1]).  Results are written region by region,
 * read-major, haplotype-minor -- the order of the reference's output file.
 */
typedef struct agx_phmm_desc {
    const uint8_t *read_bases, *q_base, *q_ins, *q_del, *q_gcp;
    const uint64_t *read_off; /* n_reads + 1 */
    uint32_t n_reads;
    const uint8_t *hap_bases;
    const uint64_t *hap_off; /* n_haps + 1 */
    uint32_t n_haps;
    const uint32_t *region_read; /* n_regions + 1 */
    const uint32_t *region_hap;  /* n_regions + 1 */
    uint32_t n_regions;
} agx_phmm_desc;

/* Arithmetic of the recurrence. */
#define AGX_PHMM_F64 0      /* double, reference expression order, no FMA contraction:
                               the raw sum is bit-identical to pairHMMmatrix.c */
#define AGX_PHMM_F64_FMA 1  /* double with FMA contraction (<= 1e-12 relative on log10) */
#define AGX_PHMM_F32 2      /* float with initial constant FLT_MAX/16 (BASELINE config 3);
                               pairs whose float sum falls below AGX_PHMM_F32_RESCUE are
                               recomputed with AGX_PHMM_F64 on the device */
#define AGX_PHMM_F32_FMA 3  /* float, two haplotypes per lane group on packed FMA instructions: the fastest
                               mode; <= 1e-6 relative on log10 like AGX_PHMM_F32 but not bit-identical to a
                               plain float evaluation; same double rescue of underflowing pairs */
#define AGX_PHMM_F32_RESCUE 1e-28f
/* OR-able into `precision` (8f n4, default off): mismatch prior Qr/3 as GATK's PairHMM uses, instead
 * of the reference's Qr (antidiagsPairHMM.c:111-113).  Not a behaviour of the
 * reference: checked against the oracle's own restatement only. */
#define AGX_PHMM_GATK_PRIOR 0x100

typedef struct agx_phmm_batch agx_phmm_batch;

typedef struct agx_phmm_info {
    int64_t n_pairs;
    int64_t cells;        /* sum R*H */
    int64_t padded_cells; /* lane-steps x columns issued */
    int64_t input_bytes;
    int32_t n_launches;
    int32_t n_waves;
    int64_t n_rescued;    /* F32 only: pairs recomputed in double by the last launch+results */
} agx_phmm_info;

/* ctx may be NULL: plan only, as for agx_sw_batch_create. */
int agx_phmm_batch_create(agx_ctx *ctx, const agx_phmm_desc *d, int precision, agx_phmm_batch **out);
int agx_phmm_batch_launch(agx_phmm_batch *b);
/* log10_lik[k] = log10(sum_k) - log10(C), C = DBL_MAX/16 (FLT_MAX/16 for F32), both log10 taken by the
 * host libm in double exactly as antidiagsPairHMM.c:242; raw_sum (may be NULL) receives sum_k. */
int agx_phmm_batch_results(agx_phmm_batch *b, double *log10_lik, double *raw_sum);
/* Optional, the PairHMM counterpart of agx_sw_batch_bind_scores: name the page-locked array (agx_host_alloc, n_pairs
 * doubles) the log10 likelihoods are wanted in BEFORE launching.  An AGX_PHMM_F32_FMA batch in output order (one read and
 * haplotype length throughout, every pair with work) then lets its fill compute log10(sum) - log10(C) and write it there
 * itself -- no log10 kernel behind the fill -- and agx_phmm_batch_results(b, the same pointer, NULL) only waits, unless a
 * pair went to the double rescue plan.  Other batches keep the usual path; NULL unbinds. */
int agx_phmm_batch_bind_results(agx_phmm_batch *b, double *log10_lik);
int agx_phmm_batch_info(const agx_phmm_batch *b, agx_phmm_info *info);
void agx_phmm_batch_destroy(agx_phmm_batch *b);

int agx_phmm_forward(agx_ctx *ctx, const agx_phmm_desc *d, int precision, double *log10_lik);
int agx_phmm_forward_multi(int n_devices, const agx_phmm_desc *d, int precision, double *log10_lik);
/* Explicit device list, as agx_sw_score_devices. */
int agx_phmm_forward_devices(const int *devices, int n_devices, const agx_phmm_desc *d, int precision, double *log10_lik);
/* The cut rule: cut[0..n_shards] in REGIONS (whole regions stay together), balanced by
 * (read bytes x haplotype bytes + 1) per region. */
int agx_phmm_shard_cuts(const agx_phmm_desc *d, int n_shards, uint32_t *cut);

/*
 * The reference's only function-level seam, same argument list as
 * antidiagsPairHMM.c:120 (M/X/Y scratch is accepted and ignored; *likelihood is
 * overwritten with the log10 value, i.e. started from 0 -- SURVEY.md Q8).
 * Qr/Qi/Qd/Qg are probabilities, not Phred characters.  Runs one pair on
 * device 0 through a process-wide lazily created context; returns nothing, as
 * the reference does: on failure *likelihood = NaN and agx_last_error() is set.
 */
void agx_pairHMM(double *likelihood, double *M, double *X, double *Y, char *R, char *H, int read_len,
                 int haplotype_len, double *Qr, double *Qi, double *Qd, double *Qg);

/* ------------------------------------------------------------- text front end */

/*
 * Readers for the reference's two input formats, producing the flat layouts
 * above (n1 in SURVEY.md 8f).  They follow the reference's reading rules,
 * including its quirks: SW header = number of sequence LINES, fgets with a
 * 1000-byte buffer (longer lines split), newline kept as a symbol, loop ends
 * at the first missing line (antidiagonalSmithWaterman.c:201-227); PairHMM
 * read length = (strlen(line)-4)/5 (antidiagsPairHMM.c:418).
 */
typedef struct agx_sw_text {
    int32_t line_num;   /* header value as atoi() reads it */
    int64_t n_pairs;    /* pairs the reference loop would score */
    uint8_t *bases;
    uint64_t *off;      /* 2*n_pairs */
    uint32_t *len;      /* 2*n_pairs */
    char *dangling;     /* first line of an unpaired trailing pair (the reference echoes it, :225) or NULL */
} agx_sw_text;

int agx_sw_text_read(const char *path, int line_buf /* 0 = reference's 1000 */, agx_sw_text **out);
void agx_sw_text_free(agx_sw_text *t);

/* The same reader in pieces, so a host can score one chunk while the next is being parsed
 * (host/antidiagonalSmithWaterman.c does).  open reads the header; every next() returns a fresh
 * agx_sw_text object, which the caller frees, with up to max_pairs further pairs -- offsets relative to that
 * chunk's bases -- and `dangling` on the chunk that met an unpaired last line; done() turns 1 once
 * the reference's loop would have ended (header count reached or input exhausted). */
typedef struct agx_sw_reader agx_sw_reader;
int agx_sw_reader_open(const char *path, int line_buf, agx_sw_reader **out);
int32_t agx_sw_reader_line_num(const agx_sw_reader *r);
/* how many host threads read and scan a chunk of a regular file (default 0 = the library's thread pool; 1 = the
 * calling thread alone) */
void agx_sw_reader_set_threads(agx_sw_reader *r, int n_threads);
int agx_sw_reader_next(agx_sw_reader *r, int64_t max_pairs, agx_sw_text **out);
int agx_sw_reader_done(const agx_sw_reader *r);
void agx_sw_reader_close(agx_sw_reader *r);

typedef struct agx_phmm_text {
    agx_phmm_desc desc; /* points into storage owned by this object */
    int64_t n_pairs;
    int32_t n_regions_seen; /* header lines read, for the "#batch:" chatter */
    int32_t truncated;      /* 1 = a region ended early, 2 = a header with a negative haplotype count: the reference exits
                             * with failure there ("Error reading haplotypes." / "Memory allocation failed for haplotypes array") */
} agx_phmm_text;

int agx_phmm_text_read(const char *path, agx_phmm_text **out);
void agx_phmm_text_free(agx_phmm_text *t);

/* The same reader in pieces of whole regions -- the reference's batch loop, antidiagsPairHMM.c:371-433,484-489 --
 * so a host can have region k+1 parsed while region k is on the device and region k-1 is printed
 * (host/antidiagsPairHMM.c does).  Every next() returns a fresh agx_phmm_text, which the caller frees, holding the
 * regions up to the first one that brings it to max_pairs pairs (at least one region; indices and offsets
 * relative to the chunk); n_regions_seen counts the header lines read for it and `truncated` marks the chunk
 * that met a region cut short (the reader is done then, as the reference exits there). */
typedef struct agx_phmm_reader agx_phmm_reader;
int agx_phmm_reader_open(const char *path, agx_phmm_reader **out);
int agx_phmm_reader_next(agx_phmm_reader *r, int64_t max_pairs, agx_phmm_text **out);
int agx_phmm_reader_done(const agx_phmm_reader *r);
void agx_phmm_reader_close(agx_phmm_reader *r);

#ifdef __cplusplus
}
#endif
#endif /* AGX_H */
