/* scrappie_raw.c -- the `scrappie raw` command line over libscrappie_hip.so.
 *
 * Same options, defaults and output records as the reference's subcommand
 * (src/scrappie_raw.c:40-69 options, :98-121 defaults, :317-331 records), host
 * code in C over the C ABI.  The structure differs on purpose: the reference
 * basecalls one read per OpenMP thread (scrappie_raw.c:355-415); here host
 * threads only read, trim and normalise, reads are gathered into batches and
 * each batch is one engine call (the GPU needs thousands of reads in flight).
 * Output order is input (glob) order, as the reference with one thread.
 *
 * Added options: --batch N, --model-file PATH (weights are data here, see
 * INTEGRATION.md), --device N, --gpus N / --devices a,b,c: one engine and one host
 * thread per GPU, launch groups handed out from an atomic cursor over the reads sorted by
 * length (scrappie_hip_basecall_batch_multi: the analogue of the reference's
 * `schedule(dynamic)` over reads, scrappie_raw.c:355,387).  `--hdf5-*` are accepted and ignored (the dump
 * option they belong to is disabled in the reference too, scrappie_raw.c:54-55).
 */
#define _GNU_SOURCE
#include <dirent.h>
#include <getopt.h>
#include <glob.h>
#include <libgen.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <time.h>

#include "scrappie_hip.h"

#define SCRAPPIE_HIP_VERSION "scrappie (MI355X hot-path build) 0.1.0, interface of scrappie 1.4"

enum outfmt { FMT_FASTA, FMT_SAM };

struct settings {
    enum outfmt fmt;
    int limit;
    FILE *out;
    const char *prefix;
    scrappie_hip_params p;
    int trim_start, trim_end, varseg_chunk;
    float varseg_thresh;
    const char *model;
    const char *model_file;
    int uuid_primary;
    int threads, batch, batch_given, device;
    int ndev, devs[64];          /* --gpus / --devices: the GPUs to spread a batch over (default: --device alone) */
    int prep_device;             /* --prep: 1 = trim_and_segment_raw + medmad_normalise_array on the GPU (k_p0), 0 = on host threads */
    size_t prep_budget;          /* device preparation: samples per GPU and batch the preparers' buffers may hold (SCRAPPIE_PREP_SAMPLES) */
    int stats;                   /* --stats: loader / engine / wall rates on stderr at the end */
};

static void usage(FILE *fh) {
    fputs("Usage: scrappie raw [OPTION...] fast5 [fast5 ...]\n"
          "Scrappie basecaller -- basecall from raw signal\n\n"
          "  -f, --format=format        Format to output reads (FASTA or SAM)\n"
          "  -l, --limit=nreads         Maximum number of reads to call (0 is unlimited)\n"
          "  -m, --min_prob=probability Minimum bound on probability of match\n"
          "  -o, --output=filename      Write to file rather than stdout\n"
          "  -p, --prefix=string        Prefix to append to name of each read\n"
          "  -s, --skip=penalty         Penalty for skipping a base\n"
          "  -y, --stay=penalty         Penalty for staying\n"
          "      --local=penalty        Penalty for local basecalling\n"
          "      --temperature1=factor  Temperature for softmax weights\n"
          "      --temperature2=factor  Temperature for softmax bias\n"
          "  -t, --trim=start:end       Number of samples to trim, as start:end\n"
          "      --slip, --no-slip      Use slipping / disable slipping\n"
          "      --model=name           Raw model to use: \"raw_r94\", \"rgrgr_r94\", \"rgrgr_r941\", \"rgrgr_r10\", \"rnnrf_r94\"\n"
          "      --segmentation=chunk:percentile  Chunk size and percentile for variance based segmentation\n"
          "  -H, --homopolymer=calc     Homopolymer run calc. to use: \"nochange\" or \"mean\" (default). Not implemented for CRF.\n"
          "      --uuid, --no-uuid      Output UUID / read file name\n"
          "  -#, --threads=nparallel    Host threads for reading (and, with --prep=host, normalising); default: 16 per GPU, at most the CPUs of the process\n"
          "      --hdf5-compression=level, --hdf5-chunk=size   accepted, ignored\n"
          "      --licence, --license   Print licensing information\n"
          "      --batch=nreads         Reads per engine call (default 16384 = one launch group of 4000-sample reads; several GPUs with --prep=device: 65536 per GPU).\n"
          "                             A launch group lasts as long as its longest read: long-tailed read lengths want a larger batch (32768-65536)\n"
          "      --model-file=path      Weight container (.scrm); default $SCRAPPIE_MODEL_DIR/<model>.scrm\n"
          "      --device=n             GPU to use (default 0)\n"
          "      --gpus=n               Use the first n GPUs (0 = all visible); reads are handed out dynamically\n"
          "      --devices=a,b,...      Use exactly these GPUs\n"
          "      --prep=device|host     Where reads are trimmed and normalised (default device: k_p0 on the GPU that basecalls the read;\n"
          "                             host: the reference's functions on the loader threads -- with one GPU the chain-bound reads of a\n"
          "                             long-tailed batch then run beside the following batches, with several GPUs reads are handed out dynamically)\n"
          "      --stats                Report loader / engine / wall rates on stderr\n", fh);
}

static int parse_pair(const char *arg, long *a, double *b_or_null, long *b_long) {
    /* "start:end" / "chunk:percentile" */
    char *end = NULL;
    *a = strtol(arg, &end, 10);
    if (!end || *end != ':') return -1;
    if (b_long) *b_long = strtol(end + 1, NULL, 10);
    if (b_or_null) *b_or_null = strtod(end + 1, NULL);
    return 0;
}

static int parse_args(int argc, char **argv, struct settings *s) {
    enum { O_LOCAL = 256, O_T1, O_T2, O_SLIP, O_NOSLIP, O_MODEL, O_SEG, O_UUID, O_NOUUID, O_HC, O_HK, O_LIC,
           O_BATCH, O_MFILE, O_DEV, O_GPUS, O_DEVS, O_PREP, O_STATS };
    static const struct option lo[] = {
        {"format", 1, 0, 'f'}, {"limit", 1, 0, 'l'}, {"min_prob", 1, 0, 'm'}, {"output", 1, 0, 'o'},
        {"prefix", 1, 0, 'p'}, {"skip", 1, 0, 's'}, {"stay", 1, 0, 'y'}, {"local", 1, 0, O_LOCAL},
        {"temperature1", 1, 0, O_T1}, {"temperature2", 1, 0, O_T2}, {"trim", 1, 0, 't'},
        {"slip", 0, 0, O_SLIP}, {"no-slip", 0, 0, O_NOSLIP}, {"model", 1, 0, O_MODEL},
        {"segmentation", 1, 0, O_SEG}, {"homopolymer", 1, 0, 'H'}, {"uuid", 0, 0, O_UUID},
        {"no-uuid", 0, 0, O_NOUUID}, {"threads", 1, 0, '#'}, {"hdf5-compression", 1, 0, O_HC},
        {"hdf5-chunk", 1, 0, O_HK}, {"licence", 0, 0, O_LIC}, {"license", 0, 0, O_LIC},
        {"batch", 1, 0, O_BATCH}, {"model-file", 1, 0, O_MFILE}, {"device", 1, 0, O_DEV}, {"gpus", 1, 0, O_GPUS}, {"devices", 1, 0, O_DEVS},
        {"prep", 1, 0, O_PREP}, {"stats", 0, 0, O_STATS},
        {"help", 0, 0, '?'}, {0, 0, 0, 0}};
    int c;
    long a, bl;
    double bd;
    while ((c = getopt_long(argc, argv, "f:l:m:o:p:s:y:t:H:#:", lo, NULL)) != -1) {
        switch (c) {
        case 'f':
            if (0 == strcasecmp(optarg, "FASTA")) s->fmt = FMT_FASTA;
            else if (0 == strcasecmp(optarg, "SAM")) s->fmt = FMT_SAM;
            else { fprintf(stderr, "scrappie: Unrecognised format '%s'\n", optarg); return -1; }
            break;
        case 'l': s->limit = atoi(optarg); break;
        case 'm': s->p.min_prob = (float)atof(optarg); break;
        case 'o':
            s->out = fopen(optarg, "w");
            if (!s->out) { fprintf(stderr, "scrappie: Failed to open \"%s\" for output.\n", optarg); return -1; }
            break;
        case 'p': s->prefix = optarg; break;
        case 's': s->p.skip_pen = (float)atof(optarg); break;
        case 'y': s->p.stay_pen = (float)atof(optarg); break;
        case O_LOCAL: s->p.local_pen = (float)atof(optarg); break;
        case O_T1: s->p.tempW = (float)atof(optarg); break;
        case O_T2: s->p.tempb = (float)atof(optarg); break;
        case 't':
            if (parse_pair(optarg, &a, NULL, &bl)) bl = a;          /* a single number trims both ends (scrappie_raw.c:160-166) */
            if (a < 0 || bl < 0) { fprintf(stderr, "scrappie: --trim wants start:end\n"); return -1; }
            s->trim_start = (int)a; s->trim_end = (int)bl;
            break;
        case O_SLIP: s->p.use_slip = 1; break;
        case O_NOSLIP: s->p.use_slip = 0; break;
        case O_MODEL:
            if (get_raw_model(optarg) == SCRAPPIE_MODEL_INVALID) { fprintf(stderr, "scrappie: Invalid model name \"%s\"\n", optarg); return -1; }
            s->model = optarg;
            break;
        case O_SEG:
            if (parse_pair(optarg, &a, &bd, NULL) || a <= 1 || bd < 0 || bd > 100) { fprintf(stderr, "scrappie: --segmentation wants chunk:percentile\n"); return -1; }
            s->varseg_chunk = (int)a; s->varseg_thresh = (float)(bd / 100.0);
            break;
        case 'H':
            if (0 == strcmp(optarg, "mean")) s->p.homopolymer = HOMOPOLYMER_MEAN;
            else if (0 == strcmp(optarg, "nochange")) s->p.homopolymer = HOMOPOLYMER_NOCHANGE;
            else { fprintf(stderr, "scrappie: Invalid homopolymer calculation \"%s\"\n", optarg); return -1; }
            break;
        case O_UUID: s->uuid_primary = 1; break;
        case O_NOUUID: s->uuid_primary = 0; break;
        case '#': s->threads = atoi(optarg); break;
        case O_HC: case O_HK: break;
        case O_LIC: puts("Mozilla Public License 2.0 applies to the reference interface this build follows."); exit(EXIT_SUCCESS);
        case O_BATCH: s->batch = atoi(optarg); s->batch_given = 1; break;
        case O_MFILE: s->model_file = optarg; break;
        case O_DEV: s->device = atoi(optarg); break;
        case O_GPUS: {
            int n = atoi(optarg);
            const int have = scrappie_hip_device_count();
            if (n <= 0 || n > have) n = have;
            if (n > 64) n = 64;
            s->ndev = n;
            for (int i = 0; i < n; i++) s->devs[i] = i;
            break;
        }
        case O_DEVS: {
            s->ndev = 0;
            for (char *tok = strtok(optarg, ","); tok && s->ndev < 64; tok = strtok(NULL, ",")) s->devs[s->ndev++] = atoi(tok);
            break;
        }
        case O_PREP:
            if (0 == strcmp(optarg, "device")) s->prep_device = 1;
            else if (0 == strcmp(optarg, "host")) s->prep_device = 0;
            else { fprintf(stderr, "scrappie: --prep wants device or host\n"); return -1; }
            break;
        case O_STATS: s->stats = 1; break;
        default: usage(stderr); return -1;
        }
    }
    return optind;
}

/* expand command-line arguments as the reference does (scrappie_raw.c:363-377):
 * a directory means dir/\*.fast5 (plus the headerless formats of sh_fast5.c) */
static void collect(const char *arg, char ***files, size_t *n, size_t *cap) {
    glob_t gb;
    char *pat = NULL;
    DIR *d = opendir(arg);
    int rc;
    if (d) {
        closedir(d);
        if (asprintf(&pat, "%s/*.fast5", arg) < 0) return;
        rc = glob(pat, GLOB_NOSORT, NULL, &gb);
        free(pat);
        if (asprintf(&pat, "%s/*.[fi][31][26]", arg) >= 0) {
            rc = glob(pat, GLOB_NOSORT | (rc == 0 ? GLOB_APPEND : 0), NULL, &gb) == 0 ? 0 : rc;
            free(pat);
        }
    } else {
        rc = glob(arg, GLOB_NOSORT, NULL, &gb);
    }
    if (rc != 0) {
        fprintf(stderr, "scrappie: File or directory \"%s\" does not exist or no fast5 files found.\n", arg);
        if (rc != GLOB_NOMATCH) return;
        globfree(&gb);
        return;
    }
    for (size_t i = 0; i < gb.gl_pathc; i++) {
        if (*n == *cap) { *cap = *cap ? 2 * *cap : 256; *files = realloc(*files, *cap * sizeof(char *)); }
        (*files)[(*n)++] = strdup(gb.gl_pathv[i]);
    }
    globfree(&gb);
}

/* host side of calculate_post (scrappie_raw.c:270-277) for one batch of files: read_raw on host threads; then
 * trim_and_segment_raw + medmad_normalise_array either on the same threads (--prep=host) or for the whole batch on
 * the GPU (--prep=device: scrappie_hip_prep_run, k_p0), which leaves the prepared signals in device memory */
/* device preparation with several GPUs: read i of a batch belongs to GPU i mod ndev -- its samples are read into that GPU's
 * pinned staging, prepared there and basecalled there (a static, interleaved split: the hand-out from one cursor that
 * scrappie_hip_basecall_batch_multi does for host signals needs the signals on the host) */
struct share {
    scrappie_hip_prep *prep; size_t n;
    raw_table *rts; const float *d_signal; uint64_t *off; uint32_t *len, *st, *en;
    scrappie_hip_call *calls; int rc; char err[256];
    int host_prepared;           /* this share's device preparation failed (memory): its reads were prepared on the host and go through the host-signal entry point */
};
struct loader {
    char **files; size_t base, nb, full; const struct settings *s; raw_table *dst;      /* full: reads in a full batch */
    int nshare, slot; struct share sh[64];             /* device preparation: one share per GPU, the preparers' buffer slot for this batch */
    unsigned char *staged;                              /* the read's samples lie in a preparer's pinned buffer: not ours to free */
    double per_read;                                    /* samples per read seen so far (sizes the pinned buffers) */
    int rc; double read_s, prep_s, prep_ms[3]; size_t nsample;
    /* the engine's and the writer's part of the batch */
    scrappie_hip_call *calls; unsigned char *dflag; long ticket;
    int state;                                          /* ST_EMPTY -> ST_LOADED -> ST_CALLED -> ST_EMPTY (under pipe.mu) */
    struct pipe *pipe; size_t index;                    /* which batch of the run this is */
};
enum { ST_EMPTY = 0, ST_LOADED, ST_CALLED };
#define NRING 5
#define NSLOT 3          /* buffer slots of a preparer: batch k + 1 is prepared while batch k runs and the last launch group of batch k - 1 is still in flight (streaming calls) */
struct pipe {
    pthread_mutex_t mu; pthread_cond_t cv;
    struct loader ring[NRING];
    size_t nbatch, cap, *base, *nb;                     /* the run's batches */
    size_t engine_done;                                 /* batches the engine thread is through with */
    size_t prep_done;                                   /* batches whose signal preparation has run (their pinned staging may be refilled) */
    double eng_t0, eng_t1;                              /* first engine call started / last batch delivered */
    int failed;
    const struct settings *s; scrappie_hip_engine **engs; int *models; int nshare; char **files;
    double per_read, read_s, prep_s, eng_s, first_load_s, prep_ms[3]; size_t nsample;
};
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
/* signal preparation of a batch that has been read: H2D + k_p0 per share (scrappie_hip_prep_run), then the tables keep what the records need */
static void prepare_batch(struct loader *ld) {
    const struct settings *s = ld->s;
    const int K = ld->nshare;
    const double t1 = now_s();
    for (int k = 0; k < K; k++) {                      /* the share's reads, in batch order */
        struct share *sh = &ld->sh[k];
        sh->n = 0;
        for (size_t i = (size_t)k; i < ld->nb; i += (size_t)K) sh->rts[sh->n++] = ld->dst[i];
    }
#if defined(_OPENMP)
#pragma omp parallel for num_threads(K) schedule(static, 1)
#endif
    for (int k = 0; k < K; k++) {
        struct share *sh = &ld->sh[k];
        sh->rc = scrappie_hip_prep_run(sh->prep, ld->slot, sh->rts, sh->n, (size_t)s->trim_start, (size_t)s->trim_end, (size_t)s->varseg_chunk,
                                       s->varseg_thresh, &sh->d_signal, sh->off, sh->len, sh->st, sh->en);
        if (sh->rc) snprintf(sh->err, sizeof sh->err, "%s", scrappie_hip_last_error());
    }
    for (int j = 0; j < 3; j++) ld->prep_ms[j] = 0;
    for (int k = 0; k < K; k++) {
        struct share *sh = &ld->sh[k];
        double ms[3];
        scrappie_hip_prep_timing(sh->prep, ld->slot, ms);
        for (int j = 0; j < 3; j++) ld->prep_ms[j] += ms[j] / K;
    }
    for (int k = 0; k < K; k++) {
        /* a preparer could not run its share of this batch (its device or pinned buffers did not grow: reads much longer than the ones the
         * buffers were reserved for).  The share is NOT lost: a failed scrappie_hip_prep_run leaves the samples where they were (staging or
         * malloc'd), so the reference's own functions prepare them here, on the loader team (as --prep=host does), and the share's engine
         * takes them through its host-signal entry point; the other GPUs' shares go on as usual */
        struct share *sh = &ld->sh[k];
        sh->host_prepared = 0;
        if (!sh->rc) continue;
        fprintf(stderr, "scrappie: %s; device preparation failed for %zu reads, preparing them on the host\n", sh->err, sh->n);
#if defined(_OPENMP)
#pragma omp parallel for schedule(dynamic, 16) num_threads(s->threads)
#endif
        for (size_t j = 0; j < sh->n; j++) {
            const size_t i = (size_t)k + j * (size_t)K;
            raw_table rt = ld->dst[i];
            if (rt.raw && ld->staged[i]) {              /* the staging buffer is the next batch's soon */
                float *own = malloc(rt.n * sizeof(float));
                if (own) memcpy(own, rt.raw, rt.n * sizeof(float));
                else { free(rt.uuid); memset(&rt, 0, sizeof rt); }
                rt.raw = own;
            }
            ld->staged[i] = 0;
            if (rt.raw) {
                char *uuid = rt.uuid;
                rt = trim_and_segment_raw(rt, (size_t)s->trim_start, (size_t)s->trim_end, (size_t)s->varseg_chunk, s->varseg_thresh);
                if (rt.raw) medmad_normalise_array(rt.raw + rt.start, rt.end - rt.start);
                else free(uuid);
            }
            ld->dst[i] = rt; sh->rts[j] = rt;
        }
        sh->rc = 0; sh->host_prepared = 1;
    }
    ld->rc = 0;
    for (size_t i = 0; i < ld->nb; i++) {               /* the samples live on the device now; the table keeps what the records need */
        raw_table *rt = &ld->dst[i];
        const struct share *sh = &ld->sh[i % (size_t)K];
        const size_t j = i / (size_t)K;
        if (sh->host_prepared) continue;                /* (prepared signal in host memory: freed with the record) */
        if (!ld->staged[i]) free(rt->raw);
        if (ld->rc == 0 && sh->len[j]) { rt->raw = NULL; rt->start = sh->st[j]; rt->end = sh->en[j]; }
        else { free(rt->uuid); memset(rt, 0, sizeof *rt); }
    }
    ld->prep_s = now_s() - t1;
}

static void *load_batch(void *arg) {
    struct loader *ld = arg;
    const struct settings *s = ld->s;
    const int K = ld->nshare;                          /* 0: preparation on the host */
    struct pipe *P = ld->pipe;
    if (K && P && ld->index >= NSLOT) {                 /* the slot's pinned staging still feeds the preparation of batch index - NSLOT until that has run */
        pthread_mutex_lock(&P->mu);
        while (P->prep_done + NSLOT - 1 < ld->index && !P->failed) pthread_cond_wait(&P->cv, &P->mu);
        pthread_mutex_unlock(&P->mu);
    }
    const double t0 = now_s();
    size_t nsample = 0;
    /* device preparation: the loader threads read straight into the slot's pinned staging buffer (no copy between the file and
     * the DMA); its size follows the reads seen so far, and a read that does not fit any more is malloc'd and gathered later */
    void *stage[64];
    for (int k = 0; k < K; k++)
    {
        double want = 1.25 * ld->per_read * (double)((ld->full + K - 1) / K);      /* (sized for a full batch at once: the slot grows once, not with every step of the ramp) */
        if (s->prep_budget && want > 1.25 * (double)s->prep_budget) want = 1.25 * (double)s->prep_budget;      /* (reads that do not fit are malloc'd and gathered by the preparer) */
        stage[k] = ld->per_read > 0 ? scrappie_hip_prep_begin(ld->sh[k].prep, ld->slot, (size_t)want + 65536) : NULL;
    }
#if defined(_OPENMP)
#pragma omp parallel for schedule(dynamic, 16) num_threads(s->threads) reduction(+:nsample)
#endif
    for (size_t i = 0; i < ld->nb; i++) {
        const int k = K ? (int)(i % (size_t)K) : 0;
        raw_table rt = scrappie_hip_read_raw_into(ld->files[ld->base + i], true, (K && stage[k]) ? scrappie_hip_prep_alloc : NULL, K ? stage[k] : NULL);
        if (rt.raw) nsample += rt.n;
        ld->staged[i] = (rt.raw && K && scrappie_hip_prep_owns(ld->sh[k].prep, ld->slot, rt.raw)) ? 1 : 0;
        if (rt.raw && !K) {
            char *uuid = rt.uuid;
            rt = trim_and_segment_raw(rt, (size_t)s->trim_start, (size_t)s->trim_end, (size_t)s->varseg_chunk, s->varseg_thresh);
            if (rt.raw) medmad_normalise_array(rt.raw + rt.start, rt.end - rt.start);
            else free(uuid);
        }
        ld->dst[i] = rt;
    }
    ld->read_s = now_s() - t0;
    ld->rc = 0; ld->prep_s = 0; ld->nsample = nsample;
    for (int j = 0; j < 3; j++) ld->prep_ms[j] = 0;
    if (K && ld->nb && (double)nsample / (double)ld->nb > ld->per_read) ld->per_read = (double)nsample / (double)ld->nb;
    if (K > 1 || (K && !P)) {
        /* several GPUs: their engine calls block until a batch is complete, so the preparation runs HERE, beside them (one GPU: on the
         * engine thread, between two streaming calls -- the GPU has the previous call's last launch group to work on meanwhile) */
        if (P && ld->index >= NSLOT) {                  /* the slot's device buffer still belongs to batch index - NSLOT until the engine is through with it */
            pthread_mutex_lock(&P->mu);
            while (P->engine_done + NSLOT - 1 < ld->index && !P->failed) pthread_cond_wait(&P->cv, &P->mu);
            pthread_mutex_unlock(&P->mu);
        }
        prepare_batch(ld);
        if (P) { pthread_mutex_lock(&P->mu); P->prep_done = ld->index + 1; pthread_cond_broadcast(&P->cv); pthread_mutex_unlock(&P->mu); }
    }
    return NULL;
}

static void pipe_fail(struct pipe *P) { pthread_mutex_lock(&P->mu); P->failed = 1; pthread_cond_broadcast(&P->cv); pthread_mutex_unlock(&P->mu); }

static void *loader_main(void *arg) {
    struct pipe *P = arg;
    for (size_t k = 0; k < P->nbatch; k++) {
        struct loader *ld = &P->ring[k % NRING];
        pthread_mutex_lock(&P->mu);
        while (ld->state != ST_EMPTY && !P->failed) pthread_cond_wait(&P->cv, &P->mu);
        const int failed = P->failed;
        pthread_mutex_unlock(&P->mu);
        if (failed) return NULL;
        ld->base = P->base[k]; ld->nb = P->nb[k]; ld->slot = (int)(k % NSLOT); ld->pipe = P; ld->index = k;
        if (P->per_read > ld->per_read) ld->per_read = P->per_read;
        load_batch(ld);
        if (ld->rc) { fprintf(stderr, "scrappie: signal preparation failed\n"); pipe_fail(P); return NULL; }
        pthread_mutex_lock(&P->mu);
        if (ld->per_read > P->per_read) P->per_read = ld->per_read;
        P->read_s += ld->read_s; P->nsample += ld->nsample;
        if (ld->nshare != 1) { P->prep_s += ld->prep_s; for (int j = 0; j < 3; j++) P->prep_ms[j] += ld->prep_ms[j]; }
        if (k == 0) P->first_load_s = ld->read_s + ld->prep_s;
        ld->state = ST_LOADED;
        pthread_cond_broadcast(&P->cv);
        pthread_mutex_unlock(&P->mu);
    }
    return NULL;
}

/* one GPU's share of a prepared batch through its engine */
struct share_call { scrappie_hip_engine *e; int model; struct share *sh; const scrappie_hip_params *p; };
static void *run_share(void *arg);

static void *engine_main(void *arg) {
    struct pipe *P = arg;
    const struct settings *s = P->s;
    const int nshare = P->nshare;
    for (size_t k = 0; k < P->nbatch; k++) {
        struct loader *ld = &P->ring[k % NRING];
        pthread_mutex_lock(&P->mu);
        while (ld->state != ST_LOADED && !P->failed) pthread_cond_wait(&P->cv, &P->mu);
        const int failed = P->failed;
        pthread_mutex_unlock(&P->mu);
        if (failed) return NULL;
        const size_t nb = ld->nb;
        scrappie_hip_call *calls = ld->calls;
        long ticket = 0;
        memset(ld->dflag, 0, nb);
        if (nshare == 1) {               /* one GPU: the batch is prepared here, between two streaming calls (the slot's device buffer is free: the
                                          * previous call has delivered batch k - 2, and this slot last held batch k - 3) */
            prepare_batch(ld);
            pthread_mutex_lock(&P->mu);
            P->prep_done = k + 1; P->prep_s += ld->prep_s;
            for (int j = 0; j < 3; j++) P->prep_ms[j] += ld->prep_ms[j];
            pthread_cond_broadcast(&P->cv);
            pthread_mutex_unlock(&P->mu);
            if (ld->rc) { fprintf(stderr, "scrappie: signal preparation failed\n"); pipe_fail(P); return NULL; }
        }
        const double te0 = now_s();
        if (k == 0) P->eng_t0 = te0;
        if (nshare == 1 && ld->sh[0].host_prepared) {      /* (device preparation failed for this batch: host-prepared signals, as with --prep=host) */
            ticket = scrappie_hip_basecall_batch_deferred(P->engs[0], P->models[0], ld->dst, nb, &s->p, calls, ld->dflag);
            if (ticket < 0) fprintf(stderr, "scrappie: %s\n", scrappie_hip_last_error());
        } else if (nshare == 1) {        /* prepared on the GPU; chain-bound reads deferred; the call's last launch group is left running and is
                                          * delivered behind the next batch's first launch (the engine's pipeline does not drain between batches) */
            struct share *sh = &ld->sh[0];
            ticket = scrappie_hip_basecall_device_deferred_stream(P->engs[0], P->models[0], sh->d_signal, sh->off, sh->len, nb, &s->p, calls, ld->dflag);
            if (ticket < 0) fprintf(stderr, "scrappie: %s\n", scrappie_hip_last_error());
        } else if (nshare) {             /* prepared on the GPUs: every engine basecalls its share, on a host thread of its own */
            struct share_call sc[64];
            pthread_t st[64];
            int live[64];
            for (int d = 0; d < nshare; d++) {
                sc[d] = (struct share_call){P->engs[d], P->models[d], &ld->sh[d], &s->p};
                live[d] = d > 0 && 0 == pthread_create(&st[d], NULL, run_share, &sc[d]);
            }
            for (int d = 0; d < nshare; d++) if (!live[d]) run_share(&sc[d]);
            for (int d = 1; d < nshare; d++) if (live[d]) pthread_join(st[d], NULL);
            for (int d = 0; d < nshare; d++) if (ld->sh[d].rc) { fprintf(stderr, "scrappie: GPU %d: %s\n", s->devs[d], ld->sh[d].err); ticket = -1; }
            if (ticket == 0) for (size_t i = 0; i < nb; i++) calls[i] = ld->sh[i % (size_t)nshare].calls[i / (size_t)nshare];     /* (the strings move to calls[]) */
            else for (int d = 0; d < nshare; d++) if (!ld->sh[d].rc) scrappie_hip_free_calls(ld->sh[d].calls, ld->sh[d].n);
        } else if (s->ndev == 1) {
            ticket = scrappie_hip_basecall_batch_deferred(P->engs[0], P->models[0], ld->dst, nb, &s->p, calls, ld->dflag);
            if (ticket < 0) fprintf(stderr, "scrappie: %s\n", scrappie_hip_last_error());
        } else if (scrappie_hip_basecall_batch_multi(P->engs, P->models, (size_t)s->ndev, ld->dst, nb, &s->p, calls) != 0) {
            fprintf(stderr, "scrappie: %s\n", scrappie_hip_last_error());
            ticket = -1;
        }
        double dt = now_s() - te0;
        if (ticket < 0) { pipe_fail(P); return NULL; }
        /* how far the calls are complete: with a launch group of this batch still in flight, up to the batch before it */
        size_t complete = k + 1;
        if (nshare == 1 && scrappie_hip_stream_pending(P->engs[0])) {
            if (k + 1 == P->nbatch) {    /* the last batch: nothing will come behind it */
                const double tf0 = now_s();
                if (scrappie_hip_stream_flush(P->engs[0]) != 0) { fprintf(stderr, "scrappie: %s\n", scrappie_hip_last_error()); pipe_fail(P); return NULL; }
                dt += now_s() - tf0;
            } else complete = k;
        }
        pthread_mutex_lock(&P->mu);
        ld->ticket = ticket;
        P->eng_s += dt;
        P->eng_t1 = now_s();
        for (size_t j = P->engine_done; j < complete; j++) P->ring[j % NRING].state = ST_CALLED;
        P->engine_done = complete;
        pthread_cond_broadcast(&P->cv);
        pthread_mutex_unlock(&P->mu);
    }
    return NULL;
}

static void *run_share(void *arg) {
    struct share_call *c = arg;
    struct share *sh = c->sh;
    sh->rc = sh->host_prepared ? scrappie_hip_basecall_batch(c->e, c->model, sh->rts, sh->n, c->p, sh->calls)
                               : scrappie_hip_basecall_device(c->e, c->model, sh->d_signal, sh->off, sh->len, sh->n, c->p, sh->calls);
    if (sh->rc) snprintf(sh->err, sizeof sh->err, "%s", scrappie_hip_last_error());
    return NULL;
}

static void write_record(const struct settings *s, char **line, size_t *buflen, const char *fn, const raw_table *rt, const scrappie_hip_call *c) {
    const size_t need = c->basecall_length + strlen(fn) * 2 + 1024;
    if (need > *buflen) { *buflen = 2 * need; *line = realloc(*line, *buflen); }
    char *fcopy = strdup(fn);
    const char *rn = basename(fcopy);
    if (s->fmt == FMT_FASTA)
        scrappie_hip_format_fasta(*line, *buflen, rt->uuid, rn, s->uuid_primary, s->prefix, c, rt->n, rt->start, rt->end);
    else
        scrappie_hip_format_sam(*line, *buflen, rt->uuid, rn, s->uuid_primary, s->prefix, c);
    fputs(*line, s->out);
    free(fcopy);
}

/* deferred (chain-bound) reads of earlier batches: their signals and names are kept until their ticket is in */
struct pending { long ticket; size_t n; raw_table *rts; char **fn; struct pending *next; };
static int drain_pending(struct pending **head, scrappie_hip_engine *e, const struct settings *s, char **line, size_t *buflen, int wait) {
    struct pending **pp = head;
    while (*pp) {
        struct pending *pd = *pp;
        scrappie_hip_call *calls = calloc(pd->n ? pd->n : 1, sizeof *calls);
        const long k = scrappie_hip_deferred_collect(e, pd->ticket, calls, pd->n, wait);
        if (k == -2) { free(calls); pp = &pd->next; continue; }
        if (k < 0) { fprintf(stderr, "scrappie: %s\n", scrappie_hip_last_error()); free(calls); return -1; }
        for (size_t i = 0; i < pd->n; i++) {
            if (!calls[i].basecall) fprintf(stderr, "scrappie: No basecall returned for %s\n", pd->fn[i]);
            else write_record(s, line, buflen, pd->fn[i], &pd->rts[i], &calls[i]);
            free(pd->rts[i].raw); free(pd->rts[i].uuid);
        }
        scrappie_hip_free_calls(calls, pd->n);
        free(calls);
        *pp = pd->next;
        free(pd->rts); free(pd->fn); free(pd);
    }
    return 0;
}

int main_raw(int argc, char **argv) {
    struct settings s;
    memset(&s, 0, sizeof s);
    s.fmt = FMT_FASTA; s.out = stdout; s.prefix = "";
    s.p = scrappie_hip_default_params();
    s.trim_start = 200; s.trim_end = 10; s.varseg_chunk = 100; s.varseg_thresh = 0.0f;
    s.model = "rgrgr_r94"; s.threads = 0;      /* (0: 16 loader threads per GPU, at most the CPUs the process may use) */ s.batch = 16384; s.ndev = 0; s.prep_device = -1;
    const int first = parse_args(argc, argv, &s);
    if (first < 0) return EXIT_FAILURE;
    if (first >= argc) { usage(stderr); return EXIT_FAILURE; }

    char **files = NULL;
    size_t nfile = 0, cap = 0;
    for (int i = first; i < argc; i++) collect(argv[i], &files, &nfile, &cap);
    if (s.limit > 0 && nfile > (size_t)s.limit) nfile = (size_t)s.limit;
    if (nfile == 0) return EXIT_SUCCESS;

    if (s.ndev <= 0) { s.ndev = 1; s.devs[0] = s.device; }
    char *mpath = NULL;
    if (s.model_file) mpath = strdup(s.model_file);
    else if (getenv("SCRAPPIE_MODEL_DIR")) { if (asprintf(&mpath, "%s/%s.scrm", getenv("SCRAPPIE_MODEL_DIR"), s.model) < 0) mpath = NULL; }
    scrappie_hip_engine *engs[64];
    int models[64];
    for (int k = 0; k < s.ndev; k++) {          /* one engine per GPU (no GPU: fail here, there is no CPU path) */
        engs[k] = scrappie_hip_engine_create(s.devs[k]);
        if (!engs[k]) { fprintf(stderr, "scrappie: %s\n", scrappie_hip_last_error()); return EXIT_FAILURE; }
    }
    if (!mpath) { fprintf(stderr, "scrappie: no weights for model %s: give --model-file or set SCRAPPIE_MODEL_DIR\n", s.model); return EXIT_FAILURE; }
    for (int k = 0; k < s.ndev; k++) {          /* the weights replicated */
        models[k] = scrappie_hip_load_model(engs[k], s.model, mpath);
        if (models[k] < 0) { fprintf(stderr, "scrappie: %s\n", scrappie_hip_last_error()); return EXIT_FAILURE; }
    }
    free(mpath);
    if (s.threads <= 0) {                /* reading is what the host does: 16 loader threads per GPU (fast5 input: 12 feed 0.88 of an engine's rate, 16 all of it --
                                          * profiles/r6_cli_rate.txt), within the CPUs the process may use (affinity, cgroup quota, / LOCAL_WORLD_SIZE); at least 2 */
        const unsigned hb = scrappie_hip_host_cpu_budget(), want = 16u * (unsigned)s.ndev;
        s.threads = (int)(hb < 2 ? 2 : hb < want ? hb : want);
    }
    if (s.batch < 1) s.batch = 1;
    /* several GPUs: scrappie_hip_plan_dynamic cuts a call into launch groups of n / (4 GPUs) reads, at least 4096
     * (a GPU needs ~256 tiles of 16 reads to fill its CUs): 16384 reads per GPU and call give every engine four
     * groups, so that the dynamic hand-out can balance and only the last group of a call drains a pipeline */
    if (s.ndev > 1 && !s.batch_given) s.batch = 16384 * s.ndev;
    /* prepared on the device: one GPU streams its calls (a call's last launch group is delivered behind the next call's first), so a batch is
     * one full launch group; several GPUs do not stream and get four launch groups per call and GPU (only the last one drains a pipeline) */
    if (s.prep_device != 0 && !s.batch_given) s.batch = s.ndev == 1 ? 16384 : 65536 * s.ndev;

    if (s.prep_device < 0) s.prep_device = 1;
    scrappie_hip_prep *preps[64] = {0};
    size_t n0 = 0;                       /* samples of the first file: what the preparers' buffers and the engines' arenas are sized for */
    if (s.prep_device) {
        for (int k = 0; k < s.ndev; k++) {
            preps[k] = scrappie_hip_prep_create(s.devs[k]);
            if (!preps[k]) { fprintf(stderr, "scrappie: %s\n", scrappie_hip_last_error()); return EXIT_FAILURE; }
        }
        /* A device-prepared batch is bounded by SAMPLES, not reads: a preparer holds NSLOT slots of (pinned staging + device signal + device
         * scratch) of 1.25 x the batch's samples each.  With 4000-sample reads a full batch is 65 M samples per GPU (0.3 GB per buffer); with
         * real reads of 40 000+ samples the same number of reads would be 13 GB per buffer, 9 buffers per GPU, beside the engine's arena.
         * SCRAPPIE_PREP_SAMPLES (default 2^28 = 1 GiB per buffer before the 1.25) caps samples per GPU and batch; the buffers are then
         * RESERVED before the clock starts, a reservation that fails halves the batch, and below 256 reads per GPU the run prepares on the
         * host (scrappie_raw.c:540). */
        raw_table r0 = scrappie_hip_read_raw(files[0], true);
        n0 = r0.raw ? r0.n : 0;
        free(r0.raw); free(r0.uuid);
        size_t budget = (size_t)1 << 28;
        if (getenv("SCRAPPIE_PREP_SAMPLES") && atof(getenv("SCRAPPIE_PREP_SAMPLES")) >= 1.0) budget = (size_t)atof(getenv("SCRAPPIE_PREP_SAMPLES"));
        s.prep_budget = budget;
        size_t per_gpu = ((size_t)s.batch + (size_t)s.ndev - 1) / (size_t)s.ndev;
        if (per_gpu > nfile / (size_t)s.ndev + 1) per_gpu = nfile / (size_t)s.ndev + 1;
        if (n0 && per_gpu * n0 > budget) {
            per_gpu = budget / n0 > 256 ? budget / n0 : 256;
            fprintf(stderr, "scrappie: reads of ~%zu samples: batches of %zu reads per GPU (SCRAPPIE_PREP_SAMPLES = %zu samples per GPU and batch)\n", n0, per_gpu, budget);
            s.batch = (int)(per_gpu * (size_t)s.ndev);
        }
        while (n0) {
            int ok = 1;
            for (int k = 0; k < NSLOT && ok; k++)
                for (int d = 0; d < s.ndev && ok; d++)
                    if (scrappie_hip_prep_reserve(preps[d], k, (size_t)(1.25 * (double)n0 * (double)per_gpu) + 65536) != 0) ok = 0;
            if (ok) break;
            if (per_gpu <= 256) {       /* not even small batches fit beside what else is on the device: the reference's functions on the loader threads */
                fprintf(stderr, "scrappie: %s; preparing signals on the host\n", scrappie_hip_last_error());
                for (int d = 0; d < s.ndev; d++) { scrappie_hip_prep_destroy(preps[d]); preps[d] = NULL; }
                s.prep_device = 0;
                if (!s.batch_given) s.batch = s.ndev > 1 ? 16384 * s.ndev : 16384;
                break;
            }
            per_gpu = per_gpu / 2 > 256 ? per_gpu / 2 : 256;
            s.batch = (int)(per_gpu * (size_t)s.ndev);
            fprintf(stderr, "scrappie: %s; batches of %zu reads per GPU\n", scrappie_hip_last_error(), per_gpu);
        }
    }
    const int nshare = s.prep_device ? s.ndev : 0;

    const size_t B = (size_t)s.batch;
    size_t buflen = 1 << 16;
    char *line = malloc(buflen);
    /* Three stages, a host thread each, over a ring of batches: the LOADER reads (and prepares) batch k + 1 while
     * the ENGINE thread has batch k on the GPU(s) and this thread WRITES the records of batch k - 1 -- the GPU never waits for a
     * record to be formatted, nor the loader for the GPU, as long as each stage keeps up.  Batch k uses the preparers' buffer slot
     * k & 1: its preparation (which overwrites the slot's device buffer) waits until the engine is through with batch k - 2. */
    struct pipe P;
    memset(&P, 0, sizeof P);
    pthread_mutex_init(&P.mu, NULL); pthread_cond_init(&P.cv, NULL);
    P.s = &s; P.engs = engs; P.models = models; P.nshare = nshare; P.files = files;
    /* batch boundaries: a small first batch, so that the GPU starts while the next is being read, then growth by factors of two
     * up to the full size (a batch is read while the one before it is on the GPU: neither waits long while the pipeline fills) */
    /* (by factors of two: the loader is not much faster than the engine, so with factors of four the GPU idles ~0.25 s of a 400 000-read run) */
    {
        size_t base = 0, nb = (B > 2048 && nfile > B) ? 2048 : B;
        while (base < nfile) {
            if (nb > nfile - base) nb = nfile - base;
            if (P.nbatch == P.cap) { P.cap = P.cap ? 2 * P.cap : 64; P.base = realloc(P.base, P.cap * sizeof *P.base); P.nb = realloc(P.nb, P.cap * sizeof *P.nb); }
            P.base[P.nbatch] = base; P.nb[P.nbatch] = nb; P.nbatch++;
            base += nb;
            nb = (2 * nb < B) ? 2 * nb : B;
            if (nb < 2048) nb = (B < 2048) ? B : 2048;
        }
    }
    for (int k = 0; k < NRING; k++) {
        struct loader *ld = &P.ring[k];
        ld->files = files; ld->s = &s; ld->nshare = nshare; ld->full = (nfile < B) ? nfile : B;
        ld->dst = calloc(B, sizeof(raw_table)); ld->staged = calloc(B, 1); ld->per_read = 0.0;
        ld->calls = calloc(B, sizeof(scrappie_hip_call)); ld->dflag = calloc(B, 1);
        for (int d = 0; d < nshare; d++) {
            struct share *sh = &ld->sh[d];
            const size_t cap = (B + (size_t)nshare - 1) / (size_t)nshare;
            sh->prep = preps[d];
            sh->rts = calloc(cap, sizeof(raw_table)); sh->calls = calloc(cap, sizeof(scrappie_hip_call));
            sh->off = calloc(cap, sizeof(uint64_t)); sh->len = calloc(cap, sizeof(uint32_t));
            sh->st = calloc(cap, sizeof(uint32_t)); sh->en = calloc(cap, sizeof(uint32_t));
        }
    }
    struct pending *pend = NULL;
    size_t nbases = 0, ncalled = 0;
    if (nshare) {
        /* before the clock starts, like engine creation and the model load: the engines' arenas for full launch groups of reads as long as
         * the first file's (allocations of gigabytes stall the device: made piecemeal by the first calls they cost a short run a third of its time) */
        if (n0) {
            const size_t per_gpu = (P.ring[0].full + (size_t)nshare - 1) / (size_t)nshare;
            P.per_read = (double)n0;                    /* (the preparers' slots were reserved above, where the batch size was settled) */
            for (int d = 0; d < nshare; d++)
                if (scrappie_hip_warm_up(engs[d], models[d], per_gpu < 16384 ? per_gpu : 16384, n0) != 0)
                    fprintf(stderr, "scrappie: warm-up: %s\n", scrappie_hip_last_error());
        }
    }
    const double wall0 = now_s();
    pthread_t th_load, th_eng;
    if (pthread_create(&th_load, NULL, loader_main, &P) != 0 || pthread_create(&th_eng, NULL, engine_main, &P) != 0) {
        fprintf(stderr, "scrappie: cannot start the loader / engine threads\n");
        return EXIT_FAILURE;
    }
    /* One GPU: the batch's chain-bound reads (a long tail of read lengths) are left running on the engine's helper while the
     * next batches go on; their records are written when they are ready -- like the reference's OpenMP loop, whose records
     * appear in completion order (scrappie_raw.c:377,402) */
    int rc = EXIT_SUCCESS;
    for (size_t k = 0; k < P.nbatch; k++) {
        struct loader *ld = &P.ring[k % NRING];
        pthread_mutex_lock(&P.mu);
        while (ld->state != ST_CALLED && !P.failed) pthread_cond_wait(&P.cv, &P.mu);
        const int failed = P.failed;
        pthread_mutex_unlock(&P.mu);
        if (failed) { rc = EXIT_FAILURE; break; }
        const size_t nb = ld->nb, base = ld->base;
        raw_table *rts = ld->dst;
        scrappie_hip_call *calls = ld->calls;
        if (ld->ticket > 0) {            /* remember what the deferred reads need for their records */
            struct pending *pd = calloc(1, sizeof *pd);
            size_t nd = 0;
            for (size_t i = 0; i < nb; i++) nd += ld->dflag[i];
            pd->ticket = ld->ticket; pd->n = nd; pd->rts = calloc(nd, sizeof *pd->rts); pd->fn = calloc(nd, sizeof *pd->fn);
            for (size_t i = 0, j = 0; i < nb; i++) if (ld->dflag[i]) { pd->rts[j] = rts[i]; pd->fn[j] = files[base + i]; j++; }
            pd->next = pend; pend = pd;
        }
        if (drain_pending(&pend, engs[0], &s, &line, &buflen, 0)) { rc = EXIT_FAILURE; break; }
        for (size_t i = 0; i < nb; i++) {
            char *fn = files[base + i];
            if (ld->dflag[i]) continue;
            if (!calls[i].basecall) {
                fprintf(stderr, "scrappie: No basecall returned for %s\n", fn);     /* scrappie_raw.c:398 */
            } else {
                write_record(&s, &line, &buflen, fn, &rts[i], &calls[i]);
                nbases += calls[i].basecall_length; ncalled++;
            }
            free(rts[i].raw); free(rts[i].uuid);
        }
        scrappie_hip_free_calls(calls, nb);
        pthread_mutex_lock(&P.mu);
        ld->state = ST_EMPTY;
        pthread_cond_broadcast(&P.cv);
        pthread_mutex_unlock(&P.mu);
    }
    if (rc != EXIT_SUCCESS) { pthread_mutex_lock(&P.mu); P.failed = 1; pthread_cond_broadcast(&P.cv); pthread_mutex_unlock(&P.mu); }
    pthread_join(th_load, NULL); pthread_join(th_eng, NULL);
    if (rc != EXIT_SUCCESS) return rc;
    if (drain_pending(&pend, engs[0], &s, &line, &buflen, 1)) return EXIT_FAILURE;
    const double wall = now_s() - wall0;
    if (s.stats) {
        /* read + prepare = the loader thread, engine = the engine thread's basecall calls, write = this thread; the three run side by side */
        fprintf(stderr, "scrappie stats: %zu files, %zu called, %zu samples, %zu bases; prep=%s, %d host threads, batch %d\n", nfile, ncalled, P.nsample, nbases,
                nshare ? "device" : "host", s.threads, s.batch);
        /* engine = first engine call started to last batch delivered (streaming calls return with a launch group in flight: the time spent INSIDE the calls says little) */
        const double span = P.eng_t1 - P.eng_t0;
        fprintf(stderr, "scrappie stats: read %.3f s (%.3e samples/s)  prepare %.3f s (%.3e samples/s)  engine %.3f s (%.3e samples/s)  first batch load %.3f s\n",
                P.read_s, (double)P.nsample / (P.read_s > 0 ? P.read_s : 1e-9), P.prep_s, (double)P.nsample / (P.prep_s > 0 ? P.prep_s : 1e-9), span,
                (double)P.nsample / (span > 0 ? span : 1e-9), P.first_load_s);
        if (nshare) fprintf(stderr, "scrappie stats: prepare = gather %.3f s + host-to-device copy %.3f s + k_p0 %.3f s + waiting for the slot's previous batch\n", 1e-3 * P.prep_ms[0], 1e-3 * P.prep_ms[1], 1e-3 * P.prep_ms[2]);
        fprintf(stderr, "scrappie stats: wall %.3f s = %.3e samples/s, %.1f kbases/s\n", wall, (double)P.nsample / wall, 1e-3 * (double)nbases / wall);
    }
    free(line);
    for (int k = 0; k < NRING; k++) {
        struct loader *ld = &P.ring[k];
        free(ld->dst); free(ld->staged); free(ld->calls); free(ld->dflag);
        for (int d = 0; d < nshare; d++) { struct share *sh = &ld->sh[d]; free(sh->rts); free(sh->calls); free(sh->off); free(sh->len); free(sh->st); free(sh->en); }
    }
    free(P.base); free(P.nb);
    for (size_t i = 0; i < nfile; i++) free(files[i]);
    free(files);
    for (int k = 0; k < nshare; k++) scrappie_hip_prep_destroy(preps[k]);
    for (int k = 0; k < s.ndev; k++) scrappie_hip_engine_destroy(engs[k]);
    if (s.out != stdout) fclose(s.out);
    return EXIT_SUCCESS;
}

/* subcommand dispatch (src/scrappie.c:13, scrappie_subcommands.c:6): only `raw`
 * is part of this build */
int main(int argc, char **argv) {
    if (argc < 2 || 0 == strcmp(argv[1], "help") || 0 == strcmp(argv[1], "--help")) {
        puts("Usage: scrappie <subcommand> [options]\n  raw        Basecall from raw signal (MI355X)\n  version    Print version\n"
             "Other subcommands of the reference (events, squiggle, mappy, seqmappy, event_table)\nare not part of this build.");
        return argc < 2 ? EXIT_FAILURE : EXIT_SUCCESS;
    }
    if (0 == strcmp(argv[1], "version") || 0 == strcmp(argv[1], "--version")) { puts(SCRAPPIE_HIP_VERSION); return EXIT_SUCCESS; }
    if (0 == strcmp(argv[1], "raw")) return main_raw(argc - 1, argv + 1);
    fprintf(stderr, "scrappie: subcommand \"%s\" is not part of this build (only `raw`)\n", argv[1]);
    return EXIT_FAILURE;
}
