/* sh_fast5.c -- raw signal ingestion for `scrappie raw`.
 *
 * Replaces read_raw() (src/fast5_interface.c:130-217): first read group under
 * /Raw/Reads/, its "Signal" dataset as float, its "read_id" string attribute,
 * and the pA scaling (raw + offset) * range / digitisation from
 * /UniqueGlobalKey/channel_id (fast5_interface.c:109-128).
 *
 * HDF5 is not a build dependency: the dozen entry points needed are resolved
 * with dlopen at run time; where libhdf5 is absent (e.g. on a bare GPU box) the
 * built-in reader of the HDF5 subset fast5 files use (sh_h5mini.c) takes over.
 * Headerless formats are read as well:
 *     *.f32  little-endian float32 samples, already in pA
 *     *.i16  little-endian int16 DAC counts preceded by three float32:
 *            offset, range, digitisation
 */
#define _GNU_SOURCE
#include "scrappie_hip.h"
#include "sh_internal.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NAN_F ((float)NAN)
typedef int64_t hid_t;          /* HDF5 >= 1.10 */
typedef int herr_t;
typedef unsigned long long hsize_t;
typedef long long hssize_t;

static struct {
    void *lib;
    int tried;
    herr_t (*H5open)(void);
    herr_t (*H5Eset_auto2)(hid_t, void *, void *);
    hid_t (*H5Fopen)(const char *, unsigned, hid_t);
    herr_t (*H5Fclose)(hid_t);
    hid_t (*H5Gopen2)(hid_t, const char *, hid_t);
    herr_t (*H5Gclose)(hid_t);
    ssize_t (*H5Lget_name_by_idx)(hid_t, const char *, int, int, hsize_t, char *, size_t, hid_t);
    hid_t (*H5Dopen2)(hid_t, const char *, hid_t);
    herr_t (*H5Dclose)(hid_t);
    hid_t (*H5Dget_space)(hid_t);
    hssize_t (*H5Sget_simple_extent_npoints)(hid_t);
    herr_t (*H5Sclose)(hid_t);
    herr_t (*H5Dread)(hid_t, hid_t, hid_t, hid_t, hid_t, void *);
    hid_t (*H5Aopen)(hid_t, const char *, hid_t);
    herr_t (*H5Aclose)(hid_t);
    herr_t (*H5Aread)(hid_t, hid_t, void *);
    hid_t (*H5Aget_type)(hid_t);
    int (*H5Tis_variable_str)(hid_t);
    size_t (*H5Tget_size)(hid_t);
    herr_t (*H5Tclose)(hid_t);
    herr_t (*H5free_memory)(void *);
    hid_t *native_float;
} h5;

/* The CLI's loader calls scrappie_hip_read_raw from an OpenMP loop: the library is resolved exactly
 * once (pthread_once: every caller returns only after all symbols are in place), and every HDF5 call is
 * made under h5_mu, because distribution builds of libhdf5 are usually not thread-safe.  Scaling,
 * trimming and normalisation of the samples stay outside the lock. */
static pthread_once_t h5_once = PTHREAD_ONCE_INIT;
static pthread_mutex_t h5_mu = PTHREAD_MUTEX_INITIALIZER;

static int h5_resolve(void) {
    const char *cands[] = { getenv("SCRAPPIE_HDF5_LIB"), "libhdf5.so", "libhdf5_serial.so", "libhdf5.so.103",
                            "libhdf5_serial.so.103", "libhdf5.so.200", "libhdf5_serial.so.200", "libhdf5.so.310",
                            "/opt/conda/lib/libhdf5.so", NULL };
    void *lib = NULL;
    for (size_t i = 0; i < sizeof cands / sizeof *cands && !lib; i++)
        if (cands[i]) lib = dlopen(cands[i], RTLD_NOW | RTLD_LOCAL);
    if (!lib) return -1;
#define SYM(n) do { *(void **)&h5.n = dlsym(lib, #n); if (!h5.n) { dlclose(lib); return -1; } } while (0)
    SYM(H5open); SYM(H5Eset_auto2); SYM(H5Fopen); SYM(H5Fclose); SYM(H5Gopen2); SYM(H5Gclose);
    SYM(H5Lget_name_by_idx); SYM(H5Dopen2); SYM(H5Dclose); SYM(H5Dget_space);
    SYM(H5Sget_simple_extent_npoints); SYM(H5Sclose); SYM(H5Dread); SYM(H5Aopen); SYM(H5Aclose);
    SYM(H5Aread); SYM(H5Aget_type); SYM(H5Tis_variable_str); SYM(H5Tget_size); SYM(H5Tclose);
#undef SYM
    *(void **)&h5.H5free_memory = dlsym(lib, "H5free_memory");
    h5.native_float = (hid_t *)dlsym(lib, "H5T_NATIVE_FLOAT_g");
    if (!h5.native_float || h5.H5open() < 0) { dlclose(lib); return -1; }
    h5.H5Eset_auto2(0, NULL, NULL);
    h5.lib = lib;
    return 0;
}

static void h5_load_once(void) { if (h5_resolve() != 0) h5.lib = NULL; h5.tried = 1; }

static int h5_load(void) {
    pthread_once(&h5_once, h5_load_once);
    return h5.lib ? 0 : -1;
}

int scrappie_hip_have_hdf5(void) { return h5_load() == 0; }

static float attr_float(hid_t grp, const char *name) {       /* fast5_interface.c:24-43 */
    float v = NAN_F;
    hid_t a = h5.H5Aopen(grp, name, 0);
    if (a < 0) return v;
    if (h5.H5Aread(a, *h5.native_float, &v) < 0) v = NAN_F;
    h5.H5Aclose(a);
    return v;
}

static char *attr_string(hid_t grp, const char *name) {      /* fast5_interface.c:46-103 */
    char *out = NULL;
    hid_t a = h5.H5Aopen(grp, name, 0);
    if (a < 0) return NULL;
    hid_t t = h5.H5Aget_type(a);
    if (t >= 0) {
        if (h5.H5Tis_variable_str(t) > 0) {
            char *tmp = NULL;
            if (h5.H5Aread(a, t, &tmp) >= 0 && tmp) {
                out = strdup(tmp);
                if (h5.H5free_memory) h5.H5free_memory(tmp); else free(tmp);
            }
        } else {
            const size_t n = h5.H5Tget_size(t);
            out = calloc(n + 1, 1);
            if (out && h5.H5Aread(a, t, out) < 0) { free(out); out = NULL; }
        }
        h5.H5Tclose(t);
    }
    h5.H5Aclose(a);
    return out;
}

/* sh_h5mini.c: the HDF5 subset single-read fast5 files use, without libhdf5 */
raw_table sh_h5mini_read_raw(const char *filename, float scal[3], char *msg, size_t msgcap);

/* which reader: SCRAPPIE_FAST5_READER=own -> the built-in subset reader only, =hdf5 -> libhdf5 only; otherwise the built-in
 * reader first (no global lock: 6-8e8 samples/s from 8-16 loader threads against 3e7 through libhdf5, which serialises every
 * call -- profiles/r5_cli_rate.txt) and libhdf5, when it can be loaded, for the files the subset reader refuses */
static int reader_choice(void) {      /* 0: own, then libhdf5; 1: own only; 2: libhdf5 only */
    const char *e = getenv("SCRAPPIE_FAST5_READER");
    if (e && !strcmp(e, "own")) return 1;
    if (e && !strcmp(e, "hdf5")) return 2;
    return 0;
}
static int use_own_reader(void) { return reader_choice() != 2; }

/* where the samples of a read go: NULL allocator = malloc */
static float *take(scrappie_hip_sample_alloc alloc, void *ctx, size_t n, int *heap) {
    float *p = alloc ? alloc(ctx, n) : NULL;
    *heap = p == NULL;
    return p ? p : malloc(n * sizeof(float));
}

static raw_table read_fast5_own(const char *filename, bool scale_to_pA, int quiet, scrappie_hip_sample_alloc alloc, void *ctx) {
    float scal[3];
    char msg[200] = "";
    raw_table rt = sh_h5mini_read_raw(filename, scal, msg, sizeof msg);
    if (!rt.raw) { if (!quiet) fprintf(stderr, "scrappie: Failed to read %s with the built-in fast5 reader: %s.\n", filename, msg); return rt; }
    float *dst = alloc ? alloc(ctx, rt.n) : NULL;      /* (the counts are turned into samples on their way there) */
    if (scale_to_pA) {                                    /* fast5_interface.c:196-203 */
        const float unit = scal[1] / scal[2];
        float *out = dst ? dst : rt.raw;
        for (size_t i = 0; i < rt.n; i++) out[i] = (rt.raw[i] + scal[0]) * unit;
    } else if (dst) memcpy(dst, rt.raw, rt.n * sizeof(float));
    if (dst) { free(rt.raw); rt.raw = dst; }
    return rt;
}

static raw_table read_fast5(const char *filename, bool scale_to_pA, scrappie_hip_sample_alloc alloc, void *ctx) {
    raw_table rt = { NULL, 0, 0, 0, NULL };
    const int choice = reader_choice();
    if (choice == 1) return read_fast5_own(filename, scale_to_pA, 0, alloc, ctx);
    if (choice == 0) {
        const int have = h5_load() == 0;
        rt = read_fast5_own(filename, scale_to_pA, have, alloc, ctx);       /* (quiet when libhdf5 can still try) */
        if (rt.raw || !have) return rt;
    }
    if (h5_load() != 0) {
        fprintf(stderr, "scrappie: no HDF5 library found (set SCRAPPIE_HDF5_LIB); cannot read %s\n", filename);
        return rt;
    }
    pthread_mutex_lock(&h5_mu);
    hid_t f = h5.H5Fopen(filename, 0 /* H5F_ACC_RDONLY */, 0);
    if (f < 0) { pthread_mutex_unlock(&h5_mu); fprintf(stderr, "scrappie: Failed to open %s for reading.\n", filename); return rt; }
    static const char root[] = "/Raw/Reads/";
    float dig = NAN_F, off = NAN_F, range = NAN_F;
    char *name = NULL, *path = NULL, *uuid = NULL;
    float *buf = NULL;
    hid_t dset = -1, space = -1;
    do {
        ssize_t sz = h5.H5Lget_name_by_idx(f, root, 0, 0, 0, NULL, 0, 0);
        if (sz < 0) { fprintf(stderr, "scrappie: Failed find read name under %s.\n", root); break; }
        name = calloc((size_t)sz + 1, 1);
        path = calloc(sizeof root + (size_t)sz + 8, 1);
        if (!name || !path) break;
        h5.H5Lget_name_by_idx(f, root, 0, 0, 0, name, (size_t)sz + 1, 0);
        sprintf(path, "%s%s", root, name);
        hid_t g = h5.H5Gopen2(f, path, 0);
        if (g < 0) break;
        uuid = attr_string(g, "read_id");
        h5.H5Gclose(g);
        sprintf(path, "%s%s/Signal", root, name);
        dset = h5.H5Dopen2(f, path, 0);
        if (dset < 0) { fprintf(stderr, "scrappie: Failed to open dataset '%s'.\n", path); break; }
        space = h5.H5Dget_space(dset);
        if (space < 0) break;
        const hssize_t n = h5.H5Sget_simple_extent_npoints(space);
        if (n <= 0) break;
        int heap = 1;
        buf = take(alloc, ctx, (size_t)n, &heap);
        if (!buf || h5.H5Dread(dset, *h5.native_float, 0, 0, 0, buf) < 0) { if (heap) free(buf); buf = NULL; break; }
        if (scale_to_pA) {
            hid_t cg = h5.H5Gopen2(f, "/UniqueGlobalKey/channel_id", 0);
            if (cg >= 0) {
                dig = attr_float(cg, "digitisation"); off = attr_float(cg, "offset"); range = attr_float(cg, "range");
                h5.H5Gclose(cg);
            }
        }
        rt = (raw_table){ uuid, (size_t)n, 0, (size_t)n, buf };
        uuid = NULL;
    } while (0);
    free(uuid);
    if (space >= 0) h5.H5Sclose(space);
    if (dset >= 0) h5.H5Dclose(dset);
    free(path); free(name);
    h5.H5Fclose(f);
    pthread_mutex_unlock(&h5_mu);
    if (rt.raw && scale_to_pA) {                          /* fast5_interface.c:196-203 */
        const float unit = range / dig;
        for (size_t i = 0; i < rt.n; i++) rt.raw[i] = (rt.raw[i] + off) * unit;
    }
    return rt;
}

/* offset, range, digitisation of a fast5 file (fast5_interface.c:109-128); 0 on success */
int scrappie_hip_fast5_scaling(const char *filename, float out[3]) {
    if (use_own_reader()) {
        raw_table rt = sh_h5mini_read_raw(filename, out, NULL, 0);
        const int ok = rt.raw != NULL;
        free(rt.raw); free(rt.uuid);
        if (ok || reader_choice() == 1) return ok ? 0 : -1;
        /* a file the built-in reader refuses (superblock 2 / 3, other filters): libhdf5 when it is there, as scrappie_hip_read_raw does */
    }
    if (h5_load() != 0) return -1;
    pthread_mutex_lock(&h5_mu);
    hid_t f = h5.H5Fopen(filename, 0, 0);
    if (f < 0) { pthread_mutex_unlock(&h5_mu); return -1; }
    hid_t cg = h5.H5Gopen2(f, "/UniqueGlobalKey/channel_id", 0);
    int rc = -1;
    if (cg >= 0) {
        out[0] = attr_float(cg, "offset"); out[1] = attr_float(cg, "range"); out[2] = attr_float(cg, "digitisation");
        h5.H5Gclose(cg);
        rc = 0;
    }
    h5.H5Fclose(f);
    pthread_mutex_unlock(&h5_mu);
    return rc;
}

static int has_suffix(const char *s, const char *suf) {
    const size_t n = strlen(s), m = strlen(suf);
    return n >= m && 0 == strcmp(s + n - m, suf);
}

/* headerless signal files by plain POSIX I/O straight into the destination (a loader thread reads ~1e5 of these per second) */
static raw_table read_flat(const char *filename, int is_i16, scrappie_hip_sample_alloc alloc, void *ctx) {
    raw_table rt = { NULL, 0, 0, 0, NULL };
    const int fd = open(filename, O_RDONLY | O_CLOEXEC);
    if (fd < 0) { fprintf(stderr, "scrappie: Failed to open %s for reading.\n", filename); return rt; }
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size < 0) { close(fd); return rt; }
    size_t bytes = (size_t)sb.st_size;
    float hdr[3] = { 0, 1, 1 };
    if (is_i16) {
        if (bytes < 12 || read(fd, hdr, 12) != 12) { close(fd); return rt; }
        bytes -= 12;
    }
    const size_t n = bytes / (is_i16 ? 2 : 4);
    int heap = 1;
    float *buf = n ? take(alloc, ctx, n, &heap) : NULL;
    if (buf) {
        /* int16 counts are read into the upper half of the float buffer and expanded front to back */
        char *dst = is_i16 ? (char *)buf + n * 2 : (char *)buf;
        size_t want = n * (is_i16 ? 2 : 4), got = 0;
        while (got < want) {
            const ssize_t k = read(fd, dst + got, want - got);
            if (k <= 0) break;
            got += (size_t)k;
        }
        if (got == want) {
            if (is_i16) {
                const float unit = hdr[1] / hdr[2];
                for (size_t i = 0; i < n; i++) {
                    int16_t v;                            /* (memcpy: the counts share the float buffer's storage -- no int16 lvalue on float objects under -fstrict-aliasing) */
                    memcpy(&v, dst + 2 * i, sizeof v);
                    buf[i] = ((float)v + hdr[0]) * unit;
                }
            }
            rt = (raw_table){ NULL, n, 0, n, buf };
        } else if (heap) free(buf);
    }
    close(fd);
    return rt;
}

/* read_raw (fast5_interface.c:130): caller frees .raw and .uuid */
raw_table scrappie_hip_read_raw(const char *filename, bool scale_to_pA) {
    return scrappie_hip_read_raw_into(filename, scale_to_pA, NULL, NULL);
}

/* ... with the samples placed where the caller's allocator says (a pinned staging buffer: scrappie_hip_prep_alloc); where it
 * returns NULL they are malloc'd as above */
raw_table scrappie_hip_read_raw_into(const char *filename, bool scale_to_pA, scrappie_hip_sample_alloc alloc, void *ctx) {
    if (!filename) return (raw_table){ NULL, 0, 0, 0, NULL };
    if (has_suffix(filename, ".f32")) return read_flat(filename, 0, alloc, ctx);
    if (has_suffix(filename, ".i16")) return read_flat(filename, 1, alloc, ctx);
    return read_fast5(filename, scale_to_pA, alloc, ctx);
}
