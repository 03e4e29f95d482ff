/* sh_h5mini.c -- a minimal reader of the HDF5 subset single-read fast5 files use, for `scrappie raw` on a box without libhdf5
 * ("else a minimal own reader").  It does what read_raw() asks libhdf5 for
 * (src/fast5_interface.c:130-217, :24-128): the first group under /Raw/Reads (in name order), its "Signal" dataset (16-bit
 * integers, contiguous or chunked with the deflate filter), its "read_id" string attribute (fixed or variable length), and the
 * offset / range / digitisation attributes of /UniqueGlobalKey/channel_id.
 *
 * Format subset (HDF5 File Format Specification 2.0/3.0): superblock versions 0 and 1; version-1 object headers with continuation
 * blocks; old-style groups (symbol-table message -> version-1 B-tree of group nodes + local heap + symbol nodes); dataspace
 * messages v1 / v2; datatype classes fixed-point, floating-point, string, variable-length string; data layout message v3
 * (contiguous, chunked through a version-1 B-tree of raw-data chunks, compact); filter pipeline v1 / v2 with deflate (id 1) and
 * shuffle (id 2); attribute messages v1 - v3; the global heap.  Anything else (superblock v2 / v3, version-2 object headers,
 * link messages, dense attribute storage, other filters) is refused with a message: such files need libhdf5.
 * Deflate-compressed chunks are inflated by sh_inflate.c (round 6: built for MinKNOW's level-1 streams of int16 noise, 1.7x zlib's rate);
 * zlib's uncompress() is resolved with dlopen only for a stream that decoder refuses (none so far). */
#define _GNU_SOURCE
#include "scrappie_hip.h"
#include "sh_internal.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

typedef struct {
    const unsigned char *p;      /* the whole file */
    size_t n;
    uint64_t base;               /* base address (superblock) */
    int so, sl;                  /* size of offsets / lengths */
    char err[160];
} h5m;

#define UNDEF (~(uint64_t)0)
static uint64_t rd(const h5m *f, size_t at, int nbytes) {        /* little-endian, bounds-checked (UNDEF past the end) */
    if (at > f->n || (size_t)nbytes > f->n - at) return UNDEF;
    uint64_t v = 0;
    for (int i = nbytes - 1; i >= 0; i--) v = (v << 8) | f->p[at + (size_t)i];
    return v;
}
static unsigned by(const h5m *f, size_t at) { return at < f->n ? f->p[at] : 0u; }
static int fail(h5m *f, const char *what) { if (!f->err[0]) snprintf(f->err, sizeof f->err, "%s", what); return -1; }
static int inside(const h5m *f, uint64_t at, uint64_t len) { return at != UNDEF && at <= f->n && len <= f->n - at; }

/* ---- object headers (version 1) ------------------------------------------------------------------------------ */
typedef struct { int type; size_t at, size; } h5msg;
/* collect the messages of the object header at `addr` (up to cap); returns their number or -1 */
static int messages(h5m *f, uint64_t addr, h5msg *out, int cap) {
    addr += f->base;
    if (!inside(f, addr, 16)) return fail(f, "object header outside the file");
    if (by(f, addr) != 1) return fail(f, "object header version is not 1 (a file written with a newer format: needs libhdf5)");
    const int total = (int)rd(f, addr + 2, 2);
    size_t blk = addr + 16, blkend = addr + 16 + rd(f, addr + 8, 4);
    struct { size_t at, end; } cont[16];
    int ncont = 0, n = 0, seen = 0;
    for (;;) {
        while (seen < total && blk + 8 <= blkend && inside(f, blk, 8)) {
            const int type = (int)rd(f, blk, 2);
            const size_t size = (size_t)rd(f, blk + 2, 2);
            const size_t data = blk + 8;
            if (!inside(f, data, size)) return fail(f, "header message outside the file");
            seen++;
            if (type == 0x0010) {                       /* continuation */
                if (ncont < 16) { cont[ncont].at = f->base + rd(f, data, f->so); cont[ncont].end = cont[ncont].at + rd(f, data + (size_t)f->so, f->sl); ncont++; }
            } else if (type != 0 && n < cap) { out[n].type = type; out[n].at = data; out[n].size = size; n++; }
            blk = data + ((size + 7) & ~(size_t)7);
        }
        if (seen >= total || ncont == 0) break;
        ncont--;
        blk = cont[ncont].at; blkend = cont[ncont].end;
        if (!inside(f, blk, blkend - blk)) return fail(f, "continuation block outside the file");
    }
    return n;
}

/* ---- old-style groups ---------------------------------------------------------------------------------------------- */
/* the object header address of child `name` of the group whose header is at `gaddr`; name == NULL: the child whose name sorts first
 * (its name copied to first[]).  UNDEF if absent */
static uint64_t group_child(h5m *f, uint64_t gaddr, const char *name, char *first, size_t firstcap) {
    h5msg m[64];
    const int nm = messages(f, gaddr, m, 64);
    if (nm < 0) return UNDEF;
    uint64_t btree = UNDEF, heap = UNDEF;
    for (int i = 0; i < nm; i++) if (m[i].type == 0x0011) { btree = rd(f, m[i].at, f->so); heap = rd(f, m[i].at + (size_t)f->so, f->so); }
    if (btree == UNDEF) { fail(f, "group without a symbol table (new-style group: needs libhdf5)"); return UNDEF; }
    heap += f->base;
    if (!inside(f, heap, 8 + 2 * (size_t)f->sl + (size_t)f->so) || memcmp(f->p + heap, "HEAP", 4)) { fail(f, "bad local heap"); return UNDEF; }
    const uint64_t hsize = rd(f, heap + 8, f->sl), hdata = f->base + rd(f, heap + 8 + 2 * (size_t)f->sl, f->so);
    if (!inside(f, hdata, hsize)) { fail(f, "local heap data outside the file"); return UNDEF; }
    /* walk the B-tree depth first (leftmost first: entries are in name order) */
    uint64_t stack[64];
    int sp = 0;
    stack[sp++] = btree;
    uint64_t best = UNDEF;
    char bestname[256] = "";
    long visits = 0;
    while (sp > 0) {
        const uint64_t node = f->base + stack[--sp];
        if (++visits > 100000) { fail(f, "group B-tree does not end (damaged file)"); return UNDEF; }
        if (!inside(f, node, 8 + 2 * (size_t)f->so)) { fail(f, "B-tree node outside the file"); return UNDEF; }
        if (!memcmp(f->p + node, "TREE", 4)) {
            if (by(f, node + 4) != 0) { fail(f, "group B-tree of the wrong type"); return UNDEF; }
            const int used = (int)rd(f, node + 6, 2);
            size_t at = node + 8 + 2 * (size_t)f->so;
            /* key0 child0 key1 child1 ... : push children in reverse so that the leftmost is visited first */
            uint64_t kids[1024];
            int nk = 0;
            for (int i = 0; i < used && nk < 1024; i++) { at += (size_t)f->sl; kids[nk++] = rd(f, at, f->so); at += (size_t)f->so; }
            for (int i = nk - 1; i >= 0 && sp < 64; i--) stack[sp++] = kids[i];
        } else if (!memcmp(f->p + node, "SNOD", 4)) {
            const int nsym = (int)rd(f, node + 6, 2);
            const size_t esz = 2 * (size_t)f->so + 24;
            for (int i = 0; i < nsym; i++) {
                const size_t e = node + 8 + (size_t)i * esz;
                if (!inside(f, e, esz)) { fail(f, "symbol node outside the file"); return UNDEF; }
                const uint64_t noff = rd(f, e, f->so), oh = rd(f, e + (size_t)f->so, f->so);
                if (noff >= hsize) continue;
                const char *nm_ = (const char *)f->p + hdata + noff;
                const size_t maxlen = (size_t)(hsize - noff);
                if (!memchr(nm_, 0, maxlen)) continue;
                if (name) { if (!strcmp(nm_, name)) return oh; }
                else if (best == UNDEF || strcmp(nm_, bestname) < 0) { best = oh; snprintf(bestname, sizeof bestname, "%s", nm_); }
            }
        } else { fail(f, "neither a B-tree node nor a symbol node"); return UNDEF; }
    }
    if (!name && best != UNDEF && first) snprintf(first, firstcap, "%s", bestname);
    return name ? UNDEF : best;
}
static uint64_t resolve(h5m *f, uint64_t root, const char *path) {      /* "a/b/c" from the root group */
    char buf[256];
    snprintf(buf, sizeof buf, "%s", path);
    uint64_t at = root;
    char *save = NULL;          /* (called from the command line's loader threads: no strtok) */
    for (char *tok = strtok_r(buf, "/", &save); tok && at != UNDEF; tok = strtok_r(NULL, "/", &save)) at = group_child(f, at, tok, NULL, 0);
    return at;
}

/* ---- datatypes, attributes ------------------------------------------------------------------------------------- */
typedef struct { int cls, size, sign, vlen_string; } h5type;
static int parse_type(const h5m *f, size_t at, h5type *t) {
    const unsigned cv = by(f, at);
    t->cls = (int)(cv & 15); t->size = (int)rd(f, at + 4, 4);
    t->sign = (by(f, at + 1) >> 3) & 1;
    t->vlen_string = (t->cls == 9 && (by(f, at + 1) & 15) == 1);
    if (t->cls == 0 || t->cls == 1) { if (by(f, at + 1) & 1) return -1; }       /* big-endian: not in fast5 files */
    return 0;
}
/* the attribute `name` of the object at `addr`: type and the offset / size of its raw data.  0 on success */
static int attribute(h5m *f, uint64_t addr, const char *name, h5type *t, size_t *data, size_t *dsize) {
    h5msg m[96];
    const int nm = messages(f, addr, m, 96);
    if (nm < 0) return -1;
    for (int i = 0; i < nm; i++) {
        if (m[i].type == 0x0015) return fail(f, "attributes in dense storage (needs libhdf5)");
        if (m[i].type != 0x000C) continue;
        const size_t a = m[i].at;
        const int ver = by(f, a);
        const size_t nsz = (size_t)rd(f, a + 2, 2), tsz = (size_t)rd(f, a + 4, 2), ssz = (size_t)rd(f, a + 6, 2);
        size_t at = a + 8 + (ver == 3 ? 1 : 0);
        const size_t pad = (ver == 1) ? 7 : 0;
        if (ver < 1 || ver > 3 || !inside(f, at, nsz)) continue;
        const char *an = (const char *)f->p + at;
        const int match = (nsz > 0 && strnlen(an, nsz) == strlen(name) && !strncmp(an, name, nsz));
        at += (nsz + pad) & ~pad;
        const size_t tat = at;
        at += (tsz + pad) & ~pad;
        at += (ssz + pad) & ~pad;
        if (!match) continue;
        if (at > m[i].at + m[i].size || parse_type(f, tat, t)) return fail(f, "attribute of an unsupported type");
        *data = at; *dsize = m[i].at + m[i].size - at;
        return 0;
    }
    return 1;       /* absent */
}
static float attr_float(h5m *f, uint64_t addr, const char *name) {      /* fast5_interface.c:24-43 (read as a float) */
    h5type t; size_t d = 0, n = 0;
    if (attribute(f, addr, name, &t, &d, &n) != 0) return (float)NAN;
    if (t.cls == 1 && t.size == 8 && n >= 8) { double v; memcpy(&v, f->p + d, 8); return (float)v; }
    if (t.cls == 1 && t.size == 4 && n >= 4) { float v; memcpy(&v, f->p + d, 4); return v; }
    if (t.cls == 0 && t.size <= 8 && n >= (size_t)t.size) {
        const uint64_t u = rd(f, d, t.size);
        if (t.sign) { const int sh = 64 - 8 * t.size; return (float)((int64_t)(u << sh) >> sh); }
        return (float)u;
    }
    return (float)NAN;
}
static char *attr_string(h5m *f, uint64_t addr, const char *name) {     /* fast5_interface.c:46-103 */
    h5type t; size_t d = 0, n = 0;
    if (attribute(f, addr, name, &t, &d, &n) != 0) return NULL;
    if (t.cls == 3 && (size_t)t.size <= n) {
        char *s = calloc((size_t)t.size + 1, 1);
        if (s) memcpy(s, f->p + d, (size_t)t.size);
        return s;
    }
    if (t.vlen_string && n >= 8 + (size_t)f->so) {                      /* length, global heap collection, object index */
        const uint64_t col = f->base + rd(f, d + 4, f->so);
        const unsigned idx = (unsigned)rd(f, d + 4 + (size_t)f->so, 4);
        if (!inside(f, col, 8 + (size_t)f->sl) || memcmp(f->p + col, "GCOL", 4)) return NULL;
        const uint64_t csize = rd(f, col + 8, f->sl);
        size_t at = col + 8 + (size_t)f->sl;
        while (inside(f, at, 8 + (size_t)f->sl) && at < col + csize) {
            const unsigned oi = (unsigned)rd(f, at, 2);
            const uint64_t osz = rd(f, at + 8, f->sl);
            const size_t od = at + 8 + (size_t)f->sl;
            if (oi == 0) break;
            if (oi == idx && inside(f, od, osz)) {
                char *s = calloc((size_t)osz + 1, 1);
                if (s) memcpy(s, f->p + od, (size_t)osz);
                return s;
            }
            at = od + (((size_t)osz + 7) & ~(size_t)7);
        }
    }
    return NULL;
}

/* ---- datasets ---------------------------------------------------------------------------------------------------------- */
static unsigned long n_zlib_fallbacks;
unsigned long sh_h5mini_zlib_fallbacks(void) { return __atomic_load_n(&n_zlib_fallbacks, __ATOMIC_RELAXED); }
static pthread_once_t z_once = PTHREAD_ONCE_INIT;
static int (*z_uncompress)(unsigned char *, unsigned long *, const unsigned char *, unsigned long);
static void z_load(void) {
    const char *c[] = { "libz.so.1", "libz.so", NULL };
    for (int i = 0; c[i] && !z_uncompress; i++) { void *l = dlopen(c[i], RTLD_NOW | RTLD_LOCAL); if (l) *(void **)&z_uncompress = dlsym(l, "uncompress"); }
}

/* the 1-D dataset of 16-bit integers at `addr` as floats; returns the element count or -1 */
/* number of raw-data chunks under the version-1 chunk B-tree at bt (0 for a damaged tree): what bounds an allocation */
static long count_chunks(h5m *f, uint64_t bt, int dim) {
    uint64_t stack[64];
    int sp = 0;
    long visits = 0, leaves = 0;
    if (bt != UNDEF) stack[sp++] = bt;
    while (sp > 0) {
        const uint64_t node = f->base + stack[--sp];
        if (++visits > 1000000) return 0;
        if (!inside(f, node, 8 + 2 * (size_t)f->so) || memcmp(f->p + node, "TREE", 4) || by(f, node + 4) != 1) return 0;
        const int level = by(f, node + 5), used = (int)rd(f, node + 6, 2);
        const size_t ksz = 8 + 8 * (size_t)dim;
        size_t at = node + 8 + 2 * (size_t)f->so;
        if (level == 0) { leaves += used; continue; }
        for (int i = 0; i < used; i++, at += ksz + (size_t)f->so) {
            if (!inside(f, at, ksz + (size_t)f->so)) return 0;
            if (sp < 64) stack[sp++] = rd(f, at + ksz, f->so); else return 0;
        }
    }
    return leaves;
}

static long long dataset_i16(h5m *f, uint64_t addr, float **out) {
    h5msg m[64];
    const int nm = messages(f, addr, m, 64);
    if (nm < 0) return -1;
    long long n = -1;
    h5type t = {0, 0, 0, 0};
    size_t lay = 0;
    int deflate = 0, shuffle = 0, other = 0;
    for (int i = 0; i < nm; i++) {
        const size_t a = m[i].at;
        if (m[i].type == 0x0001) {                       /* dataspace */
            const int ver = by(f, a), rank = by(f, a + 1);
            if (rank != 1) return fail(f, "Signal is not one-dimensional");
            n = (long long)rd(f, a + (ver == 1 ? 8 : 4), f->sl);
        } else if (m[i].type == 0x0003) { if (parse_type(f, a, &t)) return fail(f, "Signal of an unsupported type"); }
        else if (m[i].type == 0x0008) lay = a;
        else if (m[i].type == 0x000B) {                  /* filter pipeline */
            const int ver = by(f, a), nf = by(f, a + 1);
            size_t at = a + (ver == 1 ? 8 : 2);
            for (int k = 0; k < nf; k++) {
                const int id = (int)rd(f, at, 2);
                size_t nlen = 0;
                if (ver == 1 || id >= 256) { nlen = (size_t)rd(f, at + 2, 2); at += 2; }
                const int ncd = (int)rd(f, at + 4, 2);
                at += 6;
                if (ver == 1) nlen = (nlen + 7) & ~(size_t)7;
                at += nlen + 4 * (size_t)ncd;
                if (ver == 1 && (ncd & 1)) at += 4;
                if (id == 1) deflate = 1; else if (id == 2) shuffle = 1; else other = id;
            }
        }
    }
    if (n < 0 || !lay) return fail(f, "Signal has no dataspace or layout");
    if ((uint64_t)n > 600 * (uint64_t)f->n + 4096) return fail(f, "Signal is longer than the file can hold (damaged file)");   /* deflate: at most ~1032 : 1 */
    if (t.cls != 0 || t.size != 2) return fail(f, "Signal is not a 16-bit integer dataset (needs libhdf5)");
    if (other) return fail(f, "Signal uses a filter other than deflate / shuffle (needs libhdf5)");
    if (by(f, lay) != 3) return fail(f, "data layout message is not version 3 (needs libhdf5)");
    const int cls = by(f, lay + 1);
    {   /* the dataspace is not trusted with an allocation: what the layout can actually hold bounds n first */
        uint64_t holds = 0;
        if (cls == 1) { const uint64_t ds = rd(f, lay + 2 + (size_t)f->so, f->sl); holds = ds == UNDEF ? 0 : ds / 2; if (rd(f, lay + 2, f->so) == UNDEF) holds = f->n; }
        else if (cls == 0) holds = rd(f, lay + 2, 2) / 2;
        else if (cls == 2) {
            const int dim = by(f, lay + 2);
            const uint64_t bt = rd(f, lay + 3, f->so), chunk = rd(f, lay + 3 + (size_t)f->so, 4);
            if (dim == 2 && chunk != 0 && chunk != UNDEF) holds = (uint64_t)count_chunks(f, bt, dim) * chunk;
        }
        if ((uint64_t)n > holds) return fail(f, "Signal is longer than its storage can hold (damaged file)");
    }
    unsigned char *raw = calloc((size_t)n ? (size_t)n : 1, 2);
    if (!raw) return fail(f, "out of memory");
    int ok = 0;
    if (cls == 1) {                                      /* contiguous */
        const uint64_t da = rd(f, lay + 2, f->so), ds = rd(f, lay + 2 + (size_t)f->so, f->sl);
        if (da == UNDEF) ok = 1;                         /* never written: zeros */
        else if (inside(f, f->base + da, ds) && ds >= (uint64_t)n * 2) { memcpy(raw, f->p + f->base + da, (size_t)n * 2); ok = 1; }
    } else if (cls == 0) {                               /* compact */
        const size_t ds = (size_t)rd(f, lay + 2, 2);
        if (ds >= (size_t)n * 2 && inside(f, lay + 4, ds)) { memcpy(raw, f->p + lay + 4, (size_t)n * 2); ok = 1; }
    } else if (cls == 2) {                               /* chunked: version-1 B-tree of raw-data chunks */
        const int dim = by(f, lay + 2);
        const uint64_t bt = rd(f, lay + 3, f->so);
        const uint64_t chunk = rd(f, lay + 3 + (size_t)f->so, 4);
        if (dim != 2 || chunk == 0 || chunk == UNDEF) { free(raw); return fail(f, "unexpected chunk shape"); }
        unsigned char *tmp = malloc((size_t)chunk * 2 + 320);     /* (+ 320: room for the inflater's fast loop up to the last byte of a chunk) */
        uint64_t stack[64];
        int sp = 0;
        ok = tmp != NULL;
        if (bt != UNDEF) stack[sp++] = bt;               /* (an empty dataset has no tree) */
        long visits = 0;
        while (ok && sp > 0) {
            const uint64_t node = f->base + stack[--sp];
            if (++visits > 1000000) { ok = 0; break; }   /* a damaged tree that points back at itself */
            if (!inside(f, node, 8 + 2 * (size_t)f->so) || memcmp(f->p + node, "TREE", 4) || by(f, node + 4) != 1) { ok = 0; break; }
            const int level = by(f, node + 5), used = (int)rd(f, node + 6, 2);
            const size_t ksz = 8 + 8 * (size_t)dim;
            size_t at = node + 8 + 2 * (size_t)f->so;
            for (int i = 0; i < used && ok; i++, at += ksz + (size_t)f->so) {
                if (!inside(f, at, ksz + (size_t)f->so)) { ok = 0; break; }
                const uint64_t csz = rd(f, at, 4), mask = rd(f, at + 4, 4), off0 = rd(f, at + 8, 8), child = rd(f, at + ksz, f->so);
                if (level > 0) { if (sp < 64) stack[sp++] = child; else ok = 0; continue; }
                if (!inside(f, f->base + child, csz) || off0 >= (uint64_t)n) { ok = off0 >= (uint64_t)n; continue; }
                const unsigned char *src = f->p + f->base + child;
                unsigned long got = (unsigned long)chunk * 2;
                /* filters are applied shuffle first, deflate second when writing: undo in reverse; a set mask bit = that filter was skipped */
                if (deflate && !(mask & (shuffle ? 2u : 1u))) {
                    /* the built-in inflater (sh_inflate.c); a stream it refuses gets a second opinion from zlib when libz.so.1 can be loaded -- the two agree on
                     * every stream tried (tests/test_host_cpu.py), so that path is a safety net and is counted (sh_h5mini_zlib_fallbacks) */
                    size_t g = 0;
                    if (sh_zlib_inflate(tmp, (size_t)chunk * 2 + 320, &g, src, (size_t)csz) == 0 && g <= (size_t)chunk * 2) got = (unsigned long)g;
                    else {
                        pthread_once(&z_once, z_load);
                        got = (unsigned long)chunk * 2;
                        if (!z_uncompress || z_uncompress(tmp, &got, src, (unsigned long)csz) != 0) { ok = 0; break; }
                        __atomic_fetch_add(&n_zlib_fallbacks, 1, __ATOMIC_RELAXED);
                    }
                } else { got = (unsigned long)(csz < chunk * 2 ? csz : chunk * 2); memcpy(tmp, src, got); }
                const size_t cnt = (size_t)(((uint64_t)n - off0 < chunk) ? (uint64_t)n - off0 : chunk);
                if (got < cnt * 2 && got < (unsigned long)chunk * 2) { ok = 0; break; }
                if (shuffle && !(mask & 1u)) {           /* byte planes of the WHOLE chunk -> samples */
                    const size_t ne = (size_t)got / 2;
                    for (size_t k = 0; k < cnt; k++) { raw[(off0 + k) * 2] = tmp[k]; raw[(off0 + k) * 2 + 1] = tmp[ne + k]; }
                } else memcpy(raw + off0 * 2, tmp, cnt * 2);
            }
        }
        free(tmp);
    }
    if (!ok) { free(raw); return fail(f, "could not read the Signal dataset"); }
    float *buf = malloc(((size_t)n ? (size_t)n : 1) * sizeof(float));
    if (!buf) { free(raw); return fail(f, "out of memory"); }
    for (long long i = 0; i < n; i++) {
        const unsigned v = raw[2 * i] | ((unsigned)raw[2 * i + 1] << 8);
        buf[i] = t.sign ? (float)(int16_t)v : (float)v;
    }
    free(raw);
    *out = buf;
    return n;
}

/* ---- the file ----------------------------------------------------------------------------------------------------------- */
static int open_file(h5m *f, const char *filename, unsigned char **owned, uint64_t *root) {
    memset(f, 0, sizeof *f);
    /* plain POSIX I/O, one read(): a loader thread opens ~2e4 of these per second (stdio's fopen / fseek / ftell / fread cost twice the system calls) */
    const int fd = open(filename, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return fail(f, "cannot open the file");
    struct stat sb_;
    if (fstat(fd, &sb_) != 0 || sb_.st_size <= 0) { close(fd); return fail(f, "cannot read the file"); }
    const size_t sz = (size_t)sb_.st_size;
    unsigned char *p = malloc(sz + 32);                  /* (+ 32: the inflater's 8-byte loads may start up to the last byte of a chunk at the end of the file) */
    size_t have = 0;
    while (p && have < sz) { const ssize_t k = read(fd, p + have, sz - have); if (k <= 0) break; have += (size_t)k; }
    close(fd);
    if (!p || have != sz) { free(p); return fail(f, "cannot read the file"); }
    memset(p + sz, 0, 32);
    *owned = p; f->p = p; f->n = sz;
    size_t sb = 0;
    static const unsigned char sig[8] = { 0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n' };
    for (;; sb = sb ? sb * 2 : 512) {                    /* the superblock sits at 0, 512, 1024, ... */
        if (sb + 8 > f->n) return fail(f, "not an HDF5 file");
        if (!memcmp(p + sb, sig, 8)) break;
    }
    if (sb + 24 + 4 + 6 * 8 > f->n) return fail(f, "truncated superblock");      /* everything read below lies inside the file */
    const int ver = p[sb + 8];
    if (ver > 1) return fail(f, "superblock version 2 or 3 (a file written with the newer format: needs libhdf5)");
    f->so = p[sb + 13]; f->sl = p[sb + 14];
    if ((f->so != 4 && f->so != 8) || (f->sl != 4 && f->sl != 8)) return fail(f, "unexpected offset / length sizes");
    size_t at = sb + 24 + (ver == 1 ? 4 : 0);
    f->base = rd(f, at, f->so);
    at += 4 * (size_t)f->so;                             /* base, free-space info, end of file, driver info */
    *root = rd(f, at + (size_t)f->so, f->so);            /* root group symbol table entry: link name offset, object header address */
    if (f->base == UNDEF || *root == UNDEF) return fail(f, "truncated superblock");
    return 0;
}

/* read_raw() without libhdf5: raw counts as floats (NOT scaled), the read id, and offset / range / digitisation in scal[3].
 * On failure .raw == NULL and msg (if given) says why. */
raw_table sh_h5mini_read_raw(const char *filename, float scal[3], char *msg, size_t msgcap) {
    raw_table rt = { NULL, 0, 0, 0, NULL };
    h5m f;
    unsigned char *owned = NULL;
    uint64_t root = UNDEF;
    float *buf = NULL;
    char *uuid = NULL;
    long long n = -1;
    if (open_file(&f, filename, &owned, &root) == 0) {
        const uint64_t reads = resolve(&f, root, "Raw/Reads");
        char first[256] = "";
        const uint64_t rg = reads != UNDEF ? group_child(&f, reads, NULL, first, sizeof first) : UNDEF;
        if (rg == UNDEF) fail(&f, "no read group under /Raw/Reads/");
        else {
            uuid = attr_string(&f, rg, "read_id");
            const uint64_t sig = group_child(&f, rg, "Signal", NULL, 0);
            if (sig == UNDEF) fail(&f, "no Signal dataset");
            else n = dataset_i16(&f, sig, &buf);
            if (scal) {
                scal[0] = scal[1] = scal[2] = (float)NAN;
                const uint64_t cg = resolve(&f, root, "UniqueGlobalKey/channel_id");
                if (cg != UNDEF) { scal[0] = attr_float(&f, cg, "offset"); scal[1] = attr_float(&f, cg, "range"); scal[2] = attr_float(&f, cg, "digitisation"); }
            }
        }
    }
    if (n > 0 && buf) { rt = (raw_table){ uuid, (size_t)n, 0, (size_t)n, buf }; uuid = NULL; buf = NULL; }
    else if (msg && msgcap) snprintf(msg, msgcap, "%s", f.err[0] ? f.err : "empty Signal dataset");
    free(uuid); free(buf); free(owned);
    return rt;
}
