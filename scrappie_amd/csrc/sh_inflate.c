/* sh_inflate.c -- inflate (RFC 1951) of zlib streams (RFC 1950) for the built-in fast5 reader (sh_h5mini.c): host C, no device code.
 *
 * Why an own one.  The reference reads a read's Signal through libhdf5, whose deflate filter is zlib's inflate (fast5_interface.c:130-217 ->
 * H5Dread).  With the recurrent layers on a GPU the file side is what bounds `scrappie raw` on fast5 input (9.6e8 samples/s
 * from 16 loader threads against 1.5e9 for the engine), and three quarters of a loader thread's time per file was zlib's uncompress():
 * 49-56 us per chunk of 4000 samples on the build host (142-163 MB/s).  MinKNOW writes Signal as int16 through deflate level 1, and on noisy
 * signal that gives a stream of literals (the low bytes: 9-10 bit codes) mixed ~4 : 3 with matches of length 3 at any distance -- the sequence
 * "literal or match?" is as good as random (a mispredicted branch on 4 symbols in 10).  This decoder is built for that input:
 *   * a 64-bit bit buffer topped up by one unaligned 8-byte load, an 11-bit first-level table (every code of such data in one lookup), one shift
 *     per symbol (an entry says how many bits the code AND its extra bits take; the extra bits are read from the buffer as it was);
 *   * the NEXT symbol's table entry is looked up before the buffer is refilled, so the refill's load is not part of the chain of dependent
 *     lookups that bounds a Huffman decoder (~8 cycles per symbol); up to two literals per refill; matches copied eight bytes at a time.
 *     (A form with no branch between literal and match -- the distance looked up speculatively, its bits dropped under a mask -- was built and
 *     measured slower, 52 against 39 us per chunk: it puts the distance lookup and the refill on every symbol's chain.)
 *   * the Adler-32 of the output checked as zlib checks it.
 * Every stream zlib accepts decodes to the same bytes (stored, fixed and dynamic blocks, any window up to 32 KiB -- the window is the output buffer
 * itself, the whole chunk is decoded at once); every stream zlib rejects is rejected (over-subscribed or incomplete codes, distances beyond the
 * output so far, a wrong checksum; trailing input is ignored as uncompress() ignores it).  tests/test_host_cpu.py drives it against zlib over
 * random and adversarial streams at every compression level and strategy, and over 20 000 corrupted streams.
 *
 * Layout of a table entry (uint32): bits 0-7 = bits to drop (the code -- behind the root for a second-level entry -- AND its extra bits, so
 * one shift consumes a symbol and the extra bits are read from the buffer as it was before); bits 8-15 = kind / extra-bit count; bits 16-31 = value
 *   literal        kind = K_LIT, value = the byte
 *   length         kind = number of extra bits (0 .. 5), value = base length (3 .. 258)
 *   distance       kind = number of extra bits (0 .. 13), value = base distance (1 .. 24577)
 *   end of block   kind = K_EOB
 *   subtable       kind = K_SUB | (index bits of the subtable), value = its first entry; bits 0-7 = the first-level bits
 *   invalid        kind = K_BAD (an index no code maps to: incomplete code sets are refused when the table is built, this is the backstop)
 */
#define _POSIX_C_SOURCE 200809L
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "sh_internal.h"

#define LBITS 11
#define DBITS 8
#define LT_SIZE 2400            /* (288 symbols, 11-bit root, 15-bit codes: 2342 entries at most -- zlib's `enough`; build_table checks) */
#define DT_SIZE 512             /* (32 symbols, 8-bit root: 402) */
#define K_LIT 0x80u
#define K_EOB 0x40u
#define K_SUB 0x20u
#define K_BAD 0x10u
#define FAST_IN 32              /* input bytes the fast loop wants in front of it: up to three 8-byte loads per symbol, each at most 7 bytes further on */
#define FAST_OUT 280            /* output room: a match of 258 copied in words of 8 */

static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static inline uint32_t rev_bits(uint32_t c, int n) {
    uint32_t r = 0;
    for (int i = 0; i < n; i++) { r = (r << 1) | (c & 1u); c >>= 1; }
    return r;
}

/* the entry of symbol s, less its bit count: kind 0 = literal / length code, 1 = distance code, 2 = the code-length code (value = the symbol);
 * K_BAD for symbols that must not occur in a stream */
static inline uint32_t sym_entry(int s, int kind) {
    if (kind == 2) return (uint32_t)s << 16;
    if (kind == 1) return s < 30 ? ((uint32_t)dist_base[s] << 16) | ((uint32_t)dist_extra[s] << 8) : (K_BAD << 8);
    if (s < 256) return ((uint32_t)s << 16) | (K_LIT << 8);
    if (s == 256) return K_EOB << 8;
    return s < 286 ? ((uint32_t)len_base[s - 257] << 16) | ((uint32_t)len_extra[s - 257] << 8) : (K_BAD << 8);
}

/* Canonical Huffman code of n symbols with lengths lens[] (0 = unused) -> decode table with a root of `root` bits, at most cap entries.
 * Returns 0; -1 for an over-subscribed set, or an incomplete one other than the one zlib allows (inftrees.c: a single code of length 1 in a
 * literal / length or distance code; no code at all gives a table of invalid entries). */
static int build_table(const uint8_t *lens, int n, int kind, uint32_t *tab, int root, int cap) {
    int count[16] = {0}, offs[16];
    uint16_t sorted[288];
    for (int i = 0; i < n; i++) count[lens[i]]++;
    int left = 1, maxl = 0;
    for (int l = 1; l <= 15; l++) { left = (left << 1) - count[l]; if (left < 0) return -1; if (count[l]) maxl = l; }
    if (left > 0) for (int i = 0; i < (1 << root); i++) tab[i] = (K_BAD << 8) | 1u;      /* (a complete code writes every index of the root itself) */
    if (count[0] == n) return 0;
    if (left > 0 && (kind == 2 || maxl != 1)) return -1;
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = offs[l] + count[l];
    for (int i = 0; i < n; i++) if (lens[i]) sorted[offs[lens[i]]++] = (uint16_t)i;
    const int nsym = n - count[0];
    int next = 1 << root;                               /* first free entry behind the root table */
    uint32_t code = 0;                                  /* the next code of length l, MSB first */
    int k = 0, l = 1, pin = 0;                          /* pin: codes of length l already placed */
    /* codes that fit the root: every index whose low l bits are the reversed code */
    while (k < nsym) {
        while (pin >= count[l]) { pin = 0; l++; code <<= 1; }
        if (l > root) break;
        uint32_t e = sym_entry(sorted[k], kind);
        e |= (uint32_t)l + ((e >> 8) & 15u);            /* bits to drop: the code and its extra bits */
        for (uint32_t i = rev_bits(code, l); i < (1u << root); i += 1u << l) tab[i] = e;
        k++; code++; pin++;
    }
    /* longer codes: canonical order keeps the codes of one root prefix together; a subtable is as wide as the longest of them needs */
    while (k < nsym) {
        while (pin >= count[l]) { pin = 0; l++; code <<= 1; }
        const uint32_t prefix = code >> (l - root);
        int sl = l, sp = pin, sk = k, maxlen = l;
        uint32_t sc = code;
        for (;;) {                                       /* how far this prefix reaches */
            maxlen = sl;
            sk++; sc++; sp++;
            if (sk >= nsym) break;
            while (sp >= count[sl]) { sp = 0; sl++; sc <<= 1; }
            if ((sc >> (sl - root)) != prefix) break;
        }
        const int sb = maxlen - root;
        if (next + (1 << sb) > cap) return -1;
        tab[rev_bits(prefix, root)] = ((uint32_t)next << 16) | ((K_SUB | (uint32_t)sb) << 8) | (uint32_t)root;
        for (int i = 0; i < (1 << sb); i++) tab[next + i] = (K_BAD << 8) | 1u;
        while (k < sk) {
            while (pin >= count[l]) { pin = 0; l++; code <<= 1; }
            const int extra = l - root;                  /* bits of the code behind the root */
            uint32_t e = sym_entry(sorted[k], kind);
            e |= (uint32_t)extra + ((e >> 8) & 15u);
            for (uint32_t i = rev_bits(code & ((1u << extra) - 1u), extra); i < (1u << sb); i += 1u << extra) tab[next + i] = e;
            k++; code++; pin++;
        }
        next += 1 << sb;
    }
    return 0;
}

#if defined(__SSE2__)
#include <emmintrin.h>
#endif
static uint32_t adler32_of(const unsigned char *p, size_t n) {
    uint32_t a = 1, b = 0;
    while (n) {
        size_t k = n < 5552 ? n : 5552;                  /* the most bytes for which b cannot overflow 32 bits */
        n -= k;
#if defined(__SSE2__)
        /* sixteen bytes at a time: b grows by 16 a + 16 p0 + 15 p1 + ... + p15, a by their sum (psadbw for the sums, pmaddwd for the weights) */
        if (k >= 16) {
            const size_t nv = k / 16;
            const __m128i zero = _mm_setzero_si128();
            const __m128i wlo = _mm_set_epi16(9, 10, 11, 12, 13, 14, 15, 16), whi = _mm_set_epi16(1, 2, 3, 4, 5, 6, 7, 8);
            __m128i s1 = zero, s1_before = zero, s2 = zero;
            for (size_t v = 0; v < nv; v++, p += 16) {
                const __m128i x = _mm_loadu_si128((const __m128i *)p);
                s1_before = _mm_add_epi64(s1_before, s1);
                s1 = _mm_add_epi64(s1, _mm_sad_epu8(x, zero));
                s2 = _mm_add_epi32(s2, _mm_madd_epi16(_mm_unpacklo_epi8(x, zero), wlo));
                s2 = _mm_add_epi32(s2, _mm_madd_epi16(_mm_unpackhi_epi8(x, zero), whi));
            }
            uint64_t t1[2], tb[2];
            uint32_t t2[4];
            _mm_storeu_si128((__m128i *)t1, s1); _mm_storeu_si128((__m128i *)tb, s1_before); _mm_storeu_si128((__m128i *)t2, s2);
            const uint64_t bsum = (uint64_t)b + 16u * ((uint64_t)nv * a + tb[0] + tb[1]) + t2[0] + t2[1] + t2[2] + t2[3];
            a = (uint32_t)((a + t1[0] + t1[1]) % 65521u);
            b = (uint32_t)(bsum % 65521u);
            k -= nv * 16;
        }
#endif
        while (k >= 8) {
            a += p[0]; b += a; a += p[1]; b += a; a += p[2]; b += a; a += p[3]; b += a;
            a += p[4]; b += a; a += p[5]; b += a; a += p[6]; b += a; a += p[7]; b += a;
            p += 8; k -= 8;
        }
        while (k--) { a += *p++; b += a; }
        a %= 65521u; b %= 65521u;
    }
    return (b << 16) | a;
}

static inline uint64_t load64(const unsigned char *p) {
    uint64_t v;
    memcpy(&v, p, 8);
#if defined(__BYTE_ORDER__) && __BYTE_ORDER__ == __ORDER_BIG_ENDIAN__
    v = __builtin_bswap64(v);
#endif
    return v;
}

/* Raw deflate stream src[0 .. srclen) -> dst[0 .. cap); *outlen = bytes produced.  0 on success. */
static int inflate_raw(unsigned char *dst, size_t cap, size_t *outlen, const unsigned char *src, size_t srclen, size_t *consumed) {
    const unsigned char *in = src, *in_end = src + srclen;
    unsigned char *out = dst, *out_end = dst + cap;
    uint64_t bb = 0;
    unsigned bc = 0;
    uint32_t lt[LT_SIZE], dt[DT_SIZE];
    /* byte-wise refill: past the end of the input zeros come in (a stream that needs them fails at the symbol they form or at the checksum) */
#define NEED(nb) do { while (bc < (unsigned)(nb)) { bb |= (uint64_t)(in < in_end ? *in : 0) << bc; in++; bc += 8; } } while (0)
#define DROP(nb) do { bb >>= (nb); bc -= (unsigned)(nb); } while (0)
    int last = 0;
    while (!last) {
        NEED(3);
        last = (int)(bb & 1u);
        const unsigned type = (unsigned)(bb >> 1) & 3u;
        DROP(3);
        if (type == 0) {                                 /* stored */
            DROP(bc & 7u);
            NEED(32);
            const unsigned len = (unsigned)(bb & 0xffffu), nlen = (unsigned)(bb >> 16) & 0xffffu;
            DROP(32);
            if ((len ^ 0xffffu) != nlen) return -1;
            /* (whole bytes are in the bit buffer at this point: give them back) */
            in -= bc >> 3; bb = 0; bc = 0;
            if (in > in_end || (size_t)(in_end - in) < len || (size_t)(out_end - out) < len) return -1;
            memcpy(out, in, len);
            in += len; out += len;
            continue;
        }
        if (type == 3) return -1;
        if (type == 1) {                                 /* fixed code (RFC 1951 3.2.6) */
            uint8_t lens[288 + 32];
            int i = 0;
            for (; i < 144; i++) lens[i] = 8;
            for (; i < 256; i++) lens[i] = 9;
            for (; i < 280; i++) lens[i] = 7;
            for (; i < 288; i++) lens[i] = 8;
            for (i = 0; i < 32; i++) lens[288 + i] = 5;
            if (build_table(lens, 288, 0, lt, LBITS, LT_SIZE) || build_table(lens + 288, 32, 1, dt, DBITS, DT_SIZE)) return -1;
        } else {                                         /* dynamic code */
            NEED(14);
            const int hlit = (int)(bb & 31u) + 257, hdist = (int)((bb >> 5) & 31u) + 1, hclen = (int)((bb >> 10) & 15u) + 4;
            DROP(14);
            if (hlit > 286 || hdist > 30) return -1;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            for (int i = 0; i < hclen; i++) { NEED(3); cl[order[i]] = (uint8_t)(bb & 7u); DROP(3); }
            uint32_t ct[128];
            if (build_table(cl, 19, 2, ct, 7, 128)) return -1;
            uint8_t lens[286 + 30 + 8];
            int n = 0;
            while (n < hlit + hdist) {
                NEED(7 + 7);
                const uint32_t e = ct[bb & 127u];
                const int s = (int)(e >> 16), nb = (int)(e & 0xffu);
                if (((e >> 8) & 0xffu) == K_BAD) return -1;
                DROP(nb);
                if (s < 16) { lens[n++] = (uint8_t)s; continue; }
                int rep, val = 0;
                if (s == 16) { if (n == 0) return -1; val = lens[n - 1]; rep = 3 + (int)(bb & 3u); DROP(2); }
                else if (s == 17) { rep = 3 + (int)(bb & 7u); DROP(3); }
                else { rep = 11 + (int)(bb & 127u); DROP(7); }
                if (n + rep > hlit + hdist) return -1;
                while (rep--) lens[n++] = (uint8_t)val;
            }
            if (lens[256] == 0) return -1;               /* no end-of-block code */
            if (build_table(lens, hlit, 0, lt, LBITS, LT_SIZE) || build_table(lens + hlit, hdist, 1, dt, DBITS, DT_SIZE)) return -1;
        }
        /* ---- the block's symbols ---- */
#define REFILL() do { bb |= load64(in) << bc; in += (63u - bc) >> 3; bc |= 56u; } while (0)
#define EXTRA(saved, e) ((unsigned)(((saved) & (((uint64_t)1 << ((e) & 63u)) - 1u)) >> (((e) & 63u) - (((e) >> 8) & 15u))))
        int done = 0;
        while (!done) {
            if (in_end - in >= (ptrdiff_t)FAST_IN && out_end - out >= (ptrdiff_t)FAST_OUT) {
                /* ---- fast loop: e is always the (first-level) entry of the symbol at the head of the buffer, looked up BEFORE the refill ---- */
                REFILL();
                uint32_t e = lt[bb & ((1u << LBITS) - 1u)];
                do {
                    if (e & (K_SUB << 8)) { DROP(LBITS); e = lt[(e >> 16) + (bb & ((1u << ((e >> 8) & 15u)) - 1u))]; }
                    uint64_t saved = bb;
                    DROP(e & 63u);
                    if (e & (K_LIT << 8)) {
                        *out++ = (unsigned char)(e >> 16);
                        e = lt[bb & ((1u << LBITS) - 1u)];      /* (>= 41 bits are left) */
                        if ((e & ((K_LIT | K_SUB) << 8)) == (K_LIT << 8)) {
                            DROP(e & 63u);
                            *out++ = (unsigned char)(e >> 16);
                            e = lt[bb & ((1u << LBITS) - 1u)];  /* (>= 26) */
                        }
                        REFILL();
                        continue;
                    }
                    const unsigned kind = (e >> 8) & 0xffu;
                    if (kind & (K_EOB | K_BAD)) { if (kind & K_BAD) return -1; done = 1; break; }
                    const unsigned len = (e >> 16) + EXTRA(saved, e);
                    /* (>= 56 - 20 bits are left: a distance needs 15 + 13) */
                    uint32_t d = dt[bb & ((1u << DBITS) - 1u)];
                    if (d & (K_SUB << 8)) { DROP(DBITS); d = dt[(d >> 16) + (bb & ((1u << ((d >> 8) & 15u)) - 1u))]; }
                    if (d & (K_BAD << 8)) return -1;
                    saved = bb;
                    DROP(d & 63u);
                    const size_t dist = (d >> 16) + EXTRA(saved, d);
                    if (bc < LBITS + 4u) REFILL();           /* (rare: a long code with many extra bits on both sides) */
                    e = lt[bb & ((1u << LBITS) - 1u)];
                    REFILL();
                    if (dist > (size_t)(out - dst)) return -1;       /* before the start of the output */
                    const unsigned char *from = out - dist;
                    if (dist >= 8) {
                        unsigned char *to = out;
                        out += len;
                        do { memcpy(to, from, 8); to += 8; from += 8; } while (to < out);
                    } else if (dist == 1) {
                        memset(out, *from, len);
                        out += len;
                    } else {
                        for (unsigned i = 0; i < len; i++) out[i] = from[i];
                        out += len;
                    }
                } while (in_end - in >= (ptrdiff_t)FAST_IN && out_end - out >= (ptrdiff_t)FAST_OUT);
                continue;
            }
            /* ---- careful: one symbol, every byte of input and output checked ---- */
            NEED(15);
            uint32_t e = lt[bb & ((1u << LBITS) - 1u)];
            if (e & (K_SUB << 8)) { DROP(LBITS); e = lt[(e >> 16) + (bb & ((1u << ((e >> 8) & 15u)) - 1u))]; }
            if (e & (K_LIT << 8)) {
                if (out >= out_end) return -1;
                *out++ = (unsigned char)(e >> 16); DROP(e & 63u);
                continue;
            }
            const unsigned kind = (e >> 8) & 0xffu;
            if (kind & (K_EOB | K_BAD)) { DROP(e & 63u); if (kind & K_BAD) return -1; break; }
            NEED(48);
            uint64_t saved = bb;
            DROP(e & 63u);
            unsigned len = (e >> 16) + EXTRA(saved, e);
            uint32_t d = dt[bb & ((1u << DBITS) - 1u)];
            if (d & (K_SUB << 8)) { DROP(DBITS); d = dt[(d >> 16) + (bb & ((1u << ((d >> 8) & 15u)) - 1u))]; }
            if (d & (K_BAD << 8)) return -1;
            saved = bb;
            DROP(d & 63u);
            const size_t dist = (d >> 16) + EXTRA(saved, d);
            if (dist > (size_t)(out - dst) || (size_t)(out_end - out) < len) return -1;
            const unsigned char *from = out - dist;
            while (len--) *out++ = *from++;
        }
    }
    /* whole bytes still in the bit buffer were not part of the stream */
    in -= bc >> 3;
    if (in > in_end) return -1;                          /* the stream ran past its input */
    *outlen = (size_t)(out - dst);
    if (consumed) *consumed = (size_t)(in - src);
    return 0;
#undef NEED
#undef DROP
#undef REFILL
#undef EXTRA
}

/* zlib stream (2-byte header, deflate data, Adler-32 of the output, big endian) -> dst; what uncompress() does.  0 on success. */
int sh_zlib_inflate(unsigned char *dst, size_t cap, size_t *outlen, const unsigned char *src, size_t srclen) {
    if (!dst || !src || !outlen || srclen < 6) return -1;
    const unsigned cmf = src[0], flg = src[1];
    if ((cmf & 15u) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31u != 0 || (flg & 0x20u)) return -1;      /* deflate, window <= 32 KiB, no preset dictionary */
    size_t used = 0;
    if (inflate_raw(dst, cap, outlen, src + 2, srclen - 2, &used)) return -1;
    if (srclen - 2 - used < 4) return -1;
    const unsigned char *t = src + 2 + used;
    const uint32_t want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
    return adler32_of(dst, *outlen) == want ? 0 : -1;
}
