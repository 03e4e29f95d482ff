/* sh_internal.h -- symbols shared between the host C (sh_host.c) and the HIP
 * translation unit (scrappie_hip.hip); not part of the public ABI. */
#ifndef SH_INTERNAL_H
#define SH_INTERNAL_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
float sh_medianf(const float *x, size_t n, float *scratch);
float sh_madf(const float *x, size_t n, const float *med, float *scratch);
int sh_kmerlength(int nstate);
/* homopolymer correction from the 5-row side buffer [nblock][5] = {A,C,G,T homopolymer k-mer, stay} */
int sh_homopolymer_side(const float *side, int *path, int nblock, int nstate);
/* sh_inflate.c: what zlib's uncompress() does (zlib stream -> dst[0 .. cap), *outlen bytes; 0 on success), built for streams of literals */
int sh_zlib_inflate(unsigned char *dst, size_t cap, size_t *outlen, const unsigned char *src, size_t srclen);
unsigned long sh_h5mini_zlib_fallbacks(void);      /* chunks the built-in inflater refused and zlib decoded (expected: 0) */
#ifdef __cplusplus
}
#endif
#endif
