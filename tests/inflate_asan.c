/* tests/inflate_asan.c -- TEST INFRASTRUCTURE: runs sh_inflate.c over a corpus of zlib streams (records of [u32 stream bytes][u32 output room][stream]) with heap buffers of
 * EXACTLY those sizes, so that AddressSanitizer sees any read past the input or write past the output (tests/test_host_cpu.py builds it with -fsanitize=address,undefined). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
int sh_zlib_inflate(unsigned char *dst, size_t cap, size_t *outlen, const unsigned char *src, size_t srclen);
/* reads a corpus file: records of [u32 zlen][u32 cap][z bytes]; runs inflate with EXACT-size heap buffers so that ASAN sees any over-read / over-write */
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); if (!f) return 2;
    uint32_t zl, cap; long n = 0, ok = 0;
    while (fread(&zl, 4, 1, f) == 1 && fread(&cap, 4, 1, f) == 1) {
        unsigned char *z = malloc(zl ? zl : 1); if (fread(z, 1, zl, f) != zl) return 3;
        unsigned char *out = malloc(cap ? cap : 1); size_t got = 0;
        if (sh_zlib_inflate(out, cap, &got, z, zl) == 0) ok++;
        free(z); free(out); n++;
    }
    printf("%ld streams, %ld accepted\n", n, ok); return 0;
}
