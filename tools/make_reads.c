/* make_reads.c -- N synthetic raw reads as files, for measuring `scrappie raw` end to end (tools/cli_rate.sh).
 *   make_reads f32   DIR N NSAMPLE [FIRST]   (reads FIRST .. N - 1: several processes can fill one directory)
 *   make_reads f32   DIR N NSAMPLE     little-endian float32 pA samples (read by sh_fast5.c as *.f32)
 *   make_reads fast5 DIR N NSAMPLE     fast5 as MinKNOW writes them: /Raw/Reads/Read_<k>/Signal int16, chunked + deflate,
 *                                      read_id attribute, /UniqueGlobalKey/channel_id scaling attributes (needs libhdf5:
 *                                      gcc -DWITH_HDF5 -I/opt/conda/include ... -L/opt/conda/lib -lhdf5)
 * The signal: piecewise-constant current levels (dwell ~9 samples) in 60..120 pA with noise, a quiet stretch at the start --
 * the shape trim_raw_by_mad and the normalisation are made for.  Deterministic per read index. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef WITH_HDF5
#include <hdf5.h>
#endif

static uint64_t rng_s;
static inline uint32_t rnd(void) { rng_s = rng_s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng_s >> 33); }
static inline float unif(void) { return (float)rnd() / 2147483648.0f; }

static void make_signal(int16_t *dac, size_t n, uint64_t seed) {
    rng_s = seed * 0x9e3779b97f4a7c15ull + 12345;
    float level = 90.0f;
    size_t left = 0;
    for (size_t i = 0; i < n; i++) {
        if (left == 0) { level = 60.0f + 60.0f * unif(); left = 3 + rnd() % 13; }
        left--;
        const float noise = (unif() + unif() + unif() - 1.5f) * (i < 150 ? 0.3f : 3.0f);
        const float pa = (i < 150 ? 95.0f : level) + noise;
        dac[i] = (int16_t)lrintf(pa * 8192.0f / 1373.41f - 16.0f);       /* digitisation 8192, range 1373.41, offset 16 */
    }
}

int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: make_reads f32|fast5 DIR N NSAMPLE [FIRST]\n"); return 2; }
    const int fast5 = 0 == strcmp(argv[1], "fast5");
    const char *dir = argv[2];
    const size_t N = (size_t)atol(argv[3]), ns = (size_t)atol(argv[4]);
    const size_t first = argc > 5 ? (size_t)atol(argv[5]) : 0;      /* reads FIRST .. N - 1 (a read depends on its index alone: several processes can share a directory) */
    int16_t *dac = malloc(ns * sizeof *dac);
    float *pa = malloc(ns * sizeof *pa);
    char path[4096];
    for (size_t k = first; k < N; k++) {
        make_signal(dac, ns, k);
        if (!fast5) {
            for (size_t i = 0; i < ns; i++) pa[i] = ((float)dac[i] + 16.0f) * (1373.41f / 8192.0f);
            snprintf(path, sizeof path, "%s/read_%07zu.f32", dir, k);
            FILE *fh = fopen(path, "wb");
            if (!fh || fwrite(pa, sizeof *pa, ns, fh) != ns) { perror(path); return 1; }
            fclose(fh);
        } else {
#ifdef WITH_HDF5
            snprintf(path, sizeof path, "%s/read_%07zu.fast5", dir, k);
            hid_t f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
            hid_t g1 = H5Gcreate2(f, "/UniqueGlobalKey", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
            hid_t g2 = H5Gcreate2(f, "/UniqueGlobalKey/channel_id", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
            hid_t sc = H5Screate(H5S_SCALAR);
            const double vals[3] = {8192.0, 16.0, 1373.41};
            const char *names[3] = {"digitisation", "offset", "range"};
            for (int a = 0; a < 3; a++) { hid_t at = H5Acreate2(g2, names[a], H5T_NATIVE_DOUBLE, sc, H5P_DEFAULT, H5P_DEFAULT); H5Awrite(at, H5T_NATIVE_DOUBLE, &vals[a]); H5Aclose(at); }
            hid_t g3 = H5Gcreate2(f, "/Raw", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
            hid_t g4 = H5Gcreate2(f, "/Raw/Reads", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
            char gn[64]; snprintf(gn, sizeof gn, "/Raw/Reads/Read_%zu", k);
            hid_t g5 = H5Gcreate2(f, gn, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
            char id[40]; snprintf(id, sizeof id, "%08zx-0000-4000-8000-%012zx", k, (k * 2654435761u) & 0xffffffffffffull);
            hid_t st = H5Tcopy(H5T_C_S1); H5Tset_size(st, 37);
            hid_t at = H5Acreate2(g5, "read_id", st, sc, H5P_DEFAULT, H5P_DEFAULT); H5Awrite(at, st, id); H5Aclose(at); H5Tclose(st);
            hsize_t dims[1] = {ns}, chunk[1] = {ns};
            hid_t sp = H5Screate_simple(1, dims, NULL);
            hid_t pl = H5Pcreate(H5P_DATASET_CREATE); H5Pset_chunk(pl, 1, chunk); H5Pset_deflate(pl, 1);
            hid_t ds = H5Dcreate2(g5, "Signal", H5T_STD_I16LE, sp, H5P_DEFAULT, pl, H5P_DEFAULT);
            H5Dwrite(ds, H5T_NATIVE_SHORT, H5S_ALL, H5S_ALL, H5P_DEFAULT, dac);
            H5Dclose(ds); H5Pclose(pl); H5Sclose(sp); H5Sclose(sc);
            H5Gclose(g5); H5Gclose(g4); H5Gclose(g3); H5Gclose(g2); H5Gclose(g1); H5Fclose(f);
#else
            fprintf(stderr, "built without -DWITH_HDF5\n"); return 2;
#endif
        }
    }
    free(dac); free(pa);
    return 0;
}
