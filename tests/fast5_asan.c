/* tests/fast5_asan.c -- TEST INFRASTRUCTURE: sh_h5mini_read_raw (the built-in fast5 reader + sh_inflate.c) over the files named on the command line, built with
 * -fsanitize=address,undefined by tests/test_cli.py: damaged and truncated files must neither crash nor touch memory that is not theirs. */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdbool.h>
#include "scrappie_hip.h"
raw_table sh_h5mini_read_raw(const char *filename, float scal[3], char *msg, size_t msgcap);
int main(int argc, char **argv) { long ok = 0, n = 0; for (int i = 1; i < argc; i++) { float sc[3]; char msg[200]; raw_table rt = sh_h5mini_read_raw(argv[i], sc, msg, sizeof msg); if (rt.raw) ok++; n++; free(rt.raw); free(rt.uuid); } printf("%ld files, %ld read\n", n, ok); return 0; }
