/* A plain-C caller of the C ABI (include/fav.h): what a non-Python host links against.
 *
 *   gcc -std=c99 -Iinclude examples/classify_host.c -o classify_host \
 *       -Lfailure_aware_vision_amd/lib -lfav_hip -Wl,-rpath,$PWD/failure_aware_vision_amd/lib
 *   ./classify_host <arch 0..3> <checkpoint.favw> <frames.u8> <n> <H> <W>
 *
 * Reads n uint8 HxWx3 frames, classifies them with fav_classify_host and prints one
 * "label confidence fail score" line per frame.  Exit code = fav_status (0 = OK). */
#include <stdio.h>
#include <stdlib.h>
#include "fav.h"

static void* slurp(const char* path, size_t* size) {
    FILE* f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    void* p = malloc((size_t)n);
    if (p && fread(p, 1, (size_t)n, f) != (size_t)n) { free(p); p = NULL; }
    fclose(f);
    *size = (size_t)n;
    return p;
}

int main(int argc, char** argv) {
    if (argc != 7) { fprintf(stderr, "usage: %s arch checkpoint frames n H W (ABI version %d)\n", argv[0], (int)fav_abi_version()); return 64; }
    const int arch = atoi(argv[1]), n = atoi(argv[4]), H = atoi(argv[5]), W = atoi(argv[6]);
    fav_config cfg;
    fav_default_config(&cfg, arch);
    cfg.in_h = H; cfg.in_w = W; cfg.max_batch = n;
    fav_handle* h = NULL;
    fav_status st = fav_create(&cfg, &h);
    if (st != FAV_OK) { fprintf(stderr, "fav_create: %s\n", fav_last_error(NULL)); return (int)st; }
    size_t bsize = 0, fsize = 0;
    void* blob = slurp(argv[2], &bsize);
    void* frames = slurp(argv[3], &fsize);
    if (!blob || !frames || fsize < (size_t)n * H * W * 3) { fprintf(stderr, "cannot read inputs\n"); fav_destroy(h); return 65; }
    st = fav_load_weights(h, blob, bsize);
    if (st != FAV_OK) { fprintf(stderr, "fav_load_weights: %s\n", fav_last_error(h)); fav_destroy(h); return (int)st; }
    int32_t* labels = (int32_t*)malloc(sizeof(int32_t) * n);
    float* conf = (float*)malloc(sizeof(float) * n);
    float* score = (float*)malloc(sizeof(float) * n);
    uint8_t* fail = (uint8_t*)malloc((size_t)n);
    st = fav_classify_host(h, frames, n, FAV_LAYOUT_NHWC_U8, 0, labels, conf, fail, score);
    if (st != FAV_OK) { fprintf(stderr, "fav_classify_host: %s\n", fav_last_error(h)); fav_destroy(h); return (int)st; }
    for (int i = 0; i < n; ++i) printf("%d %.9g %d %.9g\n", (int)labels[i], conf[i], (int)fail[i], score[i]);
    fav_destroy(h);
    free(labels); free(conf); free(score); free(fail); free(blob); free(frames);
    return 0;
}
