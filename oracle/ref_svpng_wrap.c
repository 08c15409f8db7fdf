/* ORACLE/_ref recipe: exposes the reference's own svpng() (path_tracing_1chap/svpng.inc:77-108), compiled from the
 * reference source where it lies, so tests can check rt_write_png byte for byte.  No reference text is copied here. */
#include <stdio.h>
#include "svpng.inc"
int ref_svpng_write(const char* path, unsigned w, unsigned h, const unsigned char* img, int alpha) {
    FILE* fp = fopen(path, "wb");
    if (!fp) return -1;
    svpng(fp, w, h, img, alpha);
    fclose(fp);
    return 0;
}
