/* A plain-C caller of libd2r.so: include/d2r.h is the whole interface (C99, plain pointers and sizes — what a cgo / JNI / ctypes binding
 * sees).  This program only touches entry points that need no GPU, so that tests/test_abi.py can build and run it anywhere:
 *   - the ABI version the header and the library agree on,
 *   - the error convention (negative code + d2r_last_error text),
 *   - the host-side file writers a render-and-score pass leaves behind: pose_scores.txt in np.savetxt's format
 *     (reference dream2real.py:356-358) and cb_render/cb_rgb_%04d.png (combined_rendering.py:157-159), read back.
 * Build:  gcc -std=c99 -Iinclude examples/c_caller.c -Ldream2real_amd -ld2r -Wl,-rpath,$PWD/dream2real_amd -o c_caller
 * Run:    ./c_caller <scratch dir>        (prints "ok" and exits 0)
 * The GPU calls (d2r_ctx_create ... d2r_render_score_host) follow the same convention; INTEGRATION.md section 2 lists them in call order. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "d2r.h"

#define CHECK(cond, what)                                                         \
    do {                                                                          \
        if (!(cond)) {                                                            \
            fprintf(stderr, "FAILED: %s (%s)\n", what, d2r_last_error(NULL));     \
            return 1;                                                             \
        }                                                                         \
    } while (0)

int main(int argc, char **argv)
{
    const char *dir = argc > 1 ? argv[1] : ".";
    char path[1024];
    uint8_t frames[2][4][6][3], back[2][4][6][3];
    double scores[3][2] = {{0.25, -1.5}, {1e-300, 3.0}, {7.0, 0.1}};
    uint32_t w = 0, h = 0;
    int64_t v = 0;
    size_t i;

    CHECK(d2r_abi_version() == D2R_ABI_VERSION, "header and library disagree on the ABI version");
    /* errors: a negative code, and the message is kept for the caller */
    CHECK(d2r_ctx_set_option(NULL, "chunk", 1) < 0 && strlen(d2r_last_error(NULL)) > 0, "null context must be refused with a message");
    CHECK(d2r_ctx_get_option(NULL, "chunk", &v) < 0, "null context must be refused");

    for (i = 0; i < sizeof frames; i++) ((uint8_t *)frames)[i] = (uint8_t)(i * 37u + 11u);
    CHECK(d2r_png_write_batch(&frames[0][0][0][0], 2, 6, 4, dir, 5, 1, -1) == D2R_OK, "d2r_png_write_batch");
    snprintf(path, sizeof path, "%s/cb_rgb_0006.png", dir);
    CHECK(d2r_png_size(path, &w, &h) == D2R_OK && w == 6 && h == 4, "d2r_png_size");
    CHECK(d2r_png_read_batch(dir, NULL, 5, 2, 6, 4, &back[0][0][0][0], 1) == D2R_OK, "d2r_png_read_batch");
    CHECK(memcmp(frames, back, sizeof frames) == 0, "PNG round trip");
    CHECK(d2r_png_read_batch(dir, NULL, 50, 1, 6, 4, &back[0][0][0][0], 1) < 0 && strstr(d2r_last_error(NULL), "cb_rgb_0050") != NULL,
          "a missing file must be named in the error");

    snprintf(path, sizeof path, "%s/pose_scores.txt", dir);
    CHECK(d2r_savetxt(path, &scores[0][0], 3, 2, 1) == D2R_OK, "d2r_savetxt");
    {
        FILE *f = fopen(path, "r");
        char line[128];
        CHECK(f != NULL && fgets(line, sizeof line, f) != NULL, "pose_scores.txt unreadable");
        fclose(f);
        CHECK(strcmp(line, "2.500000000000000000e-01 -1.500000000000000000e+00\n") == 0, "np.savetxt format");
    }
    puts("ok");
    return 0;
}
