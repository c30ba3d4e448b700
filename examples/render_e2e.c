/* render_e2e.c — a COMPILED consumer of the C ABI (include/forma_hip.h): no Python, no ctypes.
 *
 *     cc -std=c11 -O2 -Iinclude examples/render_e2e.c -Lforma_amd/csrc -lforma_hip -Wl,-rpath,$PWD/forma_amd/csrc -o build/render_e2e
 *     build/render_e2e OUT_DIR            -> OUT_DIR/<scene>.rgba (64 x 64 x RGBA8, row-major), one line per scene on stdout
 *
 * It is what a reference-side binding does per frame (forma/src/cpu/renderer.rs:75-224 as the Rust shim rust/forma_hip/mod.rs
 * restates it; e2e-tests/tests/test_env.rs:40-59 is the caller it stands in for): build the flat scene tables a
 * `Composition` holds — points + line slots (SegmentBuffer, segment.rs:152-198), the geom table (layer order per slot), the
 * style words (gpu/style_map.rs:140-255 layout) — then forma_hip_create -> set_geometry / set_geoms / set_styles / set_images
 * -> forma_hip_render into caller memory, RGBA, clear {1, 1, 1, 0} (test_env.rs:45-55).
 *
 * The scenes are the polygon-only e2e scenes of the reference (e2e-tests/tests/tests.rs): straight-edged paths need no
 * curve flattening, so the tables can be written down in plain C:
 *     linear_gradient        tests.rs `linear_gradient`: triangle() filled with a 3-stop linear gradient
 *     solid_color__red       tests.rs `solid_color`: square(), opaque red
 *     solid_color__transparent_black   the same square, (0, 0, 0, 0.5)
 *     pixel                  tests.rs `pixel`: one unit square at (PADDING, PADDING)
 *     fill_rules__EvenOdd / fill_rules__NonZero   tests.rs `fill_rules`: a self-intersecting hexagon, (0, 0, 0, 0.8)
 *     covers                 tests.rs `covers` (:414-433): 32 x 32 unit squares at stride 2 + 1/32 in ONE layer
 *     blend_modes__Multiply  tests.rs `blend_modes`: square() with a horizontal rainbow under triangle() with a vertical one
 * tests/test_example_c.py (-m gpu) builds and runs this file and diffs every image with the oracle (bit-exact) and with the
 * reference's CPU golden PNG of the same name (tolerance 8, test_env.rs:278).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "forma_hip.h"

#define W 64
#define H 64
#define PAD 8.0f                      /* test_env.rs:36-38 */
#define MAX_POINTS 8192
#define MAX_WORDS 1024
#define MAX_LAYERS 8

typedef struct scene {
    float x[MAX_POINTS], y[MAX_POINTS];
    uint32_t line_slot[MAX_POINTS];   /* line i joins points i and i + 1; FORMA_NONE: no line (end of a contour) */
    size_t n_points;
    forma_geom_t geoms[MAX_LAYERS];   /* slot -> order */
    size_t n_geoms;
    uint32_t style_off[MAX_LAYERS];   /* order -> offset into words (FORMA_NONE: no such order) */
    size_t n_orders;
    uint32_t words[MAX_WORDS];
    size_t n_words;
} scene;

static uint32_t f32bits(float v) { uint32_t u; memcpy(&u, &v, 4); return u; }

static void scene_init(scene* s) {
    memset(s, 0, sizeof *s);
    for (int i = 0; i < MAX_LAYERS; i++) s->style_off[i] = FORMA_NONE;
}

/* PathBuilder::build closes an open contour with a line back to its start (path.rs:596-615); Path::push_segments_to +
 * SegmentBuffer::push_path (path.rs:677-723, segment.rs:180-198) give every point the id of its geom, the last point of a
 * contour None: the line FROM it does not exist. */
static void add_polygon(scene* s, uint32_t slot, const float (*p)[2], int n) {
    for (int i = 0; i <= n; i++) {
        const int k = i == n ? 0 : i;
        s->x[s->n_points] = p[k][0]; s->y[s->n_points] = p[k][1];
        s->line_slot[s->n_points] = i == n ? FORMA_NONE : slot;
        s->n_points++;
    }
}

static uint32_t add_layer(scene* s, uint32_t order) {            /* Composition::get_mut_or_insert_default(order) */
    const uint32_t slot = (uint32_t)s->n_geoms++;
    s->geoms[slot].order = order; s->geoms[slot].flags = 0;
    if (order + 1 > s->n_orders) s->n_orders = order + 1;
    return slot;
}

/* style words: header (blend 4 | fill 2 | even-odd 1 | clipped 1 | is-clip 1 | .. | stops 16), one reserved word, payload */
static void style_solid(scene* s, uint32_t order, int evenodd, float r, float g, float b, float a) {
    s->style_off[order] = (uint32_t)s->n_words;
    s->words[s->n_words++] = (FORMA_FILL_SOLID << 4) | (evenodd ? 1u << 6 : 0u);
    s->words[s->n_words++] = 0;
    const float c[4] = {r, g, b, a};
    for (int i = 0; i < 4; i++) s->words[s->n_words++] = f32bits(c[i]);
}

static void style_linear(scene* s, uint32_t order, uint32_t blend, float sx, float sy, float ex, float ey, const float (*stops)[5], int n) {
    s->style_off[order] = (uint32_t)s->n_words;
    s->words[s->n_words++] = blend | (FORMA_FILL_LINEAR << 4) | ((uint32_t)n << 16);
    s->words[s->n_words++] = 0;
    const float v[4] = {sx, sy, ex, ey};
    for (int i = 0; i < 4; i++) s->words[s->n_words++] = f32bits(v[i]);
    for (int k = 0; k < n; k++) for (int i = 0; i < 5; i++) s->words[s->n_words++] = f32bits(stops[k][i]);
}

static const float TRIANGLE[3][2] = {{PAD, PAD}, {W - PAD, PAD}, {W - PAD, H - PAD}};                  /* tests.rs:41-56 */
static const float SQUARE[4][2] = {{PAD, PAD}, {PAD, H - PAD}, {W - PAD, H - PAD}, {W - PAD, PAD}};    /* tests.rs:58-78 */

static void rainbow(float (*st)[5]) {                             /* tests.rs:115-183: eleven stops */
    static const float c[11][3] = {{1.00f, 0.00f, 0.00f}, {1.00f, 0.32f, 0.00f}, {0.63f, 0.73f, 0.02f}, {0.08f, 0.72f, 0.07f},
                                   {0.05f, 0.70f, 0.69f}, {0.03f, 0.58f, 0.76f}, {0.01f, 0.21f, 0.85f}, {0.11f, 0.01f, 0.89f},
                                   {0.49f, 0.00f, 0.94f}, {0.96f, 0.00f, 0.69f}, {1.00f, 0.00f, 0.00f}};
    for (int i = 0; i < 11; i++) {
        st[i][0] = c[i][0]; st[i][1] = c[i][1]; st[i][2] = c[i][2]; st[i][3] = 1.0f;
        st[i][4] = (float)i * (1.0f / 10.0f);      /* GradientBuilder::build, styling.rs:112-133: i * (1 / (n - 1)) in f32 */
    }
}

static void build(scene* s, const char* name) {
    scene_init(s);
    if (!strcmp(name, "linear_gradient")) {
        const uint32_t slot = add_layer(s, 1);
        add_polygon(s, slot, TRIANGLE, 3);
        static const float st[3][5] = {{0, 0, 1, 1, 0.0f}, {1, 1, 1, 1, 0.5f}, {1, 0, 0, 1, 1.0f}};
        style_linear(s, 1, 0, PAD, 0.0f, W - PAD, 0.0f, st, 3);
    } else if (!strcmp(name, "solid_color__red") || !strcmp(name, "solid_color__transparent_black")) {
        const uint32_t slot = add_layer(s, 1);
        add_polygon(s, slot, SQUARE, 4);
        if (name[13] == 'r') style_solid(s, 1, 0, 1, 0, 0, 1); else style_solid(s, 1, 0, 0, 0, 0, 0.5f);
    } else if (!strcmp(name, "pixel")) {
        const uint32_t slot = add_layer(s, 1);
        const float p[4][2] = {{PAD, PAD}, {PAD, PAD + 1}, {PAD + 1, PAD + 1}, {PAD + 1, PAD}};
        add_polygon(s, slot, p, 4);
        style_solid(s, 1, 0, 0, 0, 0, 1);
    } else if (!strncmp(name, "fill_rules__", 12)) {
        const uint32_t slot = add_layer(s, 0);
        const float p[6][2] = {{PAD, PAD}, {W / 2 + PAD, H / 2 + PAD}, {W / 2 - PAD, H / 2 + PAD}, {W - PAD, PAD}, {W - PAD, H - PAD}, {PAD, H - PAD}};
        add_polygon(s, slot, p, 6);
        style_solid(s, 0, name[12] == 'E', 0, 0, 0, 0.8f);
    } else if (!strcmp(name, "covers")) {
        const uint32_t slot = add_layer(s, 0);
        const float step = 2.0f + 1.0f / 32.0f;
        for (int xi = 0; xi < 32; xi++)
            for (int yi = 0; yi < 32; yi++) {
                const float x0 = (float)xi * step, y0 = (float)yi * step;
                const float p[4][2] = {{x0, y0}, {x0, y0 + 1}, {x0 + 1, y0 + 1}, {x0 + 1, y0}};
                add_polygon(s, slot, p, 4);
            }
        style_solid(s, 0, 0, 0, 0, 0, 1);
    } else if (!strcmp(name, "blend_modes__Multiply")) {
        float st[11][5];
        rainbow(st);
        const uint32_t s0 = add_layer(s, 0);
        add_polygon(s, s0, SQUARE, 4);
        style_linear(s, 0, 0 /* Over */, 0.0f, PAD, 0.0f, W - PAD, (const float (*)[5])st, 11);
        const uint32_t s1 = add_layer(s, 1);
        add_polygon(s, s1, TRIANGLE, 3);
        style_linear(s, 1, 1 /* Multiply: BlendMode ordinal, styling.rs:390-408 */, PAD, 0.0f, W - PAD, 0.0f, (const float (*)[5])st, 11);
    } else {
        fprintf(stderr, "unknown scene %s\n", name);
        exit(2);
    }
}

#define CHECK(call)                                                                                       \
    do {                                                                                                  \
        const int rc_ = (call);                                                                           \
        if (rc_ != FORMA_OK) {                                                                            \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ctx ? forma_hip_last_error(ctx) : "(no context)"); \
            return 1;                                                                                     \
        }                                                                                                 \
    } while (0)

int main(int argc, char** argv) {
    static const char* const names[] = {"linear_gradient", "solid_color__red", "solid_color__transparent_black", "pixel",
                                        "fill_rules__EvenOdd", "fill_rules__NonZero", "covers", "blend_modes__Multiply"};
    const char* out_dir = argc > 1 ? argv[1] : ".";
    forma_hip_ctx* ctx = NULL;
    CHECK(forma_hip_create(&ctx, 0));                      /* no GPU / no gfx950: FORMA_E_NO_DEVICE — there is no CPU fallback */
    static scene s;
    static uint8_t image[H][W * 4];
    const uint8_t channels[4] = {FORMA_CH_RED, FORMA_CH_GREEN, FORMA_CH_BLUE, FORMA_CH_ALPHA};   /* RGBA */
    const float clear[4] = {1.0f, 1.0f, 1.0f, 0.0f};                                              /* test_env.rs:45-55 */
    for (size_t k = 0; k < sizeof names / sizeof names[0]; k++) {
        build(&s, names[k]);
        CHECK(forma_hip_set_geometry(ctx, s.x, s.y, s.line_slot, s.n_points));
        CHECK(forma_hip_set_geoms(ctx, s.geoms, s.n_geoms));
        CHECK(forma_hip_set_styles(ctx, s.style_off, s.n_orders, s.words, s.n_words, NULL));
        CHECK(forma_hip_set_images(ctx, NULL, 0, NULL, 0));
        forma_timings_t t;
        memset(image, 0, sizeof image);
        /* twice: the first frame of a geometry runs synchronously, the second read-back-free — both must give the same bytes */
        for (int rep = 0; rep < 2; rep++) {
            static uint8_t first[H][W * 4];
            CHECK(forma_hip_render(ctx, &image[0][0], W, H, W * 4, channels, clear, NULL, -1, &t));
            if (rep == 0) memcpy(first, image, sizeof image);
            else if (memcmp(first, image, sizeof image)) { fprintf(stderr, "%s: second frame differs from the first\n", names[k]); return 1; }
        }
        char path[1024];
        snprintf(path, sizeof path, "%s/%s.rgba", out_dir, names[k]);
        FILE* f = fopen(path, "wb");
        if (!f || fwrite(image, 1, sizeof image, f) != sizeof image) { perror(path); return 1; }
        fclose(f);
        printf("%s lines %u segments %u runs %u frame_us %.1f\n", names[k], t.n_lines, t.n_segments, t.n_runs, (double)t.total_us);
    }
    forma_hip_destroy(ctx);
    printf("%s ok\n", forma_hip_version());
    return 0;
}
