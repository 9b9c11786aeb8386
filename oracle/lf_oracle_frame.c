/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see lf_oracle.h).
 * Whole-frame driver: the call sequence of
 *   /root/reference/src/line_detector/src/line_detector_node.py:141-213 (processImage_)
 *   -> ground_projection_node.py:55-65 -> line_sanity_node.py:48-72,
 * plus the descriptor stage (a-9) on the detected segments.
 */
#include "lf_oracle.h"
#include <stdlib.h>
#include <string.h>

int lfo_process_frame(const lfo_config* c, const uint8_t* bgr_in, lfo_frame_out* out, int cap)
{
    const int Hc = lfo_work_rows(c), W = lfo_work_cols(c);
    const int np = Hc * W;
    uint8_t* bgr = (uint8_t*)malloc((size_t)np * 3);
    uint8_t* hsv = (uint8_t*)malloc((size_t)np * 3);
    uint8_t* bw3 = (uint8_t*)malloc((size_t)np * 3);
    uint8_t* bwd = (uint8_t*)malloc((size_t)np);
    uint8_t* edges = (uint8_t*)malloc((size_t)np);
    uint8_t* edge_color = (uint8_t*)malloc((size_t)np);
    double* nrm = (double*)malloc(sizeof(double) * 2 * (size_t)cap);
    float* ctr = (float*)malloc(sizeof(float) * 2 * (size_t)cap);

    lfo_preprocess(c, bgr_in, bgr);                       /* node :163-175 */
    lfo_bgr2hsv(bgr, np, hsv);                            /* lsd.py:138 */
    lfo_canny_bgr(bgr, Hc, W, c->canny_lo, c->canny_hi, edges); /* lsd.py:139 */
    lfo_color_masks(c, hsv, np, bw3);

    int n = 0;
    for (int col = 0; col < 3; ++col) {                   /* node :184-186 white, yellow, red */
        lfo_dilate_ellipse(bw3 + (size_t)col * np, Hc, W, c->dilation_kernel_size, bwd);
        for (int i = 0; i < np; ++i) edge_color[i] = bwd[i] & edges[i];
        int room = cap - n;
        int k = lfo_lsd_detect(c, edge_color, Hc, W, out->lines + 4 * (size_t)n, NULL, room);
        lfo_find_normals(bwd, Hc, W, out->lines + 4 * (size_t)n, k, nrm, ctr);
        for (int i = 0; i < k; ++i) {
            out->normals[2 * (size_t)(n + i)] = (float)nrm[2 * i];
            out->normals[2 * (size_t)(n + i) + 1] = (float)nrm[2 * i + 1];
            out->color[n + i] = (uint8_t)col;
        }
        out->n_color[col] = k;
        n += k;
    }
    out->n = n;
    lfo_normalize_lines(c, out->lines, n, out->pixels_normalized);     /* node :195-205 */
    lfo_ground_project(c, out->pixels_normalized, n, out->ground);     /* ground_projection_node.py:55-65 */
    lfo_line_sanity(c, out->ground, out->color, n, out->keep, NULL, NULL);

    if (out->desc && out->code && n > 0) {                /* a-9 on the corrected, cropped frame */
        uint8_t* gray = (uint8_t*)malloc((size_t)np);
        uint8_t* blur = (uint8_t*)malloc((size_t)np);
        int16_t* dx = (int16_t*)malloc(sizeof(int16_t) * (size_t)np);
        int16_t* dy = (int16_t*)malloc(sizeof(int16_t) * (size_t)np);
        float* ext = (float*)malloc(sizeof(float) * 4 * (size_t)n);
        float* ang = (float*)malloc(sizeof(float) * (size_t)n);
        int32_t* npx = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
        lfo_bgr2gray(bgr, np, gray);
        lfo_gaussian5_u8(gray, Hc, W, blur);
        lfo_sobel3_s16(blur, Hc, W, dx, dy);
        lfo_keylines(out->lines, n, Hc, W, ext, ang, npx);
        lfo_lbd(dx, dy, Hc, W, ext, ang, npx, n, out->desc, out->code);
        free(gray); free(blur); free(dx); free(dy); free(ext); free(ang); free(npx);
    }
    free(bgr); free(hsv); free(bw3); free(bwd); free(edges); free(edge_color); free(nrm); free(ctr);
    return n;
}
