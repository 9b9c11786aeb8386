/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see lf_oracle.h).
 * Per-segment stages a-5 .. a-8.
 */
#include "lf_oracle.h"
#include "lf_detmath.h"
#include <math.h>
#include <string.h>

/* ------------------------------------------------------------------ a-5
 * line_detector_lsd.py:74-125  _findNormal + _checkBounds + _correctPixelOrdering.
 * numpy semantics: lines are float32, so length/dx/dy/centers and the sample
 * coordinates are float32; astype('int') truncates toward zero; normals become
 * float64 holding float32 values; the ordering test is evaluated in float64.
 * PINNED by tests/golden/find_normal.npz (the reference's own code run here).
 */
static int check_bounds(int v, int bound) { if (v < 0) v = 0; if (v >= bound) v = bound - 1; return v; }

void lfo_find_normals(const uint8_t* bw, int rows, int cols, float* lines, int n,
                      double* normals, float* centers)
{
    for (int i = 0; i < n; ++i) {
        float x1 = lines[4 * i], y1 = lines[4 * i + 1], x2 = lines[4 * i + 2], y2 = lines[4 * i + 3];
        float ex = x1 - x2, ey = y1 - y2;
        float len = sqrtf(ex * ex + ey * ey);
        float dx = (y2 - y1) / len;
        float dy = (x1 - x2) / len;
        float cx = (x1 + x2) / 2, cy = (y1 + y2) / 2;
        int x3 = (int)(cx - 3.f * dx), y3 = (int)(cy - 3.f * dy);
        int x4 = (int)(cx + 3.f * dx), y4 = (int)(cy + 3.f * dy);
        x3 = check_bounds(x3, cols); y3 = check_bounds(y3, rows);
        x4 = check_bounds(x4, cols); y4 = check_bounds(y4, rows);
        int sign = (bw[(size_t)y3 * cols + x3] > 0 && bw[(size_t)y4 * cols + x4] == 0) ? 1 : -1;
        double nx = (double)dx * sign, ny = (double)dy * sign;
        normals[2 * i] = nx; normals[2 * i + 1] = ny;
        centers[2 * i] = cx; centers[2 * i + 1] = cy;
        /* _correctPixelOrdering: ((x2-x1)*ny - (y2-y1)*nx) > 0 -> swap endpoints */
        double flag = (double)(x2 - x1) * ny - (double)(y2 - y1) * nx;
        if (flag > 0) {
            lines[4 * i] = x2; lines[4 * i + 1] = y2; lines[4 * i + 2] = x1; lines[4 * i + 3] = y1;
        }
    }
}

/* ------------------------------------------------------------------ a-6
 * line_detector_node.py:195-205: (float64(lines) + [0,cut,0,cut]) * [1/W,1/H,1/W,1/H]
 * with W,H = img_size; Segment.msg stores float32.
 */
void lfo_normalize_lines(const lfo_config* c, const float* lines, int n, float* out)
{
    const double rx = 1.0 / (double)c->img_cols, ry = 1.0 / (double)c->img_rows;
    const double cut = (double)c->top_cutoff;
    for (int i = 0; i < n; ++i) {
        out[4 * i + 0] = (float)(((double)lines[4 * i + 0] + 0.0) * rx);
        out[4 * i + 1] = (float)(((double)lines[4 * i + 1] + cut) * ry);
        out[4 * i + 2] = (float)(((double)lines[4 * i + 2] + 0.0) * rx);
        out[4 * i + 3] = (float)(((double)lines[4 * i + 3] + cut) * ry);
    }
}

/* ------------------------------------------------------------------ a-7
 * GroundProjection.py:38-48 vector2pixel (incl. the v > ch-1 -> 0 quirk),
 * :64-78 pixel2ground; rectifyPoint is always applied (the node sets
 * rectified_input_ but the class reads rectified_input,
 * ground_projection_node.py:34 vs GroundProjection.py:21,66).
 * rectifyPoint = cv2.undistortPoints(pt, K, D, R=R, P=P): normalise by K, five
 * fixed-point iterations of the plumb-bob inverse, apply P[:, :3]*R
 * (OpenCV 3.x cvUndistortPoints; third party, PARITY UNPINNED).
 */
static void ground_point(const lfo_config* c, double vx, double vy, double* gx, double* gy)
{
    const double cw = (double)c->cam_w, ch = (double)c->cam_h;
    double u = cw * vx, v = ch * vy;
    if (u < 0) u = 0;
    if (u > cw - 1) u = cw - 1;
    if (v < 0) v = 0;
    if (v > ch - 1) v = 0;
    const double fx = c->K[0], fy = c->K[4], cx = c->K[2], cy = c->K[5];
    const double ifx = 1. / fx, ify = 1. / fy;
    const double* k = c->D;
    double RR[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int t = 0; t < 3; ++t) s += c->P[4 * i + t] * c->R[3 * t + j];
            RR[i][j] = s;
        }
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    double x0 = x, y0 = y;
    for (int j = 0; j < 5; ++j) {
        double r2 = x * x + y * y;
        double icdist = 1 / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
        double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2];
    double yy = RR[1][0] * x + RR[1][1] * y + RR[1][2];
    double ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
    double ur = xx * ww, vr = yy * ww;
    const double* H = c->H;
    double g0 = H[0] * ur + H[1] * vr + H[2] * 1.0;
    double g1 = H[3] * ur + H[4] * vr + H[5] * 1.0;
    double g2 = H[6] * ur + H[7] * vr + H[8] * 1.0;
    *gx = g0 / g2;
    *gy = g1 / g2;
}

void lfo_ground_project(const lfo_config* c, const float* pn, int n, double* pts)
{
    for (int i = 0; i < n; ++i) {
        ground_point(c, (double)pn[4 * i + 0], (double)pn[4 * i + 1], &pts[4 * i + 0], &pts[4 * i + 1]);
        ground_point(c, (double)pn[4 * i + 2], (double)pn[4 * i + 3], &pts[4 * i + 2], &pts[4 * i + 3]);
    }
}

/* ------------------------------------------------------------------ a-8
 * line_sanity_node.py:48-72 processSegmentList, :75-117 fancyFilters.
 * PINNED by tests/golden/line_sanity.npz (the reference's own code run here).
 */
void lfo_line_sanity(const lfo_config* c, const double* pts, const uint8_t* color, int n,
                     uint8_t* keep, double* dphil, int32_t* state_out)
{
    for (int i = 0; i < n; ++i) {
        const double p1x = pts[4 * i], p1y = pts[4 * i + 1], p2x = pts[4 * i + 2], p2y = pts[4 * i + 3];
        int state = 0;
        double ex = p2x - p1x, ey = p2y - p1y;
        double nrm = sqrt(ex * ex + ey * ey);
        double tx = ex / nrm, ty = ey / nrm;
        double nx = -ty, ny = tx;
        double d1 = nx * p1x + ny * p1y;
        double d2 = nx * p2x + ny * p2y;
        double l1 = tx * p1x + ty * p1y;
        double l2 = tx * p2x + ty * p2y;
        if (l1 < 0) l1 = -l1;
        if (l2 < 0) l2 = -l2;
        double l_i = (l1 + l2) / 2;
        double d_i = (d1 + d2) / 2;
        double phi_i = lfo_asin(ty);
        if (color[i] == LFO_WHITE) {
            if (p1x > p2x) { d_i = d_i - c->linewidth_white; state = 1; }
            else { d_i = -d_i; phi_i = -phi_i; state = 2; }
            d_i = d_i - c->lanewidth / 2;
        } else if (color[i] == LFO_YELLOW) {
            if (p2x > p1x) { d_i = d_i - c->linewidth_yellow; phi_i = -phi_i; state = 3; }
            else { d_i = -d_i; state = 4; }
            d_i = c->lanewidth / 2 - d_i;
        }
        int k = 1;
        if (p1x < 0 || p2x < 0) k = 0;
        else if (color[i] != LFO_WHITE && color[i] != LFO_YELLOW) k = 0;
        else if (state == 0) k = 0;
        else if (d_i > c->d_max || d_i < c->d_min || phi_i < c->phi_min || phi_i > c->phi_max) k = 0;
        keep[i] = (uint8_t)k;
        if (dphil) { dphil[3 * i] = d_i; dphil[3 * i + 1] = phi_i; dphil[3 * i + 2] = l_i; }
        if (state_out) state_out[i] = state;
    }
}
