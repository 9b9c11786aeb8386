/*
 * CPU ORACLE -- test infrastructure only (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 * The product never links or calls this file.
 *
 * Host ingest: what the reference obtains from
 *     cv2.imdecode(np.fromstring(data, np.uint8), cv2.IMREAD_COLOR)
 * (ref: src/duckietown/include/duckietown_utils/jpg.py:21-31, called from
 * src/line_detector/src/line_detector_node.py:153-158): a BGR u8 image, or failure.
 *
 * The decoder behind cv2.imdecode is libjpeg-turbo with its default settings (third-party, not
 * vendored by the reference, version unpinned; ROS Kinetic/Melodic link the system libjpeg-turbo).
 * This file restates the published algorithm: ITU-T T.81 baseline sequential Huffman decoding, and
 * the arithmetic libjpeg 6b / libjpeg-turbo document for their default path -- the "slow but
 * accurate" 13-bit integer inverse DCT (Loeffler-Ligtenberg-Moschytz, two passes), the triangle
 * ("fancy") chroma upsampling for 2x1 and 2x2 subsampling with its alternating rounding constants,
 * and the 16-bit fixed-point YCbCr -> RGB conversion.  PINNED: tests/golden/jpeg_*.npz hold JPEG
 * streams and the pixels Pillow's bundled libjpeg-turbo decodes them to (generated in the build
 * container by tests/golden/make_golden_jpeg.py); this decoder reproduces them bit for bit.
 *
 * Deliberate differences from libjpeg: corrupt or truncated entropy data is an error (libjpeg pads
 * with zero bits and warns); progressive / arithmetic / lossless / 12-bit / CMYK streams and
 * sampling layouts other than 4:4:4, 4:2:2 (2x1), 4:2:0 (2x2) and grayscale are "unsupported".
 * Integer intermediates wrap in 32 bits (only reachable with adversarial coefficients).
 */
#include "lf_oracle.h"

#include <stdlib.h>
#include <string.h>

enum { JERR_OK = 0, JERR_CORRUPT = -6, JERR_UNSUPPORTED = -5 };

/* zigzag position -> natural (row-major) position, T.81 figure A.6 */
static const uint8_t kNatural[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63
};

/* T.81 Annex K.3 typical Huffman tables, used by streams that carry none (MJPEG-style frames) */
static const uint8_t kStdDcLumBits[16] = { 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0 };
static const uint8_t kStdDcChrBits[16] = { 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0 };
static const uint8_t kStdDcVals[12] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11 };
static const uint8_t kStdAcLumBits[16] = { 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d };
static const uint8_t kStdAcLumVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91,
    0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a,
    0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53,
    0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79,
    0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
    0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9,
    0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2,
    0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa
};
static const uint8_t kStdAcChrBits[16] = { 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77 };
static const uint8_t kStdAcChrVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14,
    0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17,
    0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a,
    0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78,
    0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7,
    0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2,
    0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa
};

typedef struct {
    int present;
    uint8_t bits[16];
    uint8_t vals[256];
} huff_t;

typedef struct {
    int id, h, v, tq, td, ta;
    int bw, bh;          /* blocks per row / column of the padded plane */
    int dw, dh;          /* real (downsampled) size */
    uint8_t* plane;
    int pred;
} comp_t;

typedef struct {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc;
    int nbits;
    int bad;
} bits_t;

static int next_bit(bits_t* b)
{
    if (b->nbits == 0) {
        if (b->p >= b->end) { b->bad = 1; return 0; }
        uint8_t c = *b->p++;
        if (c == 0xFF) {
            if (b->p >= b->end) { b->bad = 1; return 0; }
            if (*b->p == 0x00) b->p++;                     /* stuffed zero */
            else { b->p--; b->bad = 1; return 0; }         /* a marker inside the entropy data */
        }
        b->acc = c;
        b->nbits = 8;
    }
    b->nbits--;
    return (int)((b->acc >> b->nbits) & 1u);
}

static int receive(bits_t* b, int s)
{
    int v = 0;
    for (int i = 0; i < s; ++i) v = (v << 1) | next_bit(b);
    return v;
}

/* T.81 F.2.2.3: canonical code walk, one bit at a time */
static int huff_symbol(bits_t* b, const huff_t* t)
{
    int code = 0, first = 0, index = 0;
    for (int len = 0; len < 16; ++len) {
        code |= next_bit(b);
        if (b->bad) return -1;
        const int count = t->bits[len];
        if (code - first < count) return t->vals[index + (code - first)];
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    b->bad = 1;
    return -1;
}

static int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

/* post-IDCT range limit of libjpeg: index (x & 1023) into a table that is x+128 clamped to 0..255
 * for -384 <= x < 384 and wraps outside */
static uint8_t idct_limit(int32_t x)
{
    const int idx = (int)((uint32_t)x & 1023u);
    if (idx < 128) return (uint8_t)(128 + idx);
    if (idx < 512) return 255;
    if (idx < 896) return 0;
    return (uint8_t)(idx - 896);
}

#define MULW(a, c) ((int32_t)((uint32_t)(a) * (uint32_t)(c)))
#define ADDW(a, b) ((int32_t)((uint32_t)(a) + (uint32_t)(b)))
#define SUBW(a, b) ((int32_t)((uint32_t)(a) - (uint32_t)(b)))
#define SHLW(a, n) ((int32_t)((uint32_t)(a) << (n)))

/* one 8-point inverse DCT of the accurate integer method; in[] and out[] may alias through stride */
static void idct_1d(const int32_t in[8], int32_t out[8], int shift)
{
    int32_t z1 = MULW(ADDW(in[2], in[6]), 4433);
    int32_t t2 = ADDW(z1, MULW(in[6], -15137));
    int32_t t3 = ADDW(z1, MULW(in[2], 6270));
    int32_t t0 = SHLW(ADDW(in[0], in[4]), 13);
    int32_t t1 = SHLW(SUBW(in[0], in[4]), 13);
    const int32_t e0 = ADDW(t0, t3), e3 = SUBW(t0, t3), e1 = ADDW(t1, t2), e2 = SUBW(t1, t2);
    int32_t o0 = in[7], o1 = in[5], o2 = in[3], o3 = in[1];
    int32_t y1 = ADDW(o0, o3), y2 = ADDW(o1, o2), y3 = ADDW(o0, o2), y4 = ADDW(o1, o3);
    const int32_t y5 = MULW(ADDW(y3, y4), 9633);
    o0 = MULW(o0, 2446); o1 = MULW(o1, 16819); o2 = MULW(o2, 25172); o3 = MULW(o3, 12299);
    y1 = MULW(y1, -7373); y2 = MULW(y2, -20995);
    y3 = ADDW(MULW(y3, -16069), y5); y4 = ADDW(MULW(y4, -3196), y5);
    o0 = ADDW(o0, ADDW(y1, y3)); o1 = ADDW(o1, ADDW(y2, y4));
    o2 = ADDW(o2, ADDW(y2, y3)); o3 = ADDW(o3, ADDW(y1, y4));
    const int32_t r = (int32_t)1 << (shift - 1);
    out[0] = ADDW(ADDW(e0, o3), r) >> shift; out[7] = ADDW(SUBW(e0, o3), r) >> shift;
    out[1] = ADDW(ADDW(e1, o2), r) >> shift; out[6] = ADDW(SUBW(e1, o2), r) >> shift;
    out[2] = ADDW(ADDW(e2, o1), r) >> shift; out[5] = ADDW(SUBW(e2, o1), r) >> shift;
    out[3] = ADDW(ADDW(e3, o0), r) >> shift; out[4] = ADDW(SUBW(e3, o0), r) >> shift;
}

static void idct_block(const int32_t coef[64], uint8_t* dst, int stride)
{
    int32_t ws[64], col[8], res[8];
    for (int c = 0; c < 8; ++c) {                 /* pass 1: columns, keep 2 extra fraction bits */
        for (int r = 0; r < 8; ++r) col[r] = coef[r * 8 + c];
        idct_1d(col, res, 13 - 2);
        for (int r = 0; r < 8; ++r) ws[r * 8 + c] = res[r];
    }
    for (int r = 0; r < 8; ++r) {                 /* pass 2: rows, remove 2 + 3 bits, level shift, limit */
        idct_1d(&ws[r * 8], res, 13 + 2 + 3);
        for (int c = 0; c < 8; ++c) dst[r * stride + c] = idct_limit(res[c]);
    }
}

static uint8_t clamp255(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* chroma sample for output pixel (x, y) under libjpeg's fancy upsampling */
static int chroma_at(const comp_t* c, int hx, int vx, int x, int y)
{
    const uint8_t* pl = c->plane;
    const int stride = c->bw * 8;
    if (hx == 1 && vx == 1) return pl[y * stride + x];
    if (hx == 2 && vx == 1) {
        const uint8_t* in = pl + y * stride;
        const int cc = x >> 1;
        if (c->dw <= 2) return in[cc];                              /* plain replication for tiny planes */
        if ((x & 1) == 0) return cc == 0 ? in[0] : (3 * in[cc] + in[cc - 1] + 1) >> 2;
        return cc == c->dw - 1 ? in[cc] : (3 * in[cc] + in[cc + 1] + 2) >> 2;
    }
    /* 2 x 2 */
    {
        const int r = y >> 1, cc = x >> 1;
        if (c->dw <= 2) return pl[r * stride + cc];
        int rn = (y & 1) ? r + 1 : r - 1;                            /* the nearer neighbour row */
        if (rn < 0) rn = 0;
        if (rn > c->dh - 1) rn = c->dh - 1;
        const uint8_t* in0 = pl + r * stride;
        const uint8_t* in1 = pl + rn * stride;
        const int here = 3 * in0[cc] + in1[cc];
        if ((x & 1) == 0) {
            if (cc == 0) return (here * 4 + 8) >> 4;
            return (here * 3 + (3 * in0[cc - 1] + in1[cc - 1]) + 8) >> 4;
        }
        if (cc == c->dw - 1) return (here * 4 + 7) >> 4;
        return (here * 3 + (3 * in0[cc + 1] + in1[cc + 1]) + 7) >> 4;
    }
}

static int rd16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

typedef struct {
    int rows, cols, ncomp;
    comp_t comp[3];
    uint16_t qt[4][64];
    int qt_present[4];
    huff_t dc[4], ac[4];
    int restart;
    int hmax, vmax;
    int is_rgb, saw_jfif, saw_adobe, adobe_transform;
    const uint8_t* scan;           /* entropy-coded data */
    const uint8_t* end;
} jpeg_t;

static void set_std_table(huff_t* t, const uint8_t* bits, const uint8_t* vals, int n)
{
    t->present = 1;
    memcpy(t->bits, bits, 16);
    memcpy(t->vals, vals, (size_t)n);
}

/* parse everything up to and including the SOS header */
static int parse_headers(const uint8_t* d, size_t size, jpeg_t* j)
{
    memset(j, 0, sizeof(*j));
    if (size < 4 || d[0] != 0xFF || d[1] != 0xD8) return JERR_CORRUPT;
    size_t pos = 2;
    int have_sof = 0;
    for (;;) {
        if (pos + 4 > size) return JERR_CORRUPT;
        if (d[pos] != 0xFF) return JERR_CORRUPT;
        while (pos < size && d[pos] == 0xFF) pos++;                  /* fill bytes */
        if (pos >= size) return JERR_CORRUPT;
        const int m = d[pos++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return JERR_CORRUPT;                           /* EOI before any scan */
        if (pos + 2 > size) return JERR_CORRUPT;
        const int len = rd16(d + pos);
        if (len < 2 || pos + (size_t)len > size) return JERR_CORRUPT;
        const uint8_t* s = d + pos + 2;
        const int n = len - 2;
        if (m == 0xDB) {                                              /* DQT */
            int k = 0;
            while (k < n) {
                const int pq = s[k] >> 4, tq = s[k] & 15;
                k++;
                if (tq > 3 || pq > 1 || k + 64 * (pq + 1) > n) return JERR_CORRUPT;
                for (int i = 0; i < 64; ++i) {
                    const int v = pq ? rd16(s + k + 2 * i) : s[k + i];
                    j->qt[tq][kNatural[i]] = (uint16_t)v;
                }
                j->qt_present[tq] = 1;
                k += 64 * (pq + 1);
            }
        } else if (m == 0xC4) {                                       /* DHT */
            int k = 0;
            while (k < n) {
                if (k + 17 > n) return JERR_CORRUPT;
                const int tc = s[k] >> 4, th = s[k] & 15;
                if (tc > 1 || th > 3) return JERR_CORRUPT;
                huff_t* t = tc ? &j->ac[th] : &j->dc[th];
                int total = 0;
                for (int i = 0; i < 16; ++i) { t->bits[i] = s[k + 1 + i]; total += t->bits[i]; }
                if (total > 256 || k + 17 + total > n) return JERR_CORRUPT;
                memcpy(t->vals, s + k + 17, (size_t)total);
                t->present = 1;
                k += 17 + total;
            }
        } else if (m == 0xC0 || m == 0xC1) {                          /* baseline / extended sequential Huffman */
            if (n < 6 || s[0] != 8) return JERR_UNSUPPORTED;
            j->rows = rd16(s + 1);
            j->cols = rd16(s + 3);
            j->ncomp = s[5];
            if (j->rows <= 0 || j->cols <= 0) return JERR_CORRUPT;
            if (j->ncomp != 1 && j->ncomp != 3) return JERR_UNSUPPORTED;
            if (n < 6 + 3 * j->ncomp) return JERR_CORRUPT;
            for (int c = 0; c < j->ncomp; ++c) {
                comp_t* cp = &j->comp[c];
                cp->id = s[6 + 3 * c];
                cp->h = s[7 + 3 * c] >> 4;
                cp->v = s[7 + 3 * c] & 15;
                cp->tq = s[8 + 3 * c];
                if (cp->h < 1 || cp->h > 4 || cp->v < 1 || cp->v > 4 || cp->tq > 3) return JERR_CORRUPT;
            }
            have_sof = 1;
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return JERR_UNSUPPORTED;                                  /* progressive, lossless, arithmetic, ... */
        } else if (m == 0xDD) {
            if (n < 2) return JERR_CORRUPT;
            j->restart = rd16(s);
        } else if (m == 0xE0) {
            if (n >= 5 && memcmp(s, "JFIF", 5) == 0) j->saw_jfif = 1;
        } else if (m == 0xEE) {
            if (n >= 12 && memcmp(s, "Adobe", 5) == 0) { j->saw_adobe = 1; j->adobe_transform = s[11]; }
        } else if (m == 0xDA) {                                       /* SOS */
            if (!have_sof) return JERR_CORRUPT;
            if (n < 1 || s[0] != j->ncomp) return JERR_UNSUPPORTED;   /* one interleaved scan only */
            if (n < 1 + 2 * j->ncomp + 3) return JERR_CORRUPT;
            for (int c = 0; c < j->ncomp; ++c) {
                if (s[1 + 2 * c] != j->comp[c].id) return JERR_UNSUPPORTED;
                j->comp[c].td = s[2 + 2 * c] >> 4;
                j->comp[c].ta = s[2 + 2 * c] & 15;
                if (j->comp[c].td > 3 || j->comp[c].ta > 3) return JERR_CORRUPT;
            }
            const uint8_t* t = s + 1 + 2 * j->ncomp;
            if (t[0] != 0 || t[1] != 63 || t[2] != 0) return JERR_UNSUPPORTED;
            j->scan = d + pos + len;
            j->end = d + size;
            break;
        }
        pos += (size_t)len;
    }
    /* colour space guess of libjpeg (jdapimin.c default_decompress_parms) */
    if (j->ncomp == 3) {
        if (j->saw_jfif) j->is_rgb = 0;
        else if (j->saw_adobe) {
            if (j->adobe_transform == 0) j->is_rgb = 1;
            else if (j->adobe_transform == 1) j->is_rgb = 0;
            else j->is_rgb = 0;
        } else {
            j->is_rgb = (j->comp[0].id == 'R' && j->comp[1].id == 'G' && j->comp[2].id == 'B');
        }
    }
    j->hmax = j->vmax = 1;
    for (int c = 0; c < j->ncomp; ++c) {
        if (j->comp[c].h > j->hmax) j->hmax = j->comp[c].h;
        if (j->comp[c].v > j->vmax) j->vmax = j->comp[c].v;
    }
    if (j->ncomp == 1) { j->comp[0].h = j->comp[0].v = 1; j->hmax = j->vmax = 1; }   /* a single component is never subsampled */
    else {
        if (j->comp[0].h != j->hmax || j->comp[0].v != j->vmax) return JERR_UNSUPPORTED;
        if (j->comp[1].h != 1 || j->comp[1].v != 1 || j->comp[2].h != 1 || j->comp[2].v != 1) return JERR_UNSUPPORTED;
        if (!((j->hmax == 1 && j->vmax == 1) || (j->hmax == 2 && j->vmax == 1) || (j->hmax == 2 && j->vmax == 2)))
            return JERR_UNSUPPORTED;
    }
    for (int c = 0; c < j->ncomp; ++c) {
        if (!j->qt_present[j->comp[c].tq]) return JERR_CORRUPT;
        if (!j->dc[j->comp[c].td].present || !j->ac[j->comp[c].ta].present) {
            /* streams without DHT: Annex K tables, luminance in slot 0, chrominance in slot 1 */
            if (!j->dc[0].present) set_std_table(&j->dc[0], kStdDcLumBits, kStdDcVals, 12);
            if (!j->dc[1].present) set_std_table(&j->dc[1], kStdDcChrBits, kStdDcVals, 12);
            if (!j->ac[0].present) set_std_table(&j->ac[0], kStdAcLumBits, kStdAcLumVals, 162);
            if (!j->ac[1].present) set_std_table(&j->ac[1], kStdAcChrBits, kStdAcChrVals, 162);
            if (!j->dc[j->comp[c].td].present || !j->ac[j->comp[c].ta].present) return JERR_CORRUPT;
        }
    }
    return JERR_OK;
}

int lfo_jpeg_info(const uint8_t* data, size_t size, int* rows, int* cols, int* ncomp, int* hmax, int* vmax)
{
    jpeg_t j;
    const int rc = parse_headers(data, size, &j);
    if (rc != JERR_OK) return rc;
    *rows = j.rows; *cols = j.cols; *ncomp = j.ncomp; *hmax = j.hmax; *vmax = j.vmax;
    return JERR_OK;
}

/* bgr: rows x cols x 3, caller allocated for the size lfo_jpeg_info reports (NULL: skip the pixels);
 * qcoef: optional dump of the QUANTISED coefficients, [block in scan order][64 natural order] */
static int decode_impl(const uint8_t* data, size_t size, uint8_t* bgr, int16_t* qcoef, int cap_blocks, int* n_blocks)
{
    jpeg_t j;
    int rc = parse_headers(data, size, &j);
    if (rc != JERR_OK) return rc;
    const int mcux = (j.cols + 8 * j.hmax - 1) / (8 * j.hmax);
    const int mcuy = (j.rows + 8 * j.vmax - 1) / (8 * j.vmax);
    for (int c = 0; c < j.ncomp; ++c) {
        comp_t* cp = &j.comp[c];
        cp->bw = mcux * cp->h;
        cp->bh = mcuy * cp->v;
        cp->dw = (j.cols * cp->h + j.hmax - 1) / j.hmax;
        cp->dh = (j.rows * cp->v + j.vmax - 1) / j.vmax;
        cp->plane = (uint8_t*)malloc((size_t)cp->bw * 8 * cp->bh * 8);
        cp->pred = 0;
    }
    bits_t b = { j.scan, j.end, 0, 0, 0 };
    int next_rst = 0;
    long mcu_index = 0;
    long block_index = 0;
    for (int my = 0; my < mcuy && rc == JERR_OK; ++my) {
        for (int mx = 0; mx < mcux && rc == JERR_OK; ++mx, ++mcu_index) {
            if (j.restart && mcu_index > 0 && mcu_index % j.restart == 0) {
                b.nbits = 0;                                          /* discard the padding bits */
                if (b.p + 2 > b.end || b.p[0] != 0xFF || b.p[1] != (0xD0 + next_rst)) { rc = JERR_CORRUPT; break; }
                b.p += 2;
                next_rst = (next_rst + 1) & 7;
                for (int c = 0; c < j.ncomp; ++c) j.comp[c].pred = 0;
            }
            for (int c = 0; c < j.ncomp && rc == JERR_OK; ++c) {
                comp_t* cp = &j.comp[c];
                for (int v = 0; v < cp->v && rc == JERR_OK; ++v)
                    for (int h = 0; h < cp->h; ++h) {
                        int32_t coef[64];
                        memset(coef, 0, sizeof(coef));
                        int s = huff_symbol(&b, &j.dc[cp->td]);
                        if (s < 0 || s > 11) { rc = JERR_CORRUPT; break; }
                        const int diff = s ? extend(receive(&b, s), s) : 0;
                        cp->pred += diff;
                        int16_t* qd = (qcoef && block_index < cap_blocks) ? qcoef + block_index * 64 : NULL;
                        if (qd) { memset(qd, 0, 64 * sizeof(int16_t)); qd[0] = (int16_t)cp->pred; }
                        ++block_index;
                        coef[0] = MULW((int16_t)cp->pred, j.qt[cp->tq][0]);
                        for (int k = 1; k < 64;) {
                            const int rs = huff_symbol(&b, &j.ac[cp->ta]);
                            if (rs < 0) { rc = JERR_CORRUPT; break; }
                            const int run = rs >> 4, sz = rs & 15;
                            if (sz == 0) {
                                if (run != 15) break;                 /* end of block */
                                k += 16;
                                continue;
                            }
                            k += run;
                            if (k > 63) { rc = JERR_CORRUPT; break; }
                            const int val = extend(receive(&b, sz), sz);
                            coef[kNatural[k]] = MULW((int16_t)val, j.qt[cp->tq][kNatural[k]]);
                            if (qd) qd[kNatural[k]] = (int16_t)val;
                            ++k;
                        }
                        if (b.bad) rc = JERR_CORRUPT;
                        if (rc != JERR_OK) break;
                        const int stride = cp->bw * 8;
                        idct_block(coef, cp->plane + (size_t)((my * cp->v + v) * 8) * stride + (mx * cp->h + h) * 8, stride);
                    }
            }
        }
    }
    if (n_blocks) *n_blocks = (int)block_index;
    if (rc == JERR_OK && bgr) {
        /* libjpeg's YCbCr -> RGB constants: 16-bit fixed point, rounded once per table entry */
        const int32_t c_r = (int32_t)(1.40200 * 65536 + 0.5), c_b = (int32_t)(1.77200 * 65536 + 0.5);
        const int32_t c_gr = (int32_t)(0.71414 * 65536 + 0.5), c_gb = (int32_t)(0.34414 * 65536 + 0.5);
        for (int y = 0; y < j.rows; ++y)
            for (int x = 0; x < j.cols; ++x) {
                uint8_t* o = bgr + ((size_t)y * j.cols + x) * 3;
                const int Y = j.comp[0].plane[(size_t)y * j.comp[0].bw * 8 + x];
                if (j.ncomp == 1) { o[0] = o[1] = o[2] = (uint8_t)Y; continue; }
                const int u = chroma_at(&j.comp[1], j.hmax, j.vmax, x, y);
                const int w = chroma_at(&j.comp[2], j.hmax, j.vmax, x, y);
                if (j.is_rgb) { o[2] = (uint8_t)Y; o[1] = (uint8_t)u; o[0] = (uint8_t)w; continue; }
                const int cb = u - 128, cr = w - 128;
                const int r = Y + ((c_r * cr + 32768) >> 16);
                const int g = Y + ((-c_gb * cb + 32768 - c_gr * cr) >> 16);
                const int bl = Y + ((c_b * cb + 32768) >> 16);
                o[0] = clamp255(bl); o[1] = clamp255(g); o[2] = clamp255(r);
            }
    }
    for (int c = 0; c < j.ncomp; ++c) free(j.comp[c].plane);
    return rc;
}

int lfo_jpeg_decode(const uint8_t* data, size_t size, uint8_t* bgr)
{
    return decode_impl(data, size, bgr, NULL, 0, NULL);
}

/* quantised coefficients of every 8x8 block in scan (MCU) order, natural order inside a block: what
 * the product's host-side entropy decoder must reproduce (tests/test_jpeg_host.py) */
int lfo_jpeg_coefficients(const uint8_t* data, size_t size, int16_t* qcoef, int cap_blocks, int* n_blocks)
{
    return decode_impl(data, size, NULL, qcoef, cap_blocks, n_blocks);
}
