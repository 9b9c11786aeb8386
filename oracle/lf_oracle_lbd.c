/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see lf_oracle.h).
 * a-9 LBD descriptor and a-10 matcher semantics, restated from the reference's
 * in-tree C++ (never built by the reference, needs OpenCV headers -> cannot be
 * compiled here without stand-ins -> PARITY UNPINNED):
 *   /root/reference/src/line_descriptor/src/binary_descriptor_custom.cpp
 *     :74-107   pair table            :217-259  Gaussian weights
 *     :401-412  binaryConversion      :653-667  32-byte code
 *     :1026-1372 computeLBD
 *   /root/reference/src/line_descriptor/src/LSDDetector_custom.cpp:73-102,169-197 (KeyLine fill)
 *   /root/reference/src/line_descriptor/src/binary_descriptor_matcher.cpp:197-254,635-753
 *   /root/reference/src/line_descriptor/src/bitops_custom.hpp:83-96 (Hamming)
 */
#include "lf_oracle.h"
#include "lf_detmath.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

#define NUM_OF_BANDS 9
#define MAX_WIDTH_OF_BAND 32
/* BinaryDescriptor::Params::widthOfBand_ (binary_descriptor_custom.cpp:108-115: 7 by default; setWidthOfBand :134-176 recomputes both
 * Gaussian tables from it).  A process-wide switch like lfo_lsd_set_seed_order: every lfo_lbd call -- frame, KeyLines, compute -- uses it. */
static int g_width_of_band = 7;
void lfo_lbd_set_width_of_band(int w) { g_width_of_band = w < 1 ? 1 : (w > MAX_WIDTH_OF_BAND ? MAX_WIDTH_OF_BAND : w); }
int lfo_lbd_get_width_of_band(void) { return g_width_of_band; }

static const int COMB[32][2] = {
    {0,1},{0,2},{0,3},{0,4},{0,5},{0,6},{1,2},{1,3},{1,4},{1,5},{1,6},{2,3},{2,4},{2,5},{2,6},{2,7},
    {2,8},{3,4},{3,5},{3,6},{3,7},{3,8},{4,5},{4,6},{4,7},{4,8},{5,6},{5,7},{5,8},{6,7},{6,8},{7,8} };

static int cv_round(double v)
{
    double f = (double)(long long)v;
    double d = v - f;
    long long i = (long long)f;
    if (d > 0.5 || (d == 0.5 && (i & 1))) i += 1;
    else if (d < -0.5 || (d == -0.5 && (i & 1))) i -= 1;
    return (int)i;
}

static double c_round(double v) /* C round(): half away from zero */
{
    double f = (double)(long long)v;
    double d = v - f;
    if (d >= 0.5) return f + 1.0;
    if (d <= -0.5) return f - 1.0;
    return f;
}

/* LSDDetector_custom.cpp:73-102 checkLineExtremes, :169-197 KeyLine fields (octave 0, scale 1) */
void lfo_keylines(const float* lines, int n, int rows, int cols, float* ext, float* angle, int32_t* npx)
{
    for (int i = 0; i < n; ++i) {
        float e[4] = { lines[4 * i], lines[4 * i + 1], lines[4 * i + 2], lines[4 * i + 3] };
        if (e[0] < 0) e[0] = 0;
        if (e[0] >= cols) e[0] = (float)cols - 1.0f;
        if (e[2] < 0) e[2] = 0;
        if (e[2] >= cols) e[2] = (float)cols - 1.0f;
        if (e[1] < 0) e[1] = 0;
        if (e[1] >= rows) e[1] = (float)rows - 1.0f;
        if (e[3] < 0) e[3] = 0;
        if (e[3] >= rows) e[3] = (float)rows - 1.0f;
        for (int k = 0; k < 4; ++k) ext[4 * i + k] = e[k];
        /* cv::LineIterator(img, Point(round), Point(round)), 8-connected: count = max(|dx|,|dy|)+1 */
        int x0 = cv_round(e[0]), y0 = cv_round(e[1]), x1 = cv_round(e[2]), y1 = cv_round(e[3]);
        int adx = x1 - x0; if (adx < 0) adx = -adx;
        int ady = y1 - y0; if (ady < 0) ady = -ady;
        npx[i] = (adx > ady ? adx : ady) + 1;
        float ddy = e[3] - e[1], ddx = e[2] - e[0];
        angle[i] = (float)lfo_atan2((double)ddy, (double)ddx);
    }
}

void lfo_lbd(const int16_t* pdx, const int16_t* pdy, int rows, int cols,
             const float* ext, const float* angle, const int32_t* npx, int n,
             float* desc72, uint8_t* code32)
{
    /* binary_descriptor_custom.cpp:217-259 = :134-176 (integer divisions in u and sigma kept) */
    const int WIDTH_OF_BAND = g_width_of_band;
    double gaussCoefL[MAX_WIDTH_OF_BAND * 3], gaussCoefG[NUM_OF_BANDS * MAX_WIDTH_OF_BAND];
    {
        double u = (WIDTH_OF_BAND * 3 - 1) / 2;
        double sigma = (WIDTH_OF_BAND * 2 + 1) / 2;
        double invsigma2 = -1 / (2 * sigma * sigma);
        for (int i = 0; i < WIDTH_OF_BAND * 3; ++i) { double dis = i - u; gaussCoefL[i] = lfo_exp(dis * dis * invsigma2); }
        u = (NUM_OF_BANDS * WIDTH_OF_BAND - 1) / 2;
        sigma = u;
        invsigma2 = -1 / (2 * sigma * sigma);
        for (int i = 0; i < NUM_OF_BANDS * WIDTH_OF_BAND; ++i) { double dis = i - u; gaussCoefG[i] = lfo_exp(dis * dis * invsigma2); }
    }
    const short heightOfLSP = (short)(WIDTH_OF_BAND * NUM_OF_BANDS);
    const short halfHeight = (heightOfLSP - 1) / 2;
    const short realWidth = (short)cols;
    const short imageWidth = realWidth - 1;
    const short imageHeight = (short)(rows - 1);
    for (int li = 0; li < n; ++li) {
        float pgdLBandSum[NUM_OF_BANDS] = {0}, ngdLBandSum[NUM_OF_BANDS] = {0};
        float pgdL2BandSum[NUM_OF_BANDS] = {0}, ngdL2BandSum[NUM_OF_BANDS] = {0};
        float pgdOBandSum[NUM_OF_BANDS] = {0}, ngdOBandSum[NUM_OF_BANDS] = {0};
        float pgdO2BandSum[NUM_OF_BANDS] = {0}, ngdO2BandSum[NUM_OF_BANDS] = {0};
        const short lengthOfLSP = (short)npx[li];
        const short halfWidth = (lengthOfLSP - 1) / 2;
        const float sX = ext[4 * li], sY = ext[4 * li + 1], eX = ext[4 * li + 2], eY = ext[4 * li + 3];
        const float lineMiddlePointX = (float)(0.5 * (sX + eX));
        const float lineMiddlePointY = (float)(0.5 * (sY + eY));
        float dL[2], dO[2];
        /* :1130-1131 `dL[0] = cos( pSingleLine->direction )`, direction a float (descriptor_custom.hpp:396).  Unlike
         * sqrt, cos / sin are NOT among cvstd.hpp's using-declarations, so inside namespace cv::line_descriptor the
         * unqualified name resolves to the global ::cos(double) of <math.h> that <cmath> (precomp_custom.hpp:62)
         * drags in: the float is promoted, the cosine is a double and is rounded once into the float dL (checked with
         * g++ 11 / libstdc++ on the same include list: decltype(cos(float)) is double there, decltype(sqrt(float))
         * float once `using std::sqrt` is in scope). */
        dL[0] = (float)lfo_cos((double)angle[li]);
        dL[1] = (float)lfo_sin((double)angle[li]);
        dO[0] = -dL[1];
        dO[1] = dL[0];
        float sCorX0 = -dL[0] * halfWidth + dL[1] * halfHeight + lineMiddlePointX;
        float sCorY0 = -dL[1] * halfWidth - dL[0] * halfHeight + lineMiddlePointY;
        for (short hID = 0; hID < heightOfLSP; hID++) {
            float sCorX = sCorX0, sCorY = sCorY0;
            float pgdLRowSum = 0, ngdLRowSum = 0, pgdORowSum = 0, ngdORowSum = 0;
            for (short wID = 0; wID < lengthOfLSP; wID++) {
                short tempCor = (short)c_round((double)sCorX);
                short xCor = (tempCor < 0) ? 0 : (tempCor > imageWidth) ? imageWidth : tempCor;
                tempCor = (short)c_round((double)sCorY);
                short yCor = (tempCor < 0) ? 0 : (tempCor > imageHeight) ? imageHeight : tempCor;
                short dx = pdx[yCor * realWidth + xCor];
                short dy = pdy[yCor * realWidth + xCor];
                float gDL = dx * dL[0] + dy * dL[1];
                float gDO = dx * dO[0] + dy * dO[1];
                if (gDL > 0) pgdLRowSum += gDL; else ngdLRowSum -= gDL;
                if (gDO > 0) pgdORowSum += gDO; else ngdORowSum -= gDO;
                sCorX += dL[0];
                sCorY += dL[1];
            }
            sCorX0 -= dL[1];
            sCorY0 += dL[0];
            float coef = (float)gaussCoefG[hID];
            pgdLRowSum = coef * pgdLRowSum;
            ngdLRowSum = coef * ngdLRowSum;
            float pgdL2RowSum = pgdLRowSum * pgdLRowSum;
            float ngdL2RowSum = ngdLRowSum * ngdLRowSum;
            pgdORowSum = coef * pgdORowSum;
            ngdORowSum = coef * ngdORowSum;
            float pgdO2RowSum = pgdORowSum * pgdORowSum;
            float ngdO2RowSum = ngdORowSum * ngdORowSum;
            short bandID = (short)(hID / WIDTH_OF_BAND);
            for (int pass = 0; pass < 3; ++pass) {
                short b; int tap;
                if (pass == 0) { b = bandID; tap = hID % WIDTH_OF_BAND + WIDTH_OF_BAND; }
                else if (pass == 1) { b = bandID - 1; tap = hID % WIDTH_OF_BAND + 2 * WIDTH_OF_BAND; if (b < 0) continue; }
                else { b = bandID + 1; tap = hID % WIDTH_OF_BAND; if (b >= NUM_OF_BANDS) continue; }
                coef = (float)gaussCoefL[tap];
                pgdLBandSum[b] += coef * pgdLRowSum;
                ngdLBandSum[b] += coef * ngdLRowSum;
                pgdL2BandSum[b] += coef * coef * pgdL2RowSum;
                ngdL2BandSum[b] += coef * coef * ngdL2RowSum;
                pgdOBandSum[b] += coef * pgdORowSum;
                ngdOBandSum[b] += coef * ngdORowSum;
                pgdO2BandSum[b] += coef * coef * pgdO2RowSum;
                ngdO2BandSum[b] += coef * coef * ngdO2RowSum;
            }
        }
        float* desVec = desc72 + (size_t)72 * li;
        const float invN2 = (float)(1.0 / (WIDTH_OF_BAND * 2.0));
        const float invN3 = (float)(1.0 / (WIDTH_OF_BAND * 3.0));
        for (int b = 0; b < NUM_OF_BANDS; ++b) {
            float invN = (b == 0 || b == NUM_OF_BANDS - 1) ? invN2 : invN3;
            int d = b * 8;
            float temp = pgdLBandSum[b] * invN;
            desVec[d] = temp;
            desVec[d + 4] = (float)sqrt((double)(pgdL2BandSum[b] * invN - temp * temp));
            temp = ngdLBandSum[b] * invN;
            desVec[d + 1] = temp;
            desVec[d + 5] = (float)sqrt((double)(ngdL2BandSum[b] * invN - temp * temp));
            temp = pgdOBandSum[b] * invN;
            desVec[d + 2] = temp;
            desVec[d + 6] = (float)sqrt((double)(pgdO2BandSum[b] * invN - temp * temp));
            temp = ngdOBandSum[b] * invN;
            desVec[d + 3] = temp;
            desVec[d + 7] = (float)sqrt((double)(ngdO2BandSum[b] * invN - temp * temp));
        }
        float tempM = 0, tempS = 0;
        for (int b = 0; b < NUM_OF_BANDS; ++b) {
            const float* v = desVec + 8 * b;
            tempM += v[0] * v[0]; tempM += v[1] * v[1]; tempM += v[2] * v[2]; tempM += v[3] * v[3];
            tempS += v[4] * v[4]; tempS += v[5] * v[5]; tempS += v[6] * v[6]; tempS += v[7] * v[7];
        }
        /* :1301-1302 `tempM = 1 / sqrt( tempM )` with float tempM inside namespace cv: cvstd.hpp's `using std::sqrt`
         * makes this std::sqrt(float), and int / float is a float division -- TWO float roundings (sqrt, then the
         * quotient), not one rounding of a double quotient.  A float sqrt / quotient computed in double and rounded
         * once more is the correctly rounded float result (53 >= 2*24 + 2). */
        tempM = (float)(1.0 / (double)(float)sqrt((double)tempM));
        tempS = (float)(1.0 / (double)(float)sqrt((double)tempS));
        for (int b = 0; b < NUM_OF_BANDS; ++b) {
            float* v = desVec + 8 * b;
            v[0] = v[0] * tempM; v[1] = v[1] * tempM; v[2] = v[2] * tempM; v[3] = v[3] * tempM;
            v[4] = v[4] * tempS; v[5] = v[5] * tempS; v[6] = v[6] * tempS; v[7] = v[7] * tempS;
        }
        for (int i = 0; i < 72; ++i) if ((double)desVec[i] > 0.4) desVec[i] = (float)0.4;
        float temp = 0;
        for (int i = 0; i < 72; ++i) temp += desVec[i] * desVec[i];
        temp = (float)(1.0 / (double)(float)sqrt((double)temp));          /* :1337, same two roundings */
        for (int i = 0; i < 72; ++i) desVec[i] = desVec[i] * temp;
        /* :653-667 + :401-412 */
        uint8_t* code = code32 + (size_t)32 * li;
        for (int cidx = 0; cidx < 32; ++cidx) {
            const float* f1 = desVec + 8 * COMB[cidx][0];
            const float* f2 = desVec + 8 * COMB[cidx][1];
            unsigned r = 0;
            for (int i = 0; i < 8; ++i) if (f1[i] > f2[i]) r += 1u << i;
            code[cidx] = (uint8_t)r;
        }
    }
}

/* binary_descriptor_matcher.cpp:197-254 with Mihasher(256,32), K=1: exact Hamming
 * nearest neighbour; candidates farther than D = 128 are never reported
 * (:721, the result slot stays unset -> defined here as idx -1, dist -1).
 * The reference breaks distance ties by hash-table discovery order; the oracle
 * (and the GPU) take the lowest train index; tests compare indices only where
 * the minimum is unique.
 */
void lfo_match(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, float* dist)
{
    for (int i = 0; i < nq; ++i) {
        int best = 1 << 30, bi = -1;
        const uint32_t* a = (const uint32_t*)(q + (size_t)32 * i);
        for (int j = 0; j < nt; ++j) {
            const uint32_t* b = (const uint32_t*)(t + (size_t)32 * j);
            int d = 0;
            for (int k = 0; k < 8; ++k) d += __builtin_popcount(a[k] ^ b[k]);
            if (d < best) { best = d; bi = j; }
        }
        if (bi >= 0 && best <= 128) { idx[i] = bi; dist[i] = (float)best; }
        else { idx[i] = -1; dist[i] = -1.f; }
    }
}

/* The same nearest neighbour with the REFERENCE's tie rule, for quantifying the documented deviation (lowest index
 * here and on the GPU).  Mihasher::query (binary_descriptor_matcher.cpp:635-753) with B = 256, m = 32 (8-bit
 * substrings, all of them: mplus = 32), D = 128, d = 4, K = 1 visits, for radius s = 0, 1, ... and substring k = 0..31, the
 * buckets H[k][chunk_k(query) ^ bitstr] over all bitstr of weight s in the order its combination loop (:681-741)
 * produces them; a bucket lists its codes in insertion = train index order (BucketGroup::insert appends at the end of
 * the bucket's range, :927-947; populate inserts i = 0 .. N-1, :806-819).  The first time an index shows up its full
 * distance is computed and the FIRST index seen per distance is kept (:716-722); after finishing (s, k) the search
 * stops as soon as a code at distance exactly s * 32 + k has been seen (:744-746) -- by then every code within that
 * distance has, so the answer is the exact nearest neighbour, and among equally near ones the one whose earliest
 * discovery (s, k, position of bitstr in the enumeration, train index) comes first. */
static int mih_rank[5][256];
static int mih_rank_ready = 0;

static void mih_build_ranks(void)
{
    for (int s = 0; s <= 4; ++s) {
        int power[8], order = 0;
        const int curb = 8;
        unsigned long long bitstr = 0;
        for (int i = 0; i < s; i++) power[i] = i;
        power[s] = curb + 1;
        int bit = s - 1;
        for (;;) {
            if (bit != -1) {
                bitstr ^= (power[bit] == bit) ? (1ull << power[bit]) : (3ull << (power[bit] - 1));
                power[bit]++;
                bit--;
            } else {
                mih_rank[s][bitstr & 255] = order++;
                while (++bit < s && power[bit] == power[bit + 1] - 1) {
                    bitstr ^= 1ull << (power[bit] - 1);
                    power[bit] = bit;
                }
                if (bit == s) break;
            }
        }
    }
    mih_rank_ready = 1;
}

/* when Mihasher::query first meets train code b while searching for a (see above): (radius, substring, position of the
 * bit string in the enumeration); -1 when no substring is within 4 bits (the code is then farther than 128 bits) */
long long lfo_mih_discovery_key(const uint8_t* a, const uint8_t* b)
{
    if (!mih_rank_ready) mih_build_ranks();
    long long key = -1;
    for (int k = 0; k < 32; ++k) {
        const int x = a[k] ^ b[k], h = __builtin_popcount((unsigned)x);
        if (h > 4) continue;
        const long long kk = (((long long)h * 32 + k) * 256 + mih_rank[h][x]);
        if (key < 0 || kk < key) key = kk;
    }
    return key;
}

void lfo_match_mih(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, float* dist, int32_t* n_ties)
{
    if (!mih_rank_ready) mih_build_ranks();
    int* dd = (int*)malloc((size_t)(nt > 0 ? nt : 1) * sizeof(int));
    for (int i = 0; i < nq; ++i) {
        const uint8_t* a = q + (size_t)32 * i;
        int best = 1 << 30;
        for (int j = 0; j < nt; ++j) {
            const uint8_t* b = t + (size_t)32 * j;
            int d = 0;
            for (int k = 0; k < 32; ++k) d += __builtin_popcount((unsigned)(a[k] ^ b[k]));
            dd[j] = d;
            if (d < best) best = d;
        }
        int ties = 0, bi = -1;
        long long bkey = 0;
        if (nt > 0 && best <= 128) {
            for (int j = 0; j < nt; ++j) {
                if (dd[j] != best) continue;
                ties++;
                const uint8_t* b = t + (size_t)32 * j;
                long long key = -1;
                for (int k = 0; k < 32; ++k) {          /* substring k = byte k (split(), bitops_custom.hpp:99-124, b = 8) */
                    const int x = a[k] ^ b[k], h = __builtin_popcount((unsigned)x);
                    if (h > 4) continue;                /* never looked up in table k: radius d = 4 */
                    const long long kk = (((long long)h * 32 + k) * 256 + mih_rank[h][x]);
                    if (key < 0 || kk < key) key = kk;
                }
                /* key >= 0 always: a code within 128 bits has a substring within 4 */
                if (bi < 0 || key < bkey) { bi = j; bkey = key; }        /* equal keys: same bucket, lower train index first */
            }
        }
        if (bi >= 0) { idx[i] = bi; dist[i] = (float)best; }
        else { idx[i] = -1; dist[i] = -1.f; }
        if (n_ties) n_ties[i] = ties;
    }
    free(dd);
}

/* knnMatch / radiusMatch (binary_descriptor_matcher.cpp:258-335, 428-504): the nearest train codes within D = 128 bits,
 * nearest first, ties in index order (the reference: discovery order).  knn: idx / dist [nq][k], missing -> -1.
 * radius: CSR lists; returns the total (idx / dist may be NULL to size them). */
void lfo_knn_match(const uint8_t* q, int nq, const uint8_t* t, int nt, int k, int32_t* idx, float* dist)
{
    for (int i = 0; i < nq; ++i) {
        int got = 0;
        for (int d = 0; d <= 128 && got < k; ++d)
            for (int j = 0; j < nt && got < k; ++j) {
                int h = 0;
                for (int b = 0; b < 32; ++b) h += __builtin_popcount((unsigned)(q[(size_t)32 * i + b] ^ t[(size_t)32 * j + b]));
                if (h == d) { idx[(size_t)i * k + got] = j; dist[(size_t)i * k + got] = (float)d; ++got; }
            }
        for (; got < k; ++got) { idx[(size_t)i * k + got] = -1; dist[(size_t)i * k + got] = -1.f; }
    }
}

/* The same two with the REFERENCE's order among equally near codes: Mihasher::query records, per distance, the codes in the
 * order it discovers them (res[hammd * K + numres[hammd]], :716-722) and hands out the first K by distance, so the lists are
 * ordered by (distance, discovery key, train index) -- lfo_mih_discovery_key above; every code within D = 128 bits is discovered
 * before the search stops. */
typedef struct { int d; long long key; int j; } MihCand;
static int mih_cand_cmp(const void* a, const void* b)
{
    const MihCand* x = (const MihCand*)a; const MihCand* y = (const MihCand*)b;
    if (x->d != y->d) return x->d < y->d ? -1 : 1;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->j < y->j ? -1 : (x->j > y->j ? 1 : 0);
}
static int mih_candidates(const uint8_t* a, const uint8_t* t, int nt, int md, MihCand* c)
{
    int n = 0;
    for (int j = 0; j < nt; ++j) {
        int h = 0;
        for (int b = 0; b < 32; ++b) h += __builtin_popcount((unsigned)(a[b] ^ t[(size_t)32 * j + b]));
        if (h > md) continue;
        c[n].d = h; c[n].key = lfo_mih_discovery_key(a, t + (size_t)32 * j); c[n].j = j; ++n;
    }
    qsort(c, (size_t)n, sizeof(MihCand), mih_cand_cmp);
    return n;
}
void lfo_knn_match_mih(const uint8_t* q, int nq, const uint8_t* t, int nt, int k, int32_t* idx, float* dist)
{
    MihCand* c = (MihCand*)malloc(sizeof(MihCand) * (size_t)(nt > 0 ? nt : 1));
    for (int i = 0; i < nq; ++i) {
        const int n = mih_candidates(q + (size_t)32 * i, t, nt, 128, c);
        for (int g = 0; g < k; ++g) {
            idx[(size_t)i * k + g] = g < n ? c[g].j : -1;
            dist[(size_t)i * k + g] = g < n ? (float)c[g].d : -1.f;
        }
    }
    free(c);
}
int lfo_radius_match_mih(const uint8_t* q, int nq, const uint8_t* t, int nt, float max_distance, int32_t* offsets, int32_t* idx, float* dist)
{
    const int md = max_distance >= 128.f ? 128 : (int)max_distance;
    MihCand* c = (MihCand*)malloc(sizeof(MihCand) * (size_t)(nt > 0 ? nt : 1));
    int total = 0;
    for (int i = 0; i < nq; ++i) {
        offsets[i] = total;
        const int n = mih_candidates(q + (size_t)32 * i, t, nt, md, c);
        for (int g = 0; g < n; ++g) { if (idx) idx[total] = c[g].j; if (dist) dist[total] = (float)c[g].d; ++total; }
    }
    offsets[nq] = total;
    free(c);
    return total;
}

int lfo_radius_match(const uint8_t* q, int nq, const uint8_t* t, int nt, float max_distance, int32_t* offsets, int32_t* idx, float* dist)
{
    const int md = max_distance >= 128.f ? 128 : (int)max_distance;
    int total = 0;
    int* hd = (int*)malloc((size_t)(nt > 0 ? nt : 1) * sizeof(int));
    for (int i = 0; i < nq; ++i) {
        offsets[i] = total;
        for (int j = 0; j < nt; ++j) {
            int h = 0;
            for (int b = 0; b < 32; ++b) h += __builtin_popcount((unsigned)(q[(size_t)32 * i + b] ^ t[(size_t)32 * j + b]));
            hd[j] = h;
        }
        for (int d = 0; d <= md; ++d)
            for (int j = 0; j < nt; ++j)
                if (hd[j] == d) { if (idx) idx[total] = j; if (dist) dist[total] = (float)d; ++total; }
    }
    offsets[nq] = total;
    free(hd);
    return total;
}

/* float LBD nearest neighbour (Euclidean); double accumulation, float result */
void lfo_match_float(const float* q, int nq, const float* t, int nt, int32_t* idx, float* dist)
{
    for (int i = 0; i < nq; ++i) {
        double best = 1e300; int bi = -1;
        for (int j = 0; j < nt; ++j) {
            double s = 0;
            for (int k = 0; k < 72; ++k) { double d = (double)q[72 * (size_t)i + k] - (double)t[72 * (size_t)j + k]; s += d * d; }
            if (s < best) { best = s; bi = j; }
        }
        idx[i] = bi;
        dist[i] = bi >= 0 ? (float)sqrt(best) : -1.f;
    }
}
