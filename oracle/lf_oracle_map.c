/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Sequential statement of the live map + associator contract
 * (include/lanefront.h, "live map"; SURVEY a-11, 8f-3).
 *
 * PARITY STATUS: the matching rule restates BinaryDescriptorMatcher::match
 * (/root/reference/src/line_descriptor/src/binary_descriptor_matcher.cpp:197-254: exact Hamming nearest neighbour,
 * nothing beyond D = 128 (:721); ties -> lowest index, a documented deviation from the reference's hash-discovery
 * order) -- pinned only as far as lfo_match is.  Everything else here (colour gating, append / merge policies, ring,
 * pose transform) has NO reference behaviour to be pinned to: the reference's line_associator node is a stub
 * (src/line_associator/src/line_associator_node.py:12-86) and its map is an append-only list in the robot frame
 * (src/show_map/src/show_map.py:28-42, pose published separately by src/odometry/src/odometry.py:110-120).  This
 * file DEFINES that contract; the HIP implementation (k_assoc.hip, k_map.hip) is held to it bit for bit.
 */
#include <stdlib.h>
#include <string.h>
#include "lf_oracle.h"
#include "lf_detmath.h"

struct lfo_map {
    lfo_map_config cfg;
    int tie_rule;
    int size, head, overflow;
    long long total_appended, total_refreshed;
    uint8_t* code;      /* capacity * 32 */
    uint8_t* color;
    double* ground;     /* capacity * 4 */
    int32_t* hits;
    int32_t* last_seen;
};

lfo_map* lfo_map_create(const lfo_map_config* cfg)
{
    lfo_map* m = (lfo_map*)calloc(1, sizeof(lfo_map));
    m->cfg = *cfg;
    const size_t cap = (size_t)cfg->capacity;
    m->code = (uint8_t*)calloc(cap, 32);
    m->color = (uint8_t*)calloc(cap, 1);
    m->ground = (double*)calloc(cap * 4, sizeof(double));
    m->hits = (int32_t*)calloc(cap, sizeof(int32_t));
    m->last_seen = (int32_t*)calloc(cap, sizeof(int32_t));
    return m;
}

void lfo_map_destroy(lfo_map* m)
{
    if (!m) return;
    free(m->code); free(m->color); free(m->ground); free(m->hits); free(m->last_seen);
    free(m);
}

void lfo_map_state(const lfo_map* m, int32_t* size, int32_t* head, int32_t* overflow, long long* total_appended,
                   long long* total_refreshed)
{
    if (size) *size = m->size;
    if (head) *head = m->head;
    if (overflow) *overflow = m->overflow;
    if (total_appended) *total_appended = m->total_appended;
    if (total_refreshed) *total_refreshed = m->total_refreshed;
}

const uint8_t* lfo_map_codes(const lfo_map* m) { return m->code; }
const uint8_t* lfo_map_colors(const lfo_map* m) { return m->color; }
const double* lfo_map_ground(const lfo_map* m) { return m->ground; }
const int32_t* lfo_map_hits(const lfo_map* m) { return m->hits; }
const int32_t* lfo_map_last_seen(const lfo_map* m) { return m->last_seen; }

static void put(lfo_map* m, int pos, const uint8_t* code, uint8_t color, const double* g, int step)
{
    memcpy(m->code + (size_t)pos * 32, code, 32);
    m->color[pos] = color;
    if (g) memcpy(m->ground + (size_t)pos * 4, g, 4 * sizeof(double));
    else memset(m->ground + (size_t)pos * 4, 0, 4 * sizeof(double));
    m->last_seen[pos] = step;
}

static void append(lfo_map* m, const uint8_t* code, uint8_t color, const double* g, int step)
{
    const int cap = m->cfg.capacity;
    m->total_appended += 1;
    if (m->cfg.when_full == LFO_MAP_RING) {
        /* wraps around and overwrites the oldest entry (whatever it is) */
        put(m, m->head, code, color, g, step);
        m->hits[m->head] = 1;
        m->head = (m->head + 1) % cap;
        if (m->size < cap) m->size += 1;
    } else {
        if (m->size >= cap) { m->overflow = 1; return; }
        put(m, m->size, code, color, g, step);
        m->hits[m->size] = 1;
        m->size += 1;
        m->head = m->size % cap;
    }
}

/* append entries as they are: colour NULL -> 255 (matches every colour), ground NULL -> zeros, last_seen -1 */
void lfo_map_seed(lfo_map* m, const uint8_t* code32, const uint8_t* color, const double* ground4, int n)
{
    for (int i = 0; i < n; ++i)
        append(m, code32 + (size_t)i * 32, color ? color[i] : (uint8_t)255, ground4 ? ground4 + (size_t)i * 4 : NULL, -1);
}

static int hamming256(const uint8_t* a, const uint8_t* b)
{
    /* bitops_custom.hpp:83-96 (popcount over the 32 bytes) */
    int d = 0;
    for (int w = 0; w < 4; ++w) {
        uint64_t x, y;
        memcpy(&x, a + 8 * w, 8);
        memcpy(&y, b + 8 * w, 8);
        d += __builtin_popcountll(x ^ y);
    }
    return d;
}

/* nearest entry of each query; with gating only entries of the query's colour (a colour >= 3 on either side matches
 * everything); farther than max_distance -> idx -1, dist -1; ties -> lowest index */
void lfo_map_set_tie_rule(lfo_map* m, int rule) { m->tie_rule = rule; }

void lfo_map_associate(const lfo_map* m, const uint8_t* code32, const uint8_t* color, int n, int32_t* idx, float* dist)
{
    for (int q = 0; q < n; ++q) {
        int best = 1 << 30, arg = -1;
        const uint8_t qc = color ? color[q] : (uint8_t)255;
        for (int j = 0; j < m->size; ++j) {
            if (m->cfg.color_gating && qc < 3 && m->color[j] < 3 && m->color[j] != qc) continue;
            const int d = hamming256(code32 + (size_t)q * 32, m->code + (size_t)j * 32);
            if (d < best) { best = d; arg = j; }
        }
        if (m->tie_rule == 1 && arg >= 0 && best <= 128) {
            /* among the equally near (and eligible) entries: the one the reference's search meets first */
            long long bkey = -1;
            for (int j = 0; j < m->size; ++j) {
                if (m->cfg.color_gating && qc < 3 && m->color[j] < 3 && m->color[j] != qc) continue;
                if (hamming256(code32 + (size_t)q * 32, m->code + (size_t)j * 32) != best) continue;
                const long long key = lfo_mih_discovery_key(code32 + (size_t)q * 32, m->code + (size_t)j * 32);
                if (bkey < 0 || key < bkey) { bkey = key; arg = j; }
            }
        }
        if (arg >= 0 && best <= m->cfg.max_distance) { idx[q] = arg; dist[q] = (float)best; }
        else { idx[q] = -1; dist[q] = -1.f; }
    }
}

/* ground endpoints of each segment into the map frame with its frame's pose (x, y, theta): the tf odometry publishes
 * (odometry.py:115-119).  cos / sin through the deterministic routines; products and sums unfused, in this order. */
void lfo_map_to_map_frame(const double* ground4, int n, const int32_t* frame_offset, int n_frames, const double* pose3,
                          double* out4)
{
    for (int f = 0; f < n_frames; ++f) {
        const double x = pose3[3 * f], y = pose3[3 * f + 1];
        const double cs = lfo_cos(pose3[3 * f + 2]), sn = lfo_sin(pose3[3 * f + 2]);
        for (int s = frame_offset[f]; s < frame_offset[f + 1] && s < n; ++s)
            for (int e = 0; e < 2; ++e) {
                const double px = ground4[(size_t)s * 4 + 2 * e], py = ground4[(size_t)s * 4 + 2 * e + 1];
                const double a = cs * px, b = sn * py, c = sn * px, d = cs * py;
                out4[(size_t)s * 4 + 2 * e] = x + (a - b);
                out4[(size_t)s * 4 + 2 * e + 1] = y + (c + d);
            }
    }
}

/* one update with n segments in SegmentList order; ground already in the map frame; idx / dist = their association
 * against the map as it stood before this call */
void lfo_map_update(lfo_map* m, const uint8_t* code32, const uint8_t* color, const uint8_t* keep, const double* ground4,
                    const int32_t* idx, const float* dist, int n, int step)
{
    const int size0 = m->size;
    uint8_t* refreshed = (uint8_t*)calloc((size_t)n + 1, 1);
    /* pass 1: refreshes, in order (the last one to touch an entry leaves its data there) */
    for (int s = 0; s < n; ++s) {
        const int eligible = !m->cfg.kept_only || (keep ? keep[s] != 0 : 1);
        if (!eligible || m->cfg.policy != LFO_MAP_MERGE) continue;
        if (idx[s] < 0 || idx[s] >= size0 || !(dist[s] >= 0.f) || dist[s] > (float)m->cfg.merge_distance) continue;
        const int j = idx[s];
        put(m, j, code32 + (size_t)s * 32, color ? color[s] : (uint8_t)255, ground4 ? ground4 + (size_t)s * 4 : NULL, step);
        m->hits[j] += 1;
        m->total_refreshed += 1;
        refreshed[s] = 1;
    }
    /* pass 2: every other eligible segment is appended, in order */
    for (int s = 0; s < n; ++s) {
        const int eligible = !m->cfg.kept_only || (keep ? keep[s] != 0 : 1);
        if (!eligible || refreshed[s]) continue;
        append(m, code32 + (size_t)s * 32, color ? color[s] : (uint8_t)255, ground4 ? ground4 + (size_t)s * 4 : NULL, step);
    }
    free(refreshed);
}
