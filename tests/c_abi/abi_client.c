/* A plain-C client of include/lanefront.h: what a non-Python integrator links against.
 * Reads a config blob and raw frames written by tests/test_c_abi.py, runs the batch entry point with host
 * pointers, associates the codes against themselves reversed, serialises the filtered SegmentList bodies and
 * writes everything back for the test to compare with the Python binding and the oracle.
 *
 *   abi_client <config.bin> <frames.bin> <n_frames> <out.bin>
 *
 * Compiled by gcc (C11), no HIP headers: only the C ABI.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/lanefront.h"

static void die(const char* what, lf_handle* h, int rc)
{
    fprintf(stderr, "%s failed: %d (%s)\n", what, rc, h ? lf_last_error(h) : "no handle");
    exit(2);
}

static void* xread(const char* path, size_t bytes)
{
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    void* p = malloc(bytes);
    if (fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "%s: short read\n", path); exit(2); }
    fclose(f);
    return p;
}

int main(int argc, char** argv)
{
    if (argc != 5) { fprintf(stderr, "usage: abi_client config.bin frames.bin n_frames out.bin\n"); return 2; }
    if (lf_abi_version() != LF_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 2; }
    lf_config* cfg = (lf_config*)xread(argv[1], sizeof(lf_config));
    const int n = atoi(argv[3]);
    const size_t frame_bytes = (size_t)cfg->in_rows * cfg->in_cols * 3;
    uint8_t* frames = (uint8_t*)xread(argv[2], frame_bytes * (size_t)n);
    const int cap_lines = 512, cap = n * 3 * cap_lines;

    lf_handle* h = NULL;
    int rc = lf_create(cfg, 0, n, cap_lines, &h);
    if (rc != LF_OK) die("lf_create", h, rc);

    lf_segments s;
    memset(&s, 0, sizeof(s));
    s.capacity = cap;
    s.frame_offset = (int32_t*)calloc((size_t)n + 1, sizeof(int32_t));
    s.lines = (float*)malloc(sizeof(float) * 4 * (size_t)cap);
    s.normals = (float*)malloc(sizeof(float) * 2 * (size_t)cap);
    s.color = (uint8_t*)malloc((size_t)cap);
    s.pixels_normalized = (float*)malloc(sizeof(float) * 4 * (size_t)cap);
    s.ground = (double*)malloc(sizeof(double) * 4 * (size_t)cap);
    s.keep = (uint8_t*)malloc((size_t)cap);
    s.code = (uint8_t*)malloc(32 * (size_t)cap);
    int total = 0;
    rc = lf_process_batch(h, frames, n, 0, &s, 0, 1, &total);
    if (rc != LF_OK) die("lf_process_batch", h, rc);

    /* association of the codes against the same codes in reverse order */
    uint8_t* rev = (uint8_t*)malloc(32 * (size_t)(total > 0 ? total : 1));
    for (int i = 0; i < total; ++i) memcpy(rev + 32 * (size_t)i, s.code + 32 * (size_t)(total - 1 - i), 32);
    int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)(total > 0 ? total : 1));
    float* dist = (float*)malloc(sizeof(float) * (size_t)(total > 0 ? total : 1));
    if (total > 0) {
        rc = lf_associate(h, s.code, total, rev, total, idx, dist, 0);
        if (rc != LF_OK) die("lf_associate", h, rc);
    }

    /* the associator component from plain C: two steps against a live map (append, then merge with colour gating off):
     * the second step must match every kept segment of the first at distance 0 */
    lf_map_config mc;
    memset(&mc, 0, sizeof(mc));
    mc.capacity = 4096; mc.max_distance = 128; mc.policy = LF_MAP_MERGE; mc.kept_only = 1; mc.merge_distance = 0; mc.when_full = LF_MAP_RING;
    lf_map* map = NULL;
    rc = lf_map_create(0, &mc, &map);
    if (rc != LF_OK) { fprintf(stderr, "lf_map_create failed: %d (%s)\n", rc, lf_map_last_error(NULL)); return 2; }
    int32_t* midx = (int32_t*)malloc(sizeof(int32_t) * (size_t)(total > 0 ? total : 1));
    float* mdist = (float*)malloc(sizeof(float) * (size_t)(total > 0 ? total : 1));
    int map_size1 = 0, map_size2 = 0, matched0 = 0;
    int64_t appended = 0, refreshed = 0;
    rc = lf_map_step_host(map, &s, total, n, NULL, 0, midx, mdist);
    if (rc != LF_OK) { fprintf(stderr, "lf_map_step_host failed: %d (%s)\n", rc, lf_map_last_error(map)); return 2; }
    rc = lf_map_size(map, &map_size1, NULL, NULL, NULL);
    if (rc != LF_OK) return 2;
    rc = lf_map_step_host(map, &s, total, n, NULL, 1, midx, mdist);
    if (rc != LF_OK) { fprintf(stderr, "lf_map_step_host (2) failed: %d (%s)\n", rc, lf_map_last_error(map)); return 2; }
    rc = lf_map_size(map, &map_size2, NULL, &appended, &refreshed);
    if (rc != LF_OK) return 2;
    for (int i = 0; i < total; ++i) if (s.keep[i] && midx[i] >= 0 && mdist[i] == 0.f) ++matched0;
    lf_map_destroy(map);

    /* SegmentList bodies line_sanity_node would publish */
    const size_t body_cap = 4 * (size_t)n + 73 * (size_t)total;
    uint8_t* body = (uint8_t*)malloc(body_cap ? body_cap : 1);
    int64_t* boff = (int64_t*)calloc((size_t)n + 1, sizeof(int64_t));
    rc = lf_serialize_segments(h, &s, 0, n, LF_MSG_FILTERED, body, body_cap, 0, boff);
    if (rc != LF_OK) die("lf_serialize_segments", h, rc);

    /* error path: too small a capacity must be reported, not overrun */
    lf_segments tiny = s;
    tiny.capacity = total > 1 ? total - 1 : 0;
    int t2 = 0;
    const int rc_small = total > 1 ? lf_process_batch(h, frames, n, 0, &tiny, 0, 0, &t2) : LF_ERR_CAPACITY;

    /* the second detector from plain C: EDLines over two octaves + descriptors (lf_keylines_batch) */
    const int kcap = 2048 * n;
    lf_keylines kl;
    memset(&kl, 0, sizeof(kl));
    kl.capacity = kcap;
    kl.frame_offset = (int32_t*)calloc((size_t)n + 1, sizeof(int32_t));
    kl.in_octave = (float*)malloc(sizeof(float) * 4 * (size_t)kcap);
    kl.octave = (int32_t*)malloc(sizeof(int32_t) * (size_t)kcap);
    kl.class_id = (int32_t*)malloc(sizeof(int32_t) * (size_t)kcap);
    kl.code = (uint8_t*)malloc(32 * (size_t)kcap);
    int n_kl = 0;
    rc = lf_keylines_batch(h, frames, n, 0, 0, 2, NULL, &kl, 0, 1, &n_kl, NULL);
    if (rc != LF_OK) die("lf_keylines_batch", h, rc);

    FILE* f = fopen(argv[4], "wb");
    if (!f) { perror(argv[4]); return 2; }
    int32_t head[4] = { total, rc_small, (int32_t)boff[n], LF_N_STAGES };
    int32_t maphead[5] = { map_size1, map_size2, matched0, (int32_t)appended, (int32_t)refreshed };
    fwrite(head, sizeof(head), 1, f);
    fwrite(maphead, sizeof(maphead), 1, f);
    fwrite(s.frame_offset, sizeof(int32_t), (size_t)n + 1, f);
    fwrite(s.lines, sizeof(float) * 4, (size_t)total, f);
    fwrite(s.ground, sizeof(double) * 4, (size_t)total, f);
    fwrite(s.keep, 1, (size_t)total, f);
    fwrite(s.code, 32, (size_t)total, f);
    fwrite(idx, sizeof(int32_t), (size_t)total, f);
    fwrite(dist, sizeof(float), (size_t)total, f);
    fwrite(boff, sizeof(int64_t), (size_t)n + 1, f);
    fwrite(body, 1, (size_t)boff[n], f);
    int32_t klhead[1] = { n_kl };
    fwrite(klhead, sizeof(klhead), 1, f);
    fwrite(kl.frame_offset, sizeof(int32_t), (size_t)n + 1, f);
    fwrite(kl.in_octave, sizeof(float) * 4, (size_t)n_kl, f);
    fwrite(kl.octave, sizeof(int32_t), (size_t)n_kl, f);
    fwrite(kl.class_id, sizeof(int32_t), (size_t)n_kl, f);
    fwrite(kl.code, 32, (size_t)n_kl, f);
    fclose(f);
    lf_destroy(h);
    printf("abi_client: %d frames, %d segments, %lld body bytes\n", n, total, (long long)boff[n]);
    return 0;
}
