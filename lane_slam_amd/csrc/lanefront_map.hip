// lanefront C ABI, live map + associator part (include/lanefront.h "live map"): host-side sequencing of
// k_assoc.hip / k_map.hip on the map's own HIP stream.  Semantics and the block format: k_map.hip.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <new>
#include <vector>
#include "common.h"

using namespace lf;

namespace {
struct Buf { void* p = nullptr; size_t bytes = 0; };
}

struct lf_map {
    lf_map_config cfg;
    int tie_rule = LF_TIE_LOWEST;
    int device = 0;
    hipStream_t stream = nullptr;
    char err[512];
    MapDevice d;
    size_t cap_pad = 0;
    // host mirror of the device state, refreshed behind every update
    int* h_state = nullptr;                  // pinned: [0..15] state, then 2 x u64 totals at +16 ints
    int errors_reported = 0;                 // failing updates (state[8]) the host has already returned an error for
    hipEvent_t ev_state = nullptr, ev_in = nullptr, ev_out = nullptr;
    bool state_pending = false;
    long long rows_in_flight = 0;            // rows handed to updates whose state copy has not been seen yet
    AssocScratch ws;
    Buf act, own_block, pose, q_in, c_in, idx_out, dist_out, seed_code, seed_color, seed_ground, tie_res;
    Buf st_fo, st_code, st_color, st_keep, st_ground, st_idx, st_dist;     // staging of lf_map_step_host
    std::vector<double> h_pose;
    // per-stage timing with HIP events on the map's stream (resolved by lf_map_get_timing)
    struct Ev { hipEvent_t a, b; int st; };
    bool profiling = false;
    std::vector<Ev> ev_free, ev_used;
    double ms[LF_MAP_N_STAGES];
    int32_t launches[LF_MAP_N_STAGES];
};

namespace {
struct MapTimer {
    lf_map* m; int st; lf_map::Ev e; bool on;
    MapTimer(lf_map* m_, int st_) : m(m_), st(st_), on(m_->profiling)
    {
        if (!on) return;
        if (m->ev_free.empty()) {
            lf_map::Ev n; n.st = 0;
            if (m->ev_used.size() >= 4096 || hipEventCreate(&n.a) != hipSuccess || hipEventCreate(&n.b) != hipSuccess) { on = false; return; }
            m->ev_free.push_back(n);
        }
        e = m->ev_free.back(); m->ev_free.pop_back();
        e.st = st;
        (void)hipEventRecord(e.a, m->stream);
    }
    ~MapTimer()
    {
        if (on) { (void)hipEventRecord(e.b, m->stream); m->ev_used.push_back(e); }
        m->launches[st] += 1;
    }
};
}

static char g_map_create_err[512] = "no error";

static void map_error(lf_map* m, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(m ? m->err : g_map_create_err, 512, fmt, ap);
    va_end(ap);
}

#define MAP_HIP(m, expr)                                                                          \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            map_error((m), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return LF_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

static int grow(lf_map* m, Buf& b, size_t bytes)
{
    if (b.bytes >= bytes) return LF_OK;
    if (b.p) { MAP_HIP(m, hipStreamSynchronize(m->stream)); (void)hipFree(b.p); }
    b.p = nullptr; b.bytes = 0;
    const size_t want = bytes + bytes / 4 + 256;
    MAP_HIP(m, hipMalloc(&b.p, want));
    b.bytes = want;
    return LF_OK;
}

// make the host mirror current: wait for the copy queued behind the last update (block = false: only look)
static int refresh_state(lf_map* m, bool block = true)
{
    if (m->state_pending) {
        if (block) MAP_HIP(m, hipEventSynchronize(m->ev_state));
        else if (hipEventQuery(m->ev_state) != hipSuccess) return LF_OK;          // still in flight: the mirror is stale
        m->state_pending = false;
        m->rows_in_flight = 0;
    }
    // a failing update is reported ONCE, by the first call that sees it; the map stays usable (nothing is sticky)
    if (m->h_state[8] != m->errors_reported) {
        const int n_new = m->h_state[8] - m->errors_reported, flags = m->h_state[2];
        m->errors_reported = m->h_state[8];
        if (flags & 2) { map_error(m, "lf_map_update was given a block with a bad header (magic / count): that update was not applied (%d failing update(s) since the last report)", n_new); return LF_ERR_BAD_ARG; }
        if (flags & 4) { map_error(m, "a rank's segments did not fit its block (overflow marker in a gathered header): that step's update was applied on no replica (%d failing update(s) since the last report)", n_new); return LF_ERR_CAPACITY; }
        map_error(m, "the map is full (capacity %d, LF_MAP_FULL_ERROR): segments were dropped (%d failing update(s) since the last report)", m->cfg.capacity, n_new);
        return LF_ERR_CAPACITY;
    }
    return LF_OK;
}

static int queue_state_copy(lf_map* m)
{
    MAP_HIP(m, hipMemcpyAsync(m->h_state, m->d.state, 16 * sizeof(int), hipMemcpyDeviceToHost, m->stream));
    MAP_HIP(m, hipMemcpyAsync(m->h_state + 16, m->d.totals, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, m->stream));
    MAP_HIP(m, hipEventRecord(m->ev_state, m->stream));
    m->state_pending = true;
    return LF_OK;
}

// the map's stream waits for everything queued so far on the handle's stream
static int after_handle(lf_map* m, lf_handle* h)
{
    if (!h) return LF_OK;
    void* hs = nullptr;
    if (lf_get_stream(h, &hs) != LF_OK) { map_error(m, "bad handle"); return LF_ERR_BAD_ARG; }
    MAP_HIP(m, hipEventRecord(m->ev_in, static_cast<hipStream_t>(hs)));
    MAP_HIP(m, hipStreamWaitEvent(m->stream, m->ev_in, 0));
    return LF_OK;
}

// the handle's later work (its next batch overwrites the segment arrays) waits for what the map has queued so far
static int release_handle(lf_map* m, lf_handle* h)
{
    if (!h) return LF_OK;
    void* hs = nullptr;
    if (lf_get_stream(h, &hs) != LF_OK) { map_error(m, "bad handle"); return LF_ERR_BAD_ARG; }
    MAP_HIP(m, hipEventRecord(m->ev_out, m->stream));
    MAP_HIP(m, hipStreamWaitEvent(static_cast<hipStream_t>(hs), m->ev_out, 0));
    return LF_OK;
}

static const char* kMapStageNames[LF_MAP_N_STAGES] = { "assoc_pack_queries", "assoc_mfma", "map_pack_block", "map_update" };   // stage 0 is gone (queries are expanded inside the association kernel): always 0 calls

extern "C" const char* lf_map_stage_name(int stage) { return (stage >= 0 && stage < LF_MAP_N_STAGES) ? kMapStageNames[stage] : "?"; }

extern "C" int lf_map_set_profiling(lf_map* m, int enabled)
{
    if (!m) return LF_ERR_NOT_INITIALISED;
    m->profiling = enabled != 0;
    if (m->profiling) {                       // fill the event pool now, not inside the first profiled steps
        (void)hipSetDevice(m->device);
        while (m->ev_free.size() < 512) {
            lf_map::Ev n; n.st = 0;
            if (hipEventCreate(&n.a) != hipSuccess) break;
            if (hipEventCreate(&n.b) != hipSuccess) { (void)hipEventDestroy(n.a); break; }
            m->ev_free.push_back(n);
        }
    }
    return LF_OK;
}

// ms accumulated and launches counted per stage since the last call; resets both
extern "C" int lf_map_get_timing(lf_map* m, double* ms_per_stage, int32_t* launches_per_stage, int n)
{
    if (!m) return LF_ERR_NOT_INITIALISED;
    for (lf_map::Ev& e : m->ev_used) {
        (void)hipEventSynchronize(e.b);
        float t = 0;
        if (hipEventElapsedTime(&t, e.a, e.b) == hipSuccess) m->ms[e.st] += t;
        m->ev_free.push_back(e);
    }
    m->ev_used.clear();
    for (int i = 0; i < LF_MAP_N_STAGES; ++i) {
        if (i < n && ms_per_stage) ms_per_stage[i] = m->ms[i];
        if (i < n && launches_per_stage) launches_per_stage[i] = m->launches[i];
        m->ms[i] = 0; m->launches[i] = 0;
    }
    return LF_OK;
}

extern "C" const char* lf_map_last_error(const lf_map* m) { return m ? m->err : g_map_create_err; }

extern "C" void lf_map_destroy(lf_map* m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    void* ptrs[] = { m->d.code, m->d.color, m->d.ground, m->d.hits, m->d.last_seen, m->d.winner, m->d.mx, m->d.mcx, m->d.state, m->d.totals };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    assoc_scratch_free(m->ws);
    for (Buf* b : { &m->act, &m->own_block, &m->pose, &m->q_in, &m->c_in, &m->idx_out, &m->dist_out, &m->seed_code, &m->seed_color, &m->seed_ground, &m->tie_res,
                     &m->st_fo, &m->st_code, &m->st_color, &m->st_keep, &m->st_ground, &m->st_idx, &m->st_dist })
        if (b->p) (void)hipFree(b->p);
    if (m->h_state) (void)hipHostFree(m->h_state);
    for (lf_map::Ev& e : m->ev_used) m->ev_free.push_back(e);
    for (lf_map::Ev& e : m->ev_free) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (hipEvent_t e : { m->ev_state, m->ev_in, m->ev_out }) if (e) (void)hipEventDestroy(e);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

extern "C" int lf_map_create(int device_id, const lf_map_config* cfg, lf_map** out)
{
    if (!cfg || !out) { map_error(nullptr, "lf_map_create: null argument"); return LF_ERR_BAD_ARG; }
    *out = nullptr;
    if (cfg->capacity < 64 || cfg->capacity > (1 << 21) || cfg->max_distance < 0 || cfg->max_distance > 128 ||
        (cfg->policy != LF_MAP_APPEND && cfg->policy != LF_MAP_MERGE) || (cfg->when_full != LF_MAP_RING && cfg->when_full != LF_MAP_FULL_ERROR) ||
        (cfg->policy == LF_MAP_MERGE && (cfg->merge_distance < 0 || cfg->merge_distance > cfg->max_distance))) {
        map_error(nullptr, "lf_map_create: bad configuration (capacity %d in [64, 2^21], max_distance %d in [0,128], policy %d, when_full %d, merge_distance %d <= max_distance)",
                  cfg->capacity, cfg->max_distance, cfg->policy, cfg->when_full, cfg->merge_distance);
        return LF_ERR_BAD_ARG;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        map_error(nullptr, "lf_map_create: no HIP device (%s); lanefront has no CPU fallback", e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return LF_ERR_HIP;
    }
    if (device_id < 0 || device_id >= ndev) { map_error(nullptr, "lf_map_create: device %d out of range (%d devices)", device_id, ndev); return LF_ERR_BAD_ARG; }
    lf_map* m = new (std::nothrow) lf_map();
    if (!m) return LF_ERR_HIP;
    m->cfg = *cfg; m->device = device_id; m->err[0] = 0;
    memset(&m->d, 0, sizeof(m->d));
    memset(m->ms, 0, sizeof(m->ms)); memset(m->launches, 0, sizeof(m->launches));
    auto fail = [&](int rc) { snprintf(g_map_create_err, sizeof(g_map_create_err), "%s", m->err); lf_map_destroy(m); return rc; };
    const size_t cap = (size_t)cfg->capacity;
    m->cap_pad = assoc_rows_padded_m(cfg->capacity);
#define CREATE_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { map_error(m, "%s failed: %s", #expr, hipGetErrorString(_e)); return fail(LF_ERR_HIP); } } while (0)
    CREATE_HIP(hipSetDevice(device_id));
    {
        // The map's steps are the one serial chain of a pipelined front end (step k's association needs step k - 1's update): its
        // kernels go to a HIGH-PRIORITY stream, so that their workgroups (76 KB of LDS each) are not the last to find room between the
        // region-growing workgroups of the batches in flight (LF_MAP_PRIORITY=0: a plain stream, for A/B)
        int least = 0, greatest = 0;
        static const bool plain = getenv("LF_MAP_PRIORITY") && atoi(getenv("LF_MAP_PRIORITY")) == 0;
        if (!plain && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
            CREATE_HIP(hipStreamCreateWithPriority(&m->stream, hipStreamNonBlocking, greatest));
        else
            CREATE_HIP(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    }
    CREATE_HIP(hipEventCreateWithFlags(&m->ev_state, hipEventDisableTiming));
    CREATE_HIP(hipEventCreateWithFlags(&m->ev_in, hipEventDisableTiming));
    CREATE_HIP(hipEventCreateWithFlags(&m->ev_out, hipEventDisableTiming));
    CREATE_HIP(hipMalloc((void**)&m->d.code, cap * 32));
    CREATE_HIP(hipMalloc((void**)&m->d.color, cap));
    CREATE_HIP(hipMalloc((void**)&m->d.ground, cap * 4 * sizeof(double)));
    CREATE_HIP(hipMalloc((void**)&m->d.hits, cap * sizeof(int)));
    CREATE_HIP(hipMalloc((void**)&m->d.last_seen, cap * sizeof(int)));
    CREATE_HIP(hipMalloc((void**)&m->d.winner, cap * sizeof(int)));
    CREATE_HIP(hipMalloc((void**)&m->d.mx, m->cap_pad * 256));
    CREATE_HIP(hipMalloc((void**)&m->d.mcx, m->cap_pad * 32));
    CREATE_HIP(hipMalloc((void**)&m->d.state, 16 * sizeof(int)));
    CREATE_HIP(hipMalloc((void**)&m->d.totals, 2 * sizeof(unsigned long long)));
    CREATE_HIP(hipHostMalloc((void**)&m->h_state, 24 * sizeof(int)));
    memset(m->h_state, 0, 24 * sizeof(int));
    // rows beyond the map's size must read as "all zero" operands (distance 128): zero everything once
    CREATE_HIP(hipMemsetAsync(m->d.mx, 0, m->cap_pad * 256, m->stream));
    CREATE_HIP(hipMemsetAsync(m->d.mcx, 0, m->cap_pad * 32, m->stream));
    CREATE_HIP(hipMemsetAsync(m->d.code, 0, cap * 32, m->stream));
    CREATE_HIP(hipMemsetAsync(m->d.color, 0, cap, m->stream));
    CREATE_HIP(hipMemsetAsync(m->d.ground, 0, cap * 4 * sizeof(double), m->stream));
    CREATE_HIP(hipMemsetAsync(m->d.hits, 0, cap * sizeof(int), m->stream));
    CREATE_HIP(hipMemsetAsync(m->d.last_seen, 0, cap * sizeof(int), m->stream));
    CREATE_HIP(hipMemsetAsync(m->d.state, 0, 16 * sizeof(int), m->stream));
    CREATE_HIP(hipMemsetAsync(m->d.totals, 0, 2 * sizeof(unsigned long long), m->stream));
    launch_fill_i32(m->d.winner, cap, -1, m->stream);
    CREATE_HIP(hipGetLastError());
    CREATE_HIP(hipStreamSynchronize(m->stream));
#undef CREATE_HIP
    m->d.capacity = cfg->capacity; m->d.policy = cfg->policy; m->d.kept_only = cfg->kept_only;
    m->d.merge_distance = cfg->merge_distance; m->d.when_full = cfg->when_full;
    m->d.fp4 = getenv("LF_ASSOC_INT8") ? 0 : 1;          // e2m1 operands for the FP4 matrix instruction (int8 rows only for A/B runs)
    // the reference's tie rule is the default (round 5); the int8 A/B kernels have no tie pass: the lowest index there, said once
    m->tie_rule = m->d.fp4 ? LF_TIE_MIHASHER : LF_TIE_LOWEST;
    if (!m->d.fp4) { static bool told = false; if (!told) { told = true; fprintf(stderr, "lanefront: LF_ASSOC_INT8 is set: the live map falls back to LF_TIE_LOWEST (the int8 A/B kernels have no tie pass)\n"); } }
    *out = m;
    return LF_OK;
}

extern "C" int lf_map_get_stream(lf_map* m, void** hip_stream)
{
    if (!m) return LF_ERR_NOT_INITIALISED;
    if (!hip_stream) { map_error(m, "lf_map_get_stream: null argument"); return LF_ERR_BAD_ARG; }
    *hip_stream = static_cast<void*>(m->stream);
    return LF_OK;
}

extern "C" int lf_map_synchronize(lf_map* m)
{
    if (!m) return LF_ERR_NOT_INITIALISED;
    MAP_HIP(m, hipSetDevice(m->device));
    MAP_HIP(m, hipStreamSynchronize(m->stream));
    return LF_OK;
}

// rows_hint: how many segment rows the blocks really hold when the host knows it (-1: assume they are full)
static int update_blocks(lf_map* m, const uint8_t* blocks, int n_blocks, int block_rows, int force_append, long long rows_hint = -1)
{
    int rc;
    const size_t rows = (size_t)n_blocks * (size_t)(block_rows - 1);
    if (rows >= (1u << 30)) { map_error(m, "lf_map_update: too many rows"); return LF_ERR_CAPACITY; }
    if ((rc = grow(m, m->act, (rows + rows / 256 + 2) * sizeof(int))) != LF_OK) return rc;      // actions + per-workgroup append counts (k_map.hip: kMapWg = 256 rows each)
    {
        MapTimer t(m, 3);
        launch_map_update(m->d, blocks, n_blocks, block_rows, force_append, static_cast<int*>(m->act.p), m->stream);
    }
    MAP_HIP(m, hipGetLastError());
    m->rows_in_flight += rows_hint >= 0 ? rows_hint : (long long)rows;
    return queue_state_copy(m);
}

extern "C" int lf_map_update(lf_map* m, const uint8_t* blocks, int n_blocks, int block_rows)
{
    if (!m) return LF_ERR_NOT_INITIALISED;
    if (!blocks || n_blocks < 1 || block_rows < 1) { map_error(m, "lf_map_update: null blocks, n_blocks < 1 or block_rows < 1"); return LF_ERR_BAD_ARG; }
    MAP_HIP(m, hipSetDevice(m->device));
    if (block_rows == 1) return LF_OK;
    return update_blocks(m, blocks, n_blocks, block_rows, 0);
}

extern "C" int lf_map_seed(lf_map* m, const uint8_t* code32, const uint8_t* color, const double* ground4, int n, int on_device)
{
    if (!m) return LF_ERR_NOT_INITIALISED;
    if (n < 0 || (n > 0 && !code32)) { map_error(m, "lf_map_seed: bad argument"); return LF_ERR_BAD_ARG; }
    if (n == 0) return LF_OK;
    MAP_HIP(m, hipSetDevice(m->device));
    int rc;
    const uint8_t *dcode = code32, *dcolor = color;
    const double* dground = ground4;
    if (!on_device) {
        if ((rc = grow(m, m->seed_code, (size_t)n * 32)) != LF_OK) return rc;
        MAP_HIP(m, hipMemcpyAsync(m->seed_code.p, code32, (size_t)n * 32, hipMemcpyHostToDevice, m->stream));
        dcode = static_cast<const uint8_t*>(m->seed_code.p);
        if (color) {
            if ((rc = grow(m, m->seed_color, (size_t)n)) != LF_OK) return rc;
            MAP_HIP(m, hipMemcpyAsync(m->seed_color.p, color, (size_t)n, hipMemcpyHostToDevice, m->stream));
            dcolor = static_cast<const uint8_t*>(m->seed_color.p);
        }
        if (ground4) {
            if ((rc = grow(m, m->seed_ground, (size_t)n * 32)) != LF_OK) return rc;
            MAP_HIP(m, hipMemcpyAsync(m->seed_ground.p, ground4, (size_t)n * 32, hipMemcpyHostToDevice, m->stream));
            dground = static_cast<const double*>(m->seed_ground.p);
        }
    }
    if ((rc = grow(m, m->own_block, (size_t)(n + 1) * LF_BLOCK_ROW_BYTES)) != LF_OK) return rc;
    launch_map_seed_block(n, dcode, dcolor, dground, static_cast<uint8_t*>(m->own_block.p), m->stream);
    rc = update_blocks(m, static_cast<const uint8_t*>(m->own_block.p), 1, n + 1, 1, n);
    if (rc != LF_OK) return rc;
    if (!on_device) MAP_HIP(m, hipStreamSynchronize(m->stream));     // the host arrays may be reused on return
    return LF_OK;
}

extern "C" int lf_map_size(lf_map* m, int* size, int* head, int64_t* total_appended, int64_t* total_refreshed)
{
    if (!m) return LF_ERR_NOT_INITIALISED;
    MAP_HIP(m, hipSetDevice(m->device));
    const int rc = refresh_state(m);
    if (size) *size = m->h_state[0];
    if (head) *head = m->h_state[1];
    const unsigned long long* t = reinterpret_cast<const unsigned long long*>(m->h_state + 16);
    if (total_appended) *total_appended = (int64_t)t[0];
    if (total_refreshed) *total_refreshed = (int64_t)t[1];
    return rc;
}

extern "C" int lf_map_associate(lf_map* m, lf_handle* h, const uint8_t* code32, const uint8_t* color, int n,
                                int32_t* idx, float* dist, int on_device)
{
    if (!m) return LF_ERR_NOT_INITIALISED;
    if (n < 0 || (n > 0 && (!code32 || !idx || !dist))) { map_error(m, "lf_map_associate: bad argument"); return LF_ERR_BAD_ARG; }
    if (m->cfg.color_gating && n > 0 && !color) { map_error(m, "lf_map_associate: colour gating is on, colours are required"); return LF_ERR_BAD_ARG; }
    if (n == 0) return LF_OK;
    MAP_HIP(m, hipSetDevice(m->device));
    int rc;
    // The map's size after the last update is needed to size the grid.  The host does not wait for that update:
    // when its state copy has not landed yet, an upper bound (last size seen + the rows handed over since) sizes
    // the grid and the kernel reads the exact size on the device; rows past it hold all-zero operands, so the
    // result is the same.  (An overflow / bad-block flag is then reported one call later.)
    if ((rc = refresh_state(m, false)) != LF_OK) return rc;
    long long bound = (long long)m->h_state[0] + (m->state_pending ? m->rows_in_flight : 0);
    if (bound > m->cfg.capacity) bound = m->cfg.capacity;
    const int size = (int)bound;
    if ((rc = after_handle(m, h)) != LF_OK) return rc;
    hipStream_t s = m->stream;
    const uint8_t *dq = code32, *dc = color;
    int32_t* didx = idx; float* ddist = dist;
    if (!on_device) {
        if ((rc = grow(m, m->q_in, (size_t)n * 32)) || (rc = grow(m, m->idx_out, (size_t)n * 4)) || (rc = grow(m, m->dist_out, (size_t)n * 4))) return rc;
        MAP_HIP(m, hipMemcpyAsync(m->q_in.p, code32, (size_t)n * 32, hipMemcpyHostToDevice, s));
        dq = static_cast<const uint8_t*>(m->q_in.p);
        if (color) {
            if ((rc = grow(m, m->c_in, (size_t)n)) != LF_OK) return rc;
            MAP_HIP(m, hipMemcpyAsync(m->c_in.p, color, (size_t)n, hipMemcpyHostToDevice, s));
            dc = static_cast<const uint8_t*>(m->c_in.p);
        }
        didx = static_cast<int32_t*>(m->idx_out.p); ddist = static_cast<float*>(m->dist_out.p);
    }
    if (m->tie_rule == LF_TIE_MIHASHER && (rc = grow(m, m->tie_res, (size_t)n * 8)) != LF_OK) return rc;
    if (size == 0) {
        // descriptor matrices cannot be void (binary_descriptor_matcher.cpp:201-205): report "no match"
        launch_assoc_nomatch(n, didx, ddist, s);
    } else {
        {
            // one launch: query operands are expanded in registers, results are written by the last workgroup to arrive
            MapTimer t(m, 1);
            m->ws.tie_res = m->tie_rule == LF_TIE_MIHASHER ? static_cast<unsigned long long*>(m->tie_res.p) : nullptr;
            MAP_HIP(m, launch_assoc_core(dq, m->cfg.color_gating ? dc : nullptr, n, m->d.mx, m->d.mcx, size, m->d.state, m->cfg.color_gating,
                                         m->cfg.max_distance, m->ws, didx, ddist, s));
            if (m->tie_rule == LF_TIE_MIHASHER)       // second pass: among the equally near entries, the one the reference's search meets first
                MAP_HIP(m, launch_assoc_ties(dq, m->cfg.color_gating ? dc : nullptr, n, m->d.mx, m->d.code, m->d.color, size, m->d.state,
                                             m->cfg.color_gating, m->ws, static_cast<unsigned long long*>(m->tie_res.p), didx, ddist, s));
        }
    }
    MAP_HIP(m, hipGetLastError());
    if ((rc = release_handle(m, h)) != LF_OK) return rc;
    if (!on_device) {
        MAP_HIP(m, hipMemcpyAsync(idx, didx, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        MAP_HIP(m, hipMemcpyAsync(dist, ddist, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        MAP_HIP(m, hipStreamSynchronize(s));
    }
    return LF_OK;
}

extern "C" int lf_map_set_tie_rule(lf_map* m, int tie_rule)
{
    if (!m) return LF_ERR_NOT_INITIALISED;
    if (tie_rule != LF_TIE_LOWEST && tie_rule != LF_TIE_MIHASHER) { map_error(m, "lf_map_set_tie_rule: unknown rule"); return LF_ERR_BAD_ARG; }
    if (tie_rule == LF_TIE_MIHASHER && !m->d.fp4) { map_error(m, "LF_ASSOC_INT8 (the int8 A/B kernels) supports LF_TIE_LOWEST only"); return LF_ERR_UNSUPPORTED; }
    m->tie_rule = tie_rule;
    return LF_OK;
}

extern "C" int lf_map_pack_block(lf_map* m, lf_handle* h, const lf_segments* segs, int n, int n_frames, const int32_t* idx,
                                 const float* dist, const double* frame_pose, int step, uint8_t* block, int block_rows)
{
    if (!m) return LF_ERR_NOT_INITIALISED;
    if (!segs || !block || n < 0 || (n > 0 && !segs->code) || (frame_pose && (n_frames < 1 || !segs->frame_offset))) {
        map_error(m, "lf_map_pack_block: bad argument (segs->code is required; frame_pose needs segs->frame_offset and n_frames >= 1)");
        return LF_ERR_BAD_ARG;
    }
    if (block_rows < 1) { map_error(m, "lf_map_pack_block: block_rows < 1"); return LF_ERR_BAD_ARG; }
    MAP_HIP(m, hipSetDevice(m->device));
    int rc;
    if (n + 1 > block_rows) {
        // never truncated: the block becomes a header with count 0 and the OVERFLOW marker (word 4 = n).  A rank of a
        // multi-GPU step still takes part in the all-gather with it, and lf_map_update skips, on every replica alike, an
        // update that contains such a block -- so the failure is collective instead of a hang.
        launch_map_overflow_block(n, n_frames, step, block, m->stream);
        MAP_HIP(m, hipGetLastError());
        map_error(m, "lf_map_pack_block: %d segments do not fit a block of %d rows (header + %d); only a header with the overflow marker was written", n, block_rows, block_rows - 1);
        return LF_ERR_CAPACITY;
    }
    if ((rc = after_handle(m, h)) != LF_OK) return rc;
    const double* dpose = nullptr;
    if (frame_pose) {
        // cos / sin with the library's deterministic routines (detmath.h), the same the oracle uses
        m->h_pose.resize((size_t)n_frames * 4);
        for (int f = 0; f < n_frames; ++f) {
            double sn, cs;
            dm::dsincos(frame_pose[3 * f + 2], sn, cs);
            m->h_pose[4 * f] = frame_pose[3 * f]; m->h_pose[4 * f + 1] = frame_pose[3 * f + 1];
            m->h_pose[4 * f + 2] = cs; m->h_pose[4 * f + 3] = sn;
        }
        if ((rc = grow(m, m->pose, (size_t)n_frames * 4 * sizeof(double))) != LF_OK) return rc;
        MAP_HIP(m, hipMemcpyAsync(m->pose.p, m->h_pose.data(), (size_t)n_frames * 4 * sizeof(double), hipMemcpyHostToDevice, m->stream));
        dpose = static_cast<const double*>(m->pose.p);
    }
    {
        MapTimer t(m, 2);
        launch_map_pack_block(n, n_frames, segs->frame_offset, segs->code, segs->color, segs->keep, segs->ground, idx, dist, dpose, step,
                              block, m->stream);
    }
    MAP_HIP(m, hipGetLastError());
    return release_handle(m, h);
}

extern "C" int lf_map_step(lf_map* m, lf_handle* h, const lf_segments* segs, int n, int n_frames, const double* frame_pose,
                           int step, int32_t* idx, float* dist)
{
    if (!m) return LF_ERR_NOT_INITIALISED;
    if (!segs || n < 0 || (n > 0 && (!idx || !dist))) { map_error(m, "lf_map_step: bad argument"); return LF_ERR_BAD_ARG; }
    int rc;
    if (n > 0 && (rc = lf_map_associate(m, h, segs->code, segs->color, n, idx, dist, 1)) != LF_OK) return rc;
    MAP_HIP(m, hipSetDevice(m->device));
    if ((rc = grow(m, m->own_block, (size_t)(n + 1) * LF_BLOCK_ROW_BYTES)) != LF_OK) return rc;
    if ((rc = lf_map_pack_block(m, h, segs, n, n_frames, idx, dist, frame_pose, step, static_cast<uint8_t*>(m->own_block.p), n + 1)) != LF_OK) return rc;
    if (n == 0) return LF_OK;
    return update_blocks(m, static_cast<const uint8_t*>(m->own_block.p), 1, n + 1, 0, n);
}

// the same with HOST arrays (what lf_process_batch returns with out_on_device = 0): for per-frame callers such as a
// ROS node, and for clients that link nothing but this C ABI
extern "C" int lf_map_step_host(lf_map* m, const lf_segments* segs, int n, int n_frames, const double* frame_pose, int step,
                                int32_t* idx, float* dist)
{
    if (!m) return LF_ERR_NOT_INITIALISED;
    if (!segs || n < 0 || n_frames < 1 || (n > 0 && (!segs->code || !idx || !dist)) || !segs->frame_offset ||
        (m->cfg.color_gating && n > 0 && !segs->color)) {
        map_error(m, "lf_map_step_host: bad argument (frame_offset and code are required, color when gating is on)");
        return LF_ERR_BAD_ARG;
    }
    MAP_HIP(m, hipSetDevice(m->device));
    hipStream_t s = m->stream;
    int rc;
    const size_t c = (size_t)(n > 0 ? n : 1);
    if ((rc = grow(m, m->st_fo, (size_t)(n_frames + 1) * 4)) || (rc = grow(m, m->st_code, c * 32)) || (rc = grow(m, m->st_color, c)) ||
        (rc = grow(m, m->st_keep, c)) || (rc = grow(m, m->st_ground, c * 32)) || (rc = grow(m, m->st_idx, c * 4)) || (rc = grow(m, m->st_dist, c * 4))) return rc;
    lf_segments d;
    memset(&d, 0, sizeof(d));
    MAP_HIP(m, hipMemcpyAsync(m->st_fo.p, segs->frame_offset, (size_t)(n_frames + 1) * 4, hipMemcpyHostToDevice, s));
    d.frame_offset = static_cast<int32_t*>(m->st_fo.p);
    if (n > 0) {
        MAP_HIP(m, hipMemcpyAsync(m->st_code.p, segs->code, (size_t)n * 32, hipMemcpyHostToDevice, s));
        d.code = static_cast<uint8_t*>(m->st_code.p);
        if (segs->color) { MAP_HIP(m, hipMemcpyAsync(m->st_color.p, segs->color, (size_t)n, hipMemcpyHostToDevice, s)); d.color = static_cast<uint8_t*>(m->st_color.p); }
        if (segs->keep) { MAP_HIP(m, hipMemcpyAsync(m->st_keep.p, segs->keep, (size_t)n, hipMemcpyHostToDevice, s)); d.keep = static_cast<uint8_t*>(m->st_keep.p); }
        if (segs->ground) { MAP_HIP(m, hipMemcpyAsync(m->st_ground.p, segs->ground, (size_t)n * 32, hipMemcpyHostToDevice, s)); d.ground = static_cast<double*>(m->st_ground.p); }
    }
    rc = lf_map_step(m, nullptr, &d, n, n_frames, frame_pose, step, static_cast<int32_t*>(m->st_idx.p), static_cast<float*>(m->st_dist.p));
    if (rc != LF_OK) return rc;
    if (n > 0) {
        MAP_HIP(m, hipMemcpyAsync(idx, m->st_idx.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        MAP_HIP(m, hipMemcpyAsync(dist, m->st_dist.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    }
    MAP_HIP(m, hipStreamSynchronize(s));
    return LF_OK;
}

extern "C" int lf_map_fetch(lf_map* m, int first, int n, uint8_t* code32, uint8_t* color, double* ground4, int32_t* hits,
                            int32_t* last_seen)
{
    if (!m) return LF_ERR_NOT_INITIALISED;
    if (first < 0 || n < 0 || (long long)first + n > m->cfg.capacity) { map_error(m, "lf_map_fetch: range [%d, %d) outside the map's capacity %d", first, first + n, m->cfg.capacity); return LF_ERR_BAD_ARG; }
    MAP_HIP(m, hipSetDevice(m->device));
    hipStream_t s = m->stream;
    const size_t f = (size_t)first, c = (size_t)n;
    if (n > 0) {
        if (code32) MAP_HIP(m, hipMemcpyAsync(code32, m->d.code + f * 32, c * 32, hipMemcpyDeviceToHost, s));
        if (color) MAP_HIP(m, hipMemcpyAsync(color, m->d.color + f, c, hipMemcpyDeviceToHost, s));
        if (ground4) MAP_HIP(m, hipMemcpyAsync(ground4, m->d.ground + f * 4, c * 32, hipMemcpyDeviceToHost, s));
        if (hits) MAP_HIP(m, hipMemcpyAsync(hits, m->d.hits + f, c * 4, hipMemcpyDeviceToHost, s));
        if (last_seen) MAP_HIP(m, hipMemcpyAsync(last_seen, m->d.last_seen + f, c * 4, hipMemcpyDeviceToHost, s));
    }
    MAP_HIP(m, hipStreamSynchronize(s));
    return LF_OK;
}
