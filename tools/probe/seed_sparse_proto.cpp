// CPU prototype of the sparse introsort chain (data-parallel formulation), checked against the real std::sort.
#include <algorithm>
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace std;

struct normPoint { int x, y, norm; };
static inline bool compare_norm(const normPoint& a, const normPoint& b) { return a.norm > b.norm; }

static inline uint32_t keyof(uint32_t v) { return v >> 20; }
static inline bool comp(uint32_t a, uint32_t b) { return keyof(a) > keyof(b); }

struct DenseRange { int off, len, depth; };

static int g_folds, g_splits, g_inserts, g_maxM, g_chain_end_len, g_chain_end_m;
static int kDenseLen = 1024;

// lower_bound over P[a, b)
static int lb(const vector<uint32_t>& P, int a, int b, uint32_t q) { return (int)(lower_bound(P.begin() + a, P.begin() + b, q) - P.begin()); }

// the chain.  P, V: explicit list sorted by position; n: virtual array length; returns dense array E and ranges
static void sparse_chain(vector<uint32_t> P, vector<uint32_t> V, int n, int nseeds, vector<uint32_t>& E, vector<DenseRange>& ranges)
{
    int f = 0, l = n, depth = 0;
    { int nn = n, lg = 0; while (nn > 1) { nn >>= 1; ++lg; } depth = 2 * lg; }
    int seeds = nseeds;
    E.clear(); ranges.clear();
    auto materialize = [&](int f, int l, const vector<uint32_t>& P, const vector<uint32_t>& V, int a, int b, int depth) {
        const int off = (int)E.size();
        E.resize(off + (l - f), 0u);
        for (int i = a; i < b; ++i) { assert(P[i] >= (uint32_t)f && P[i] < (uint32_t)l); E[off + P[i] - f] = V[i]; }
        ranges.push_back({off, l - f, depth});
    };
    int a = 0, b = (int)P.size();
    for (;;) {
        const int len = l - f, m = b - a;
        if (seeds == 0) break;
        const int zeros = len - m;
        if (len <= kDenseLen || zeros * 4 <= len || depth == 0) { g_chain_end_len = len; g_chain_end_m = m; materialize(f, l, P, V, a, b, depth); break; }
        --depth;
        // median of three
        auto val = [&](int pos, int* idx) { int i = lb(P, a, b, (uint32_t)pos); if (i < b && P[i] == (uint32_t)pos) { *idx = i; return V[i]; } *idx = -1; return 0u; };
        const int ia = f + 1, ib = f + len / 2, ic = l - 1;
        int xa, xb, xc, xf;
        const uint32_t va = val(ia, &xa), vb = val(ib, &xb), vc = val(ic, &xc), vf = val(f, &xf);
        int pick, xp; uint32_t pv;
        if (comp(va, vb)) { if (comp(vb, vc)) { pick = ib; xp = xb; pv = vb; } else if (comp(va, vc)) { pick = ic; xp = xc; pv = vc; } else { pick = ia; xp = xa; pv = va; } }
        else { if (comp(va, vc)) { pick = ia; xp = xa; pv = va; } else if (comp(vb, vc)) { pick = ic; xp = xc; pv = vc; } else { pick = ib; xp = xb; pv = vb; } }
        // pv may be an explicit zero?  no: explicit entries all have key > 0
        // list edit
        if (xf >= 0) { assert(xf == a); }
        if (xp >= 0) {                               // pivot explicit (key > 0): its slot takes vf (or dies)
            V[xp] = vf;                              // vf == 0: dead entry (skipped below)
            if (xf >= 0) ++a;
        } else {                                     // pivot is an implicit zero
            if (vf != 0u) {                          // the entry at f moves to `pick`
                ++g_inserts;
                const int ins = lb(P, a, b, (uint32_t)pick);           // entries [a+1, ins) shift left by one
                for (int i = a + 1; i < ins; ++i) { P[i - 1] = P[i]; V[i - 1] = V[i]; }
                P[ins - 1] = (uint32_t)pick; V[ins - 1] = vf;
            }
        }
        const int lo = f + 1, hi = l;
        if (keyof(pv) == 0) {
            // ---------------- FOLD
            ++g_folds;
            assert(pv == 0u);
            const int mm = b - a;
            for (int i = a; i < b; ++i) assert(V[i] != 0u && P[i] >= (uint32_t)lo && P[i] < (uint32_t)hi);
            const long long Zt = (long long)(hi - lo) - mm;
            // K: entry-parallel.  e = number of entries inside the last k positions
            auto kstart = [&](int e) -> long long { return e == 0 ? 0 : (long long)hi - P[b - e]; };
            long long K = -1; int eK = -1;
            for (int e = 0; e <= mm; ++e) {
                const bool feas = 2 * kstart(e) - e <= Zt;
                const bool feas_next = e < mm && 2 * kstart(e + 1) - (e + 1) <= Zt;
                if (feas && !feas_next) {
                    long long kmax = (Zt + e) / 2;
                    if (e < mm) kmax = min(kmax, kstart(e + 1) - 1);
                    kmax = min<long long>(kmax, hi - lo);
                    K = kmax; eK = e;
                }
            }
            assert(K >= 0);
            // moved entries: the last eK ones.  stationary: [a, b - eK)
            const int ns = mm - eK;
            auto g = [&](int j) -> long long { return (long long)P[a + j] - lo - j; };   // zeros in front of stationary entry j
            auto cOf = [&](long long k) { int x = 0, y = ns; while (x < y) { int mid = (x + y) / 2; if (g(mid) < k) x = mid + 1; else y = mid; } return x; };
            vector<uint32_t> P2(mm), V2(mm);
            for (int e = 1; e <= eK; ++e) {
                const int i = b - e;
                const long long k = (long long)hi - P[i];
                assert(k >= 1 && k <= K);
                const int c = cOf(k);
                const long long newpos = lo + (k - 1) + c;
                const int ni = c + (e - 1);
                P2[ni] = (uint32_t)newpos; V2[ni] = V[i];
            }
            for (int j = 0; j < ns; ++j) {
                const long long z = min<long long>(g(j), K);            // moved entries with k <= z
                const int cnt = b - lb(P, a, b, (uint32_t)max<long long>(hi - z, 0)); // entries with pos >= hi - z
                const int ni = j + cnt;
                P2[ni] = P[a + j]; V2[ni] = V[a + j];
            }
            // cut
            long long cut;
            auto Lk = [&](long long k) { return (long long)lo + (k - 1) + cOf(k); };
            // note: cOf over stationary entries only; for k = K + 1 the (K+1)-th zero lies at or after R_K - 1 ... entries between? use full list
            auto cOfFull = [&](long long k) { int x = 0, y = mm; while (x < y) { int mid = (x + y) / 2; if ((long long)P[a + mid] - lo - mid < k) x = mid + 1; else y = mid; } return x; };
            auto LkFull = [&](long long k) { return (long long)lo + (k - 1) + cOfFull(k); };
            if (K == 0) cut = LkFull(1);
            else { cut = hi - K; if (Zt >= K + 1) cut = min(cut, LkFull(K + 1)); }
            (void)Lk;
            for (int i = 0; i < mm; ++i) { assert(P2[i] < (uint32_t)cut); if (i) assert(P2[i - 1] < P2[i]); }
            for (int i = 0; i < mm; ++i) { P[a + i] = P2[i]; V[a + i] = V2[i]; }
            l = (int)cut;
        } else {
            // ---------------- SPLIT
            ++g_splits;
            const uint32_t kp = keyof(pv);
            const int mm = b - a;
            vector<int> prefP(mm + 1, 0), prefG(mm + 1, 0);
            for (int i = 0; i < mm; ++i) {
                const uint32_t v = V[a + i];
                prefP[i + 1] = prefP[i] + (v != 0u && keyof(v) > kp);
                prefG[i + 1] = prefG[i] + (v != 0u && keyof(v) >= kp);
            }
            const int nP = prefP[mm], nG = prefG[mm];
            vector<uint32_t> PP(nP); vector<int> GR(nG);          // P positions ascending; G entry indices by rank from the right
            for (int i = 0; i < mm; ++i) {
                const uint32_t v = V[a + i];
                if (v != 0u && keyof(v) > kp) PP[prefP[i]] = P[a + i];
                if (v != 0u && keyof(v) >= kp) GR[nG - 1 - prefG[i]] = i;
            }
            auto cP = [&](long long k) { int x = 0, y = nP; while (x < y) { int mid = (x + y) / 2; if ((long long)PP[mid] - lo - mid < k) x = mid + 1; else y = mid; } return x; };
            const long long nonP = (long long)(hi - lo) - nP;
            auto Lk = [&](long long k) -> long long { return k <= nonP ? (long long)lo + (k - 1) + cP(k) : (long long)1 << 40; };
            int K = 0;
            for (int k = 1; k <= nG; ++k) if (Lk(k) < (long long)P[a + GR[k - 1]]) ++K; else break;
            for (int k = K + 1; k <= nG; ++k) assert(!(Lk(k) < (long long)P[a + GR[k - 1]]));
            vector<int> T(K + 1, -1);
            for (int i = 0; i < mm; ++i) {
                const uint32_t v = V[a + i];
                if (v != 0u && keyof(v) <= kp) {
                    const long long lr = (long long)P[a + i] - lo - prefP[i] + 1;
                    if (lr <= K) T[lr] = i;
                }
            }
            long long cut;
            if (K == 0) cut = Lk(1); else cut = min<long long>(Lk(K + 1), P[a + GR[K - 1]]);
            assert(cut > f && cut <= hi);
            // left part, dense
            const int off = (int)E.size();
            E.resize(off + (cut - f), 0xffffffffu);
            E[off] = pv;
            int lseeds = (pv & 0xfffffu) != 0u;
            vector<uint32_t> P2, V2;
            for (int i = 0; i < mm; ++i) {
                const uint32_t v = V[a + i];
                if (v == 0u) continue;
                const uint32_t key = keyof(v);
                const bool isG = key >= kp, isL = key <= kp;
                const int grank = isG ? nG - prefG[i] : 0;                 // from the right, 1-based
                const long long lrank = isL ? (long long)P[a + i] - lo - prefP[i] + 1 : 0;
                const bool swG = isG && grank <= K, swL = isL && lrank <= K;
                assert(!(swG && swL));
                if (swG) {
                    const long long np = Lk(grank);
                    assert(np < cut);
                    assert(E[off + np - f] == 0xffffffffu);
                    E[off + np - f] = v; lseeds += (v & 0xfffffu) != 0u;
                    if (T[grank] >= 0) { P2.push_back(P[a + i]); V2.push_back(V[a + T[grank]]); }
                } else if (swL) {
                    // goes to R_lrank: emitted by that G entry
                } else if ((long long)P[a + i] < cut) {
                    assert(isG);
                    assert(E[off + P[a + i] - f] == 0xffffffffu);
                    E[off + P[a + i] - f] = v; lseeds += (v & 0xfffffu) != 0u;
                } else { assert(!(key > kp)); P2.push_back(P[a + i]); V2.push_back(v); }
            }
            for (long long q = f; q < cut; ++q) assert(E[off + q - f] != 0xffffffffu);
            ranges.push_back({off, (int)(cut - f), depth});
            seeds -= lseeds;
            for (size_t i = 1; i < P2.size(); ++i) assert(P2[i - 1] < P2[i]);
            for (size_t i = 0; i < P2.size(); ++i) { assert(P2[i] >= (uint32_t)cut); P[a + i] = P2[i]; V[a + i] = V2[i]; }
            b = a + (int)P2.size();
            f = (int)cut;
        }
    }
    g_maxM = max(g_maxM, (int)E.size());
}

// replay on a key array; compare with std::sort.  returns true on equal order of "seed" elements
static bool check(const vector<int>& keys, const vector<uint8_t>& isseed, bool verbose)
{
    const int n = (int)keys.size();
    vector<normPoint> pts(n);
    for (int i = 0; i < n; ++i) pts[i] = {i, 0, keys[i]};
    sort(pts.begin(), pts.end(), compare_norm);
    vector<int> want;
    for (int i = 0; i < n; ++i) if (isseed[pts[i].x]) want.push_back(pts[i].x);
    // sparse list
    vector<uint32_t> P, V; int ns = 0;
    for (int i = 0; i < n; ++i) if (keys[i] > 0) { P.push_back(i); V.push_back(((uint32_t)keys[i] << 20) | (isseed[i] ? (uint32_t)(ns + 1) : 0u)); if (isseed[i]) ++ns; }
    vector<int> seedpos; for (int i = 0; i < n; ++i) if (isseed[i]) seedpos.push_back(i);
    vector<uint32_t> E; vector<DenseRange> R;
    g_folds = g_splits = g_inserts = 0; g_chain_end_len = g_chain_end_m = 0;
    sparse_chain(P, V, n, ns, E, R);
    // dense phase with the real library loop
    auto cmpv = [](uint32_t x, uint32_t y) { return (x >> 20) > (y >> 20); };
    for (auto& r : R) std::__introsort_loop(E.begin() + r.off, E.begin() + r.off + r.len, (long)r.depth, __gnu_cxx::__ops::__iter_comp_iter(cmpv));
    // final insertion sort == stable sort by key descending
    vector<uint32_t> S;
    for (uint32_t v : E) if (v & 0xfffffu) S.push_back(v);
    stable_sort(S.begin(), S.end(), cmpv);
    bool ok = S.size() == want.size();
    if (ok) for (size_t i = 0; i < S.size(); ++i) if (seedpos[(S[i] & 0xfffffu) - 1] != want[i]) { ok = false; break; }
    if (verbose || !ok) printf("n %d explicit %zu seeds %d | folds %d splits %d inserts %d | dense total %zu in %zu ranges, chain end len %d m %d | %s\n", n, P.size(), ns, g_folds, g_splits, g_inserts, E.size(), R.size(), g_chain_end_len, g_chain_end_m, ok ? "OK" : "MISMATCH");
    return ok;
}

int main(int argc, char** argv)
{
    bool allok = true;
    for (int ai = 1; ai < argc; ++ai) {
        FILE* fp = fopen(argv[ai], "rb");
        if (!fp) { perror(argv[ai]); return 1; }
        int32_t np; if (fread(&np, 4, 1, fp) != 1) return 1;
        printf("%s: %d problems\n", argv[ai], np);
        for (int p = 0; p < np; ++p) {
            int32_t hw[2]; if (fread(hw, 4, 2, fp) != 2) return 1;
            const int n = hw[0] * hw[1];
            vector<int32_t> b(n); vector<uint8_t> d(n);
            if (fread(b.data(), 4, n, fp) != (size_t)n) return 1;
            if (fread(d.data(), 1, n, fp) != (size_t)n) return 1;
            vector<int> keys(b.begin(), b.end());
            allok &= check(keys, d, true);
        }
        fclose(fp);
    }
    // random tests
    srand(7);
    int cases = 0;
    for (int it = 0; it < 3000; ++it) {
        const int n = 17 + rand() % (it % 10 == 0 ? 40000 : 3000);
        const int dens = rand() % 100, nk = 1 + rand() % (rand() % 2 ? 4 : 1023);
        vector<int> keys(n, 0); vector<uint8_t> sd(n, 0);
        const int thr = rand() % (nk + 1);
        for (int i = 0; i < n; ++i) if (rand() % 100 < dens) { keys[i] = 1 + rand() % nk; sd[i] = keys[i] > thr; }
        const int style = rand() % 4;
        if (style == 1) { for (int i = 0; i < n && i < 5; ++i) { keys[i] = 1 + rand() % nk; sd[i] = 1; } }
        if (style == 2) { keys[n / 2] = 1 + rand() % nk; keys[n - 1] = 1 + rand() % nk; keys[1] = 1 + rand() % nk; }
        if (style == 3) { for (int i = 0; i < n; ++i) if (keys[i]) keys[i] = 1 + (i * nk / n); }
        kDenseLen = (rand() % 2) ? 1024 : 20;
        allok &= check(keys, sd, false);
        ++cases;
    }
    printf("%d random cases, %s\n", cases, allok ? "ALL OK" : "FAILURES");
    return allok ? 0 : 1;
}
