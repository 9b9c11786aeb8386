// TEST HARNESS ONLY -- not part of the product.
// Compiles the device-side LSD grow logic (lane_slam_amd/csrc/lsd_grow.h) for the host with
// one lane (LF_HOST_SIM) so its control flow and arithmetic can be compared with the CPU
// oracle in this GPU-less container.  The gradient/angle stage below mirrors
// lane_slam_amd/csrc/k_lsd_grad.hip's last phase; ordering mirrors k_lsd_order.hip's key.
#define LF_HOST_SIM 1
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "../../lane_slam_amd/csrc/lsd_grow.h"

using namespace lf;

// by_components != 0: the problem is split into the connected components of its defined pixels (8-adjacency, host
// BFS here; k_lsd_label on the device), every component of at least min_reg_size pixels is grown on its own --
// largest first, as k_lsd_grow hands them out -- and the lines are put back in seed order by their tags.
extern "C" int hs_lsd_detect_ex(const double* scaled, int H, int W, double rho, double prec, double p, double log_nt,
                                double log_eps, double density_th, double scale, int min_reg_size, int refine,
                                int n_bins, float* lines, int cap, int reg_lds, int by_components, int* n_components)
{
    const size_t Ps = (size_t)H * W;
    std::vector<float> ang(Ps, grow::NOTDEF_F);
    std::vector<double> mod(Ps, 0.0), cs(Ps, 0.0), sn(Ps, 0.0);
    double max_grad = -1;
    for (int y = 0; y < H - 1; ++y)
        for (int x = 0; x < W - 1; ++x) {
            size_t a = (size_t)y * W + x;
            double DA = scaled[a + W + 1] - scaled[a];
            double BC = scaled[a + 1] - scaled[a + W];
            double gx = DA + BC, gy = DA - BC;
            double norm = dm::dsqrt((gx * gx + gy * gy) / 4);
            mod[a] = norm;
            if (!(norm <= rho)) {
                float av = dm::fast_atan2_deg((float)gx, (float)(-gy));
                ang[a] = av;
                double arad = (double)av * grow::DEG2RAD;
                dm::dsincos((double)(float)arad, sn[a], cs[a]);
                if (norm > max_grad) max_grad = norm;
            }
        }
    const double bin_coef = (max_grad > 0) ? (double)(n_bins - 1) / max_grad : 0;
    // compact arrays in raster order, as k_lsd_order leaves them
    std::vector<uint32_t> gxy;
    std::vector<uint16_t> xs;
    std::vector<float> deg;
    std::vector<double> modc, csc, snc;
    std::vector<int> rows(H + 1, 0);
    std::vector<uint32_t> order;
    for (int y = 0; y < H; ++y) {
        rows[y] = (int)gxy.size();
        for (int x = 0; x < W; ++x) {
            size_t a = (size_t)y * W + x;
            if (ang[a] == grow::NOTDEF_F) continue;
            int bin = (int)(mod[a] * bin_coef);
            order.push_back(((uint32_t)((n_bins - 1) - bin) << 20) | (uint32_t)gxy.size());
            gxy.push_back(((uint32_t)y << 16) | (uint32_t)x);
            xs.push_back((uint16_t)x);
            deg.push_back(ang[a]); modc.push_back(mod[a]); csc.push_back(cs[a]); snc.push_back(sn[a]);
        }
    }
    rows[H] = (int)gxy.size();
    std::stable_sort(order.begin(), order.end(), [](uint32_t u, uint32_t v) { return (u >> 20) < (v >> 20); });
    const int n_def = (int)gxy.size();
    // reg_lds doubles as the split point of every LDS/HBM-backed structure in this harness
    std::vector<uint32_t> usedc((n_def + 63) / 32 + 1, 0u), gused((n_def + 63) / 32 + 1, 0u);
    std::vector<uint32_t> lreg(reg_lds > 0 ? reg_lds : 1), greg(Ps);
    grow::Ctx c;
    c.q = nullptr;
    c.W = W; c.H = H;
    c.rows = rows.data(); c.lxs = xs.data(); c.gxy = gxy.data(); c.def_lds = reg_lds < n_def ? reg_lds : n_def;
    c.deg = deg.data(); c.mod = modc.data(); c.cs = csc.data(); c.sn = snc.data();
    c.usedc = usedc.data(); c.gused = gused.data(); c.used_lds = (reg_lds / 32) * 32;
    c.lreg = lreg.data(); c.greg = greg.data(); c.reg_lds = reg_lds;
    c.log_nt = log_nt; c.log_eps = log_eps; c.density_th = density_th; c.prec = prec; c.p = p; c.scale = scale;
    c.min_reg_size = min_reg_size; c.refine = refine;
    c.label = nullptr; c.root = 0; c.tags = nullptr; c.line_count = nullptr;
    if (n_components) *n_components = 0;
    if (!by_components) return grow::detect(c, order.data(), (int)order.size(), lines, cap);
    // connected components: label = first entry (raster order) of the component
    std::vector<uint16_t> label(n_def, 0xffff);
    std::vector<std::pair<int, int>> comps;      // (size, root)
    auto entry_at = [&](int x, int y) -> int {
        if (x < 0 || x >= W || y < 0 || y >= H) return -1;
        for (int e = rows[y]; e < rows[y + 1]; ++e) if ((int)xs[e] == x) return e;
        return -1;
    };
    for (int e0 = 0; e0 < n_def; ++e0) {
        if (label[e0] != 0xffff) continue;
        std::vector<int> stack(1, e0);
        label[e0] = (uint16_t)e0;
        int size = 0;
        while (!stack.empty()) {
            const int e = stack.back(); stack.pop_back(); ++size;
            const int x = (int)(gxy[e] & 0xffffu), y = (int)(gxy[e] >> 16);
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int q = entry_at(x + dx, y + dy);
                    if (q >= 0 && label[q] == 0xffff) { label[q] = (uint16_t)e0; stack.push_back(q); }
                }
        }
        if (size >= (min_reg_size > 1 ? min_reg_size : 1)) comps.push_back(std::make_pair(size, e0));
    }
    std::sort(comps.begin(), comps.end(), [](const std::pair<int, int>& u, const std::pair<int, int>& v) {
        return u.first != v.first ? u.first > v.first : u.second < v.second; });
    if (n_components) *n_components = (int)comps.size();
    std::vector<float> tl((size_t)cap * 4);
    std::vector<int> tags(cap);
    int count = 0;
    c.label = label.data(); c.tags = tags.data(); c.line_count = &count;
    for (size_t k = 0; k < comps.size(); ++k) {
        c.root = comps[k].second;
        (void)grow::detect(c, order.data(), (int)order.size(), tl.data(), cap);
    }
    const int n = count < cap ? count : cap;
    for (int i = 0; i < n; ++i) {
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += tags[j] < tags[i] ? 1 : 0;
        for (int k = 0; k < 4; ++k) lines[4 * rank + k] = tl[4 * (size_t)i + k];
    }
    return count;
}

extern "C" int hs_lsd_detect(const double* scaled, int H, int W, double rho, double prec, double p, double log_nt,
                             double log_eps, double density_th, double scale, int min_reg_size, int refine,
                             int n_bins, float* lines, int cap, int reg_lds)
{
    return hs_lsd_detect_ex(scaled, H, W, rho, prec, p, log_nt, log_eps, density_th, scale, min_reg_size, refine, n_bins, lines,
                            cap, reg_lds, 0, nullptr);
}

// host parameters exactly as the product computes them (lanefront_api.hip make_lsd_params)
extern "C" void hs_lsd_params(double ang_th, double quant, int Hs, int Ws, double* rho, double* prec, double* p,
                              double* log_nt, int* min_reg_size)
{
    *prec = 3.14159265358979323846 * ang_th / 180;
    *p = ang_th / 180;
    *rho = quant / dm::dsin(*prec);
    *log_nt = 5 * (dm::dlog10((double)Ws) + dm::dlog10((double)Hs)) / 2 + dm::dlog10(11.0);
    *min_reg_size = (int)(-*log_nt / dm::dlog10(*p));
}

extern "C" void hs_detmath(int which, const double* a, const double* b, double* y, int n)
{
    for (int i = 0; i < n; ++i) {
        switch (which) {
        case 0: y[i] = dm::dexp(a[i]); break;
        case 1: y[i] = dm::dlog(a[i]); break;
        case 2: y[i] = dm::dsin(a[i]); break;
        case 3: y[i] = dm::dcos(a[i]); break;
        case 4: y[i] = dm::datan(a[i]); break;
        case 5: y[i] = dm::dasin(a[i]); break;
        case 6: y[i] = dm::dlog10(a[i]); break;
        case 7: y[i] = dm::dsinh_small(a[i]); break;
        case 8: y[i] = dm::datan2(a[i], b[i]); break;
        case 9: y[i] = dm::dpow(a[i], b[i]); break;
        }
    }
}
