// K_lsd_grow: launches lsd_grow.h -- one 64-lane workgroup per (frame, colour) LSD problem.
// See lsd_grow.h for the algorithm, the reference citation and the lane-cooperation scheme.
#include "common.h"
#include "lsd_grow.h"

namespace lf {

__global__ __launch_bounds__(64) void k_lsd_grow(LsdParams p, const float* __restrict__ ang,
                                                 const double* __restrict__ mod, const double* __restrict__ cs,
                                                 const double* __restrict__ sn, const uint32_t* __restrict__ order,
                                                 const int* __restrict__ norder, uint32_t* reg, float* lines,
                                                 int* counts, int reg_lds)
{
    extern __shared__ uint32_t lds[];
    const int pc = blockIdx.x;
    const int lane = threadIdx.x;
    const size_t Ps = (size_t)p.Hs * p.Ws;
    const int nwords = (int)((Ps + 31) / 32);
    uint32_t* used = lds;
    uint32_t* lreg = lds + ((nwords + 1) & ~1);
    const float* a = ang + (size_t)pc * Ps;
    // USED-or-NOTDEF bitmap: one ballot per 64 pixels
    for (size_t base = 0; base < Ps; base += 64) {
        size_t i = base + lane;
        bool nd = i >= Ps || a[i] == kNotDef;
        unsigned long long b = __ballot(nd);
        if (lane == 0) {
            used[base >> 5] = (uint32_t)b;
            if ((base >> 5) + 1 < (size_t)nwords) used[(base >> 5) + 1] = (uint32_t)(b >> 32);
        }
    }
    __syncthreads();
    grow::Ctx c;
    c.W = p.Ws; c.H = p.Hs;
    c.ang = a;
    c.mod = mod + (size_t)pc * Ps;
    c.cs = cs + (size_t)pc * Ps;
    c.sn = sn + (size_t)pc * Ps;
    c.used = used; c.lreg = lreg; c.greg = reg + (size_t)pc * Ps; c.reg_lds = reg_lds;
    c.log_nt = p.log_nt; c.log_eps = p.log_eps; c.density_th = p.density_th;
    c.prec = p.prec; c.p = p.p; c.scale = p.scaled ? p.scale : 1.0;
    c.min_reg_size = p.min_reg_size; c.refine = p.refine;
    int n = grow::detect(c, order + (size_t)pc * Ps, norder[pc], lines + (size_t)pc * p.cap_lines * 4, p.cap_lines);
    if (lane == 0) counts[pc] = n;     // may exceed cap_lines: the host reports LF_ERR_CAPACITY
}

void launch_lsd_grow(const LsdParams& p, int n_frames, const float* ang, const double* mod, const double* cs,
                     const double* sn, const uint32_t* order, const int* norder, uint32_t* reg,
                     float* lines, int* counts, hipStream_t s)
{
    const size_t Ps = (size_t)p.Hs * p.Ws;
    const int nwords = (int)((Ps + 31) / 32);
    int reg_lds = 8192;
    size_t lds = (size_t)(((nwords + 1) & ~1) + reg_lds) * sizeof(uint32_t);
    while (lds > 60 * 1024 && reg_lds > 256) { reg_lds /= 2; lds = (size_t)(((nwords + 1) & ~1) + reg_lds) * sizeof(uint32_t); }
    hipLaunchKernelGGL(k_lsd_grow, dim3(n_frames * 3), dim3(64), lds, s, p, ang, mod, cs, sn, order, norder, reg,
                       lines, counts, reg_lds);
}

}  // namespace lf
