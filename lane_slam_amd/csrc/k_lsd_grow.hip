// K_lsd_grow: launches lsd_grow.h -- one 64-lane workgroup per (frame, colour) LSD problem.
// See lsd_grow.h for the algorithm, the reference citation and the lane-cooperation scheme.
#include "common.h"
#include "lsd_grow.h"

namespace lf {

__global__ __launch_bounds__(64) void k_lsd_grow(LsdParams p, const float* __restrict__ ang,
                                                 const double* __restrict__ mod, const double* __restrict__ cs,
                                                 const double* __restrict__ sn,
                                                 const uint32_t* __restrict__ order,
                                                 const int* __restrict__ norder, const uint2* __restrict__ deflist,
                                                 const int* __restrict__ row_start, uint32_t* reg, float* lines,
                                                 int* counts, int reg_lds, int def_lds)
{
    extern __shared__ uint32_t lds[];
    const int pc = blockIdx.x;
    const int lane = threadIdx.x;
    const size_t Ps = (size_t)p.Hs * p.Ws;
    const int nwords = (int)((Ps + 31) / 32);
    // LDS carve: [defined-pixel list (8 B entries)] [USED bitmap] [row starts] [region list]
    uint2* ldef = reinterpret_cast<uint2*>(lds);
    uint32_t* used = lds + 2 * def_lds;
    int* rows = reinterpret_cast<int*>(used + ((nwords + 1) & ~1));
    uint32_t* lreg = reinterpret_cast<uint32_t*>(rows + ((p.Hs + 2) & ~1));
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
    const int n_def = norder[pc];
    const uint2* gdef = deflist + (size_t)pc * Ps;
    const int* grs = row_start + (size_t)pc * (p.Hs + 1);
    for (int i = lane; i <= p.Hs; i += 64) rows[i] = grs[i];
    for (int i = lane; i < n_def && i < def_lds; i += 64) ldef[i] = gdef[i];
    __syncthreads();
    grow::Ctx c;
    c.W = p.Ws; c.H = p.Hs;
    c.ang = a;
    c.mod = mod + (size_t)pc * Ps;
    c.cs = cs + (size_t)pc * Ps;
    c.sn = sn + (size_t)pc * Ps;
    c.used = used; c.lreg = lreg; c.greg = reg + (size_t)pc * Ps; c.reg_lds = reg_lds;
    c.rows = rows; c.ldef = ldef; c.gdef = gdef; c.def_lds = def_lds;
    c.log_nt = p.log_nt; c.log_eps = p.log_eps; c.density_th = p.density_th;
    c.prec = p.prec; c.p = p.p; c.scale = p.scaled ? p.scale : 1.0;
    c.min_reg_size = p.min_reg_size; c.refine = p.refine;
#ifdef LFG_STAMPS
    for (int k = 0; k < 8; ++k) c.stamps[k] = 0;
    unsigned long long tb0 = __builtin_readcyclecounter();
#endif
    int n = grow::detect(c, order + (size_t)pc * Ps, n_def, lines + (size_t)pc * p.cap_lines * 4, p.cap_lines);
    if (lane == 0) counts[pc] = n;     // may exceed cap_lines: the host reports LF_ERR_CAPACITY
#ifdef LFG_STAMPS
    if (lane == 0) {
        // diagnostic: park the phase totals in the (otherwise unused) tail of this problem's region scratch
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(reg + (size_t)pc * Ps + Ps - 32);
        for (int k = 0; k < 8; ++k) dbg[k] = c.stamps[k];
        dbg[8] = __builtin_readcyclecounter() - tb0;
        dbg[9] = (unsigned long long)norder[pc];
        dbg[10] = (unsigned long long)n;
    }
#endif
}

void launch_lsd_grow(const LsdParams& p, int n_frames, const float* ang, const double* mod, const double* cs,
                     const double* sn, const uint32_t* order, const int* norder,
                     const uint2* deflist,
                     const int* row_start, uint32_t* reg, float* lines, int* counts, hipStream_t s)
{
    const size_t Ps = (size_t)p.Hs * p.Ws;
    const int nwords = (int)((Ps + 31) / 32);
    // LDS budget per workgroup; fixed parts first, then the region
    // list (2048 points; longer regions spill to HBM) and whatever is left for the pixel list
    const size_t fixed = (size_t)(((nwords + 1) & ~1) + ((p.Hs + 2) & ~1)) * 4;
    // 52 KB -> 3 problems per CU (160 KB LDS): the 768 problems of a 256-frame batch are all resident.
    // Images whose USED bitmap alone exceeds that (1080p: 110 KB) take a whole CU's LDS per problem.
    size_t budget = 52 * 1024;
    if (fixed + 2048 * 4 + 1024 * 8 > budget) budget = 156 * 1024;
    int reg_lds = 2048;
    while (fixed + (size_t)reg_lds * 4 + 1024 * 8 > budget && reg_lds > 64) reg_lds /= 2;
    long long left = (long long)budget - (long long)fixed - (long long)reg_lds * 4;
    int def_lds = left > 0 ? (int)(left / 8) : 0;
    if ((size_t)def_lds > Ps) def_lds = (int)Ps;
    const size_t lds = fixed + (size_t)reg_lds * 4 + (size_t)def_lds * 8;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_lsd_grow), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_lsd_grow, dim3(n_frames * 3), dim3(64), lds, s, p, ang, mod, cs, sn, order, norder,
                       deflist,
                       row_start, reg, lines, counts, reg_lds, def_lds);
}

}  // namespace lf
