#!/usr/bin/env python3
"""Generates lane_slam_amd/csrc/k_assoc_loop.inc: the associator's tile loop (k_assoc.hip) as gfx950 assembly.

The loop is 32 v_mfma_i32_32x32x32_i8 per 64-row map tile and wave, and what decides its speed is the ORDER of the
instructions around them: every B fragment must be on its way from LDS while the MFMAs of the previous one run, the
arg-max of a half tile (one v_max per accumulator register) must issue under the MFMAs of the next half, the LDS-DMA
pieces of the tile three ahead must be spread between them, and nothing may wait for an MFMA result.  hipcc's scheduler
does not keep such an order (it fuses the two v_max into a v_max3 that needs both accumulator sets at once, copies the
loop-carried set, and serialises the chains), so the loop is emitted here instruction by instruction with a fixed
register map.  Run `python tools/gen_assoc_loop.py` after changing it; the .inc file is committed.

Register map (VGPRs, clobbered by the asm statement; the compiler keeps v0..v39):
  v[40:103]   A[b][s]     query fragments, b = 0,1 (32 queries each), s = 0..7 (32 bits of the code each); INPUTS of
              the statement (bound to these registers by "{v[..]}" constraints), like AX below
  v[104:135]  Bf[s]       map fragments of the current half tile; fragment s is refreshed in place right after its MFMAs
  v[136:167]  accP[b]     accumulators of half 0 (map rows 0..31 of the tile)
  v[168:199]  accQ[b]     accumulators of half 1
  v[200:231]  running[b][r]  running arg-max keys
  plain:  v[232:247] negt (the block counter, C operand of the first MFMA of a chain)
  gated:  v[232:239] AX[b] (ninth-step query operands), v[240:243] BX (ninth-step map operand), v245 scratch
  v248, v249  fragment read base in the LDS buffer of the current / next tile;  v250, v251 the same for the ninth-step rows
SGPRs s40..s57 are scratch (s[50:51] / s[52:53]: base address of the map chunk / of its ninth-step rows; s54 / s55:
byte offset of the current / next LDS buffer).
Hazards (no hardware interlock): an MFMA result is read by a VALU op at least two MFMAs later; a VALU-written MFMA
operand is followed by s_nop 1; M0 is followed by s_nop 0 before the LDS-DMA that uses it.
"""
import os

A0, BF, ACCP, ACCQ, RUN, NEGT = 40, 104, 136, 168, 200, 232
AX, BX, TMPV = 232, 240, 245
VB_CUR, VB_NEXT, VC_CUR, VC_NEXT = 248, 249, 250, 251


def rng(base, n):
    return "v[%d:%d]" % (base, base + n - 1)


def gen(gated):
    L = []
    e = L.append
    n_piece = 5 if gated else 4
    lgk_half0 = 8 if gated else 7

    def dma_piece(buf, p):
        # piece p of the tile whose address is in s[42:43] (ninth-step rows: s[48:49]) into LDS buffer `buf`
        # (a number in the prologue; None inside the loop = the buffer of the current tile, byte offset in s54)
        if p < 4:
            if buf is None:
                e("s_add_u32 s45, %[m0base], s54")
                if p:
                    e("s_add_u32 s45, s45, %d" % (p * 4096))
            else:
                e("s_add_u32 s45, %%[m0base], %d" % (buf * 16384 + p * 4096))
            e("s_mov_b32 m0, s45")
            e("s_nop 0")
            e("global_load_lds_dwordx4 %[voff], s[42:43]")
            e("s_add_u32 s42, s42, 4096")
            e("s_addc_u32 s43, s43, 0")
        else:
            if buf is None:
                e("s_lshr_b32 s45, s54, 3")
                e("s_add_u32 s45, %[m0c], s45")
            else:
                e("s_add_u32 s45, %%[m0c], %d" % (buf * 2048))
            e("s_mov_b32 m0, s45")
            e("s_nop 0")
            e("global_load_lds_dwordx4 %[voff], s[48:49]")

    def tile_address(from_k_plus):
        # s[42:43] (and s[48:49]) = address of tile min(k + from_k_plus, last)
        e("s_add_i32 s44, s40, %d" % from_k_plus)
        e("s_min_i32 s44, s44, s41")
        e("s_lshl_b32 s42, s44, 14")
        e("s_add_u32 s42, s50, s42")
        e("s_addc_u32 s43, s51, 0")
        if gated:
            e("s_lshl_b32 s48, s44, 11")
            e("s_add_u32 s48, s52, s48")
            e("s_addc_u32 s49, s53, 0")

    def tb(half):
        # s46 = the two counter bytes of column block tt = 2 k + half: [-(tt >> 4), -(tt & 15)]
        e("s_lshl_b32 s46, s40, 1")
        if half:
            e("s_or_b32 s46, s46, 1")
        e("s_lshr_b32 s47, s46, 4")
        e("s_sub_u32 s47, 0, s47")
        e("s_and_b32 s47, s47, 0xff")
        e("s_and_b32 s46, s46, 15")
        e("s_sub_u32 s46, 0, s46")
        e("s_and_b32 s46, s46, 0xff")
        e("s_lshl_b32 s46, s46, 8")
        e("s_or_b32 s46, s46, s47")

    def half_step(half):
        acc = ACCP if half == 0 else ACCQ
        other = ACCQ if half == 0 else ACCP
        vb = "v%d" % (VB_CUR if half == 0 else VB_NEXT)
        vc = "v%d" % (VC_CUR if half == 0 else VC_NEXT)
        roff = 512 if half == 0 else 0          # rows 32..63 of the current tile / rows 0..31 of the next
        if gated:
            tb(half)
            if half == 0:
                e("s_waitcnt lgkmcnt(%d)" % lgk_half0)
            e("v_and_b32 v%d, s46, %%[tmaskv]" % TMPV)
            e("v_or_b32 v%d, v%d, v%d" % (BX, BX, TMPV))
            e("s_nop 1")
            for b in range(2):
                e("v_mfma_i32_32x32x32_i8 %s, %s, %s, 0" % (rng(acc + 16 * b, 16), rng(AX + 4 * b, 4), rng(BX, 4)))
            e("ds_read_b128 %s, %s offset:%d" % (rng(BX, 4), vc, 1024 if half == 0 else 0))
        for s in range(8):
            if half == 0:
                e("s_waitcnt lgkmcnt(%d)" % lgk_half0)
            for b in range(2):
                c = rng(NEGT, 16) if (s == 0 and not gated) else rng(acc + 16 * b, 16)
                e("v_mfma_i32_32x32x32_i8 %s, %s, %s, %s" % (rng(acc + 16 * b, 16), rng(A0 + 32 * b + 4 * s, 4), rng(BF + 4 * s, 4), c))
            e("ds_read_b128 %s, %s offset:%d" % (rng(BF + 4 * s, 4), vb, s * 2048 + roff))
            for j in range(4):
                i = 4 * s + j
                e("v_max_i32 v%d, v%d, v%d" % (RUN + i, RUN + i, other + i))
            if not gated and s >= 4:
                for j in range(4):
                    r = 4 * (s - 4) + j
                    e("v_add_u32 v%d, -1, v%d" % (NEGT + r, NEGT + r))
            if half == 0 and s == 1:
                tile_address(3)
            if half == 1 and s in (0, 2, 4, 6):
                dma_piece(None, s // 2)
            if half == 1 and gated and s == 7:
                dma_piece(None, 4)

    # ---------------- prologue
    # (the query fragments A[b][s] -- and AX[b] -- arrive in their registers as operands of the statement)
    e("s_mov_b64 s[50:51], %[mbase]")
    if gated:
        e("s_mov_b64 s[52:53], %[cbase]")
    for i in range(32):
        e("v_bfrev_b32 v%d, 1" % (RUN + i))
        e("v_bfrev_b32 v%d, 1" % (ACCQ + i))
    if not gated:
        for r in range(16):
            e("v_mov_b32 v%d, 0" % (NEGT + r))
    e("v_mov_b32 v%d, %%[vfrag]" % VB_CUR)
    e("v_add_u32 v%d, 16384, %%[vfrag]" % VB_NEXT)
    if gated:
        e("v_mov_b32 v%d, %%[vcfrag]" % VC_CUR)
        e("v_add_u32 v%d, 2048, %%[vcfrag]" % VC_NEXT)
    e("s_mov_b32 s40, 0")
    e("s_add_i32 s41, %[ntiles], -1")
    e("s_mov_b32 s54, 0")
    e("s_mov_b32 s55, 16384")
    for t in range(3):
        tile_address(t)
        for p in range(n_piece):
            dma_piece(t, p)
    e("s_waitcnt vmcnt(%d)" % (2 * n_piece))
    e("s_barrier")
    for s in range(8):
        e("ds_read_b128 %s, %%[vfrag] offset:%d" % (rng(BF + 4 * s, 4), s * 2048))
    if gated:
        # issued AFTER the eight fragments here, BEFORE them in the loop: the first half 0 waits for everything
        e("ds_read_b128 %s, %%[vcfrag]" % rng(BX, 4))
        e("s_waitcnt lgkmcnt(0)")
    # ---------------- the loop: ONE copy of the tile body (instruction fetch is not free: a cold 64-byte line costs
    # hundreds of cycles and every wave of the chip runs this code at the same moment), the three LDS buffers rotate
    # through s54 / s55 (byte offset of the current / next buffer) and the read bases derived from them
    e("LT_%=:")
    half_step(0)
    e("s_waitcnt vmcnt(%d) lgkmcnt(0)" % n_piece)
    e("s_barrier")
    half_step(1)
    e("v_mov_b32 v%d, v%d" % (VB_CUR, VB_NEXT))
    e("s_mov_b32 s54, s55")
    e("s_add_u32 s55, s55, 16384")
    e("s_cmp_eq_u32 s55, 49152")
    e("s_cselect_b32 s55, 0, s55")
    e("v_add_u32 v%d, s55, %%[vfrag]" % VB_NEXT)
    if gated:
        e("v_mov_b32 v%d, v%d" % (VC_CUR, VC_NEXT))
        e("s_lshr_b32 s56, s55, 3")
        e("v_add_u32 v%d, s56, %%[vcfrag]" % VC_NEXT)
    e("s_add_i32 s40, s40, 1")
    e("s_cmp_lt_i32 s40, %[ntiles]")
    e("s_cbranch_scc1 LT_%=")
    # ---------------- epilogue: last half's keys, then the running keys go to LDS (the tile buffers are free)
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_nop 7")
    e("s_nop 7")
    for i in range(32):
        e("v_max_i32 v%d, v%d, v%d" % (RUN + i, RUN + i, ACCQ + i))
    e("s_barrier")
    for i in range(8):
        e("ds_write_b128 %%[vdump], %s offset:%d" % (rng(RUN + 4 * i, 4), i * 1024))
    e("s_waitcnt lgkmcnt(0)")
    return L


def gen_fp4(nbuf=6, gated=False, nrb=2):
    """The ungated loop on the FP4 matrix instruction (gfx950 only): code bits as e2m1 +-1 nibbles, 64 bits per
    v_mfma_scale_f32_32x32x64_f8f6f4 (same 32 cycles as the int8 32x32x32: twice the bits per cycle), the query side
    scaled by 2^9 through the E8M0 block scale, f32 accumulation (exact: |512 dot| <= 2^17).  Four matrix steps per
    32-column block instead of eight; the block counter enters through a FIFTH step computed once per half tile into a
    register set of its own (nt) that is the C operand of both row blocks' chains -- on the vector side that leaves only the
    32 v_max_f32 per half tile.  nt for half h + 1 is issued at the end of half h (two sets, no wait).
    Step five: query-side weights 4, 4, 0.5, 0.5 (block scale 2^4: 64, 64, 8, 8) in the lanes of k-half 0 and 1, 1
    (scale 2^0) in those of k-half 1; map side: the block number t = 64 a + 8 b + c as minus its octal digits, each digit
    the sum of two e2m1 values (5 = 4 + 1, 7 = 4 + 3): one dword per lane and half step, read from a 4 KB table in LDS that
    the C++ prologue fills (operand vtab = this lane's table base).
    Registers: v[80:95] Bf[s] (4 fragments), v[96:127] accP, v[128:159] accQ, v[160:191] running, v[200:215] ntP,
    v[216:231] ntQ, v[76:79] BX (v76 digits, v77..79 zero), v192 / v193 scale 2^9 / 2^0, v197 scratch, v198 / v199 read
    bases; A[b][s], AX, the step-five scale and the two lane masks are operands.
    `nbuf` 8 KB tile buffers: a half tile is only ~150 ns of matrix work per wave, so the tile nbuf - 1 ahead must be in
    flight to cover a cold (MALL / HBM) fetch when few workgroups of an XCD share a chunk.
    gated: colour gating as one more matrix step per row block, first in the chain (C = nt): query side 6.0 in the nibble
    of ITS colour (k-half 0, block scale 2^12), map side -6.0 in the nibbles of the OTHER colours (32-byte rows streamed
    beside the tiles like the int8 kernel's ninth-step rows) -> -147 456 whenever the colours differ, below every key within
    128 bits (>= -511).  v[232:235] BXC (refreshed in place like the fragments), v236 / v237 its read bases, v238 its scale.
    nrb = 1 (round 6, the SMALL shape for pipelined use: k_assoc_fp4_s): ONE 32-query row block per wave -- 16 accumulator,
    16 running and 16 + 16 counter registers less the second block's, the map in registers v48 .. v160 (three waves per SIMD),
    three tile buffers: 24 KB and ~160 registers per workgroup instead of 54 KB and 239, so that its workgroups find room on
    a chip full of region-growing waves; twice the LDS reads per matrix step (a fragment serves one row block), slower alone."""
    L = []
    e = L.append
    if nrb == 2:
        BF, ACCP, ACCQ, RUN, NTP, NTQ, BXR, SCA, SCB, TMP, VBC, VBN = 80, 96, 128, 160, 200, 216, 76, 192, 193, 197, 198, 199
        BXC, VCC, VCN, SCC = 232, 236, 237, 238
    else:
        BXR, BF, ACCP, ACCQ, RUN, NTP, NTQ, SCA, SCB, TMP, VBC, VBN = 48, 52, 68, 84, 100, 116, 132, 148, 149, 150, 151, 152
        BXC, VCC, VCN, SCC = 154, 158, 159, 160
    NA = 16 * nrb                        # accumulator / running registers per half tile
    npw = 3 if gated else 2              # LDS-DMA pieces per tile and wave
    lgk0 = 5 if gated else 4             # LDS reads in flight per half step, minus one
    MF = "v_mfma_scale_f32_32x32x64_f8f6f4"
    TAIL = "op_sel_hi:[0,0,0] cbsz:4 blgp:4"

    def dma_piece(buf, p):
        if p == 2:                               # the tile's colour rows (gated): 2 KB, piece = wave & 1 (folded into the operands)
            if buf is None:
                e("s_lshr_b32 s45, s54, 3")
                e("s_add_u32 s45, %[m0c], s45")
            else:
                e("s_add_u32 s45, %%[m0c], %d" % (buf * 2048))
            e("s_mov_b32 m0, s45")
            e("s_nop 0")
            e("global_load_lds_dwordx4 %[voff], s[48:49]")
            return
        if buf is None:
            e("s_lshr_b32 s45, s54, 1")
            e("s_add_u32 s45, %[m0base], s45")
            if p:
                e("s_add_u32 s45, s45, %d" % (p * 4096))
        else:
            e("s_add_u32 s45, %%[m0base], %d" % (buf * 8192 + p * 4096))
        e("s_mov_b32 m0, s45")
        e("s_nop 0")
        e("global_load_lds_dwordx4 %[voff], s[42:43]")
        e("s_add_u32 s42, s42, 4096")
        e("s_addc_u32 s43, s43, 0")

    def tile_address(from_k_plus):
        e("s_add_i32 s44, s40, %d" % from_k_plus)
        e("s_min_i32 s44, s44, s41")
        e("s_lshl_b32 s42, s44, 13")
        e("s_add_u32 s42, s50, s42")
        e("s_addc_u32 s43, s51, 0")
        if gated:
            e("s_lshl_b32 s48, s44, 11")
            e("s_add_u32 s48, s52, s48")
            e("s_addc_u32 s49, s53, 0")

    def t_read(tt_lines):
        # BX word of the half step whose block number the lines leave in s44: one dword per lane from the table the
        # C++ prologue wrote to LDS ([k-half][512] words: minus the octal digits of t as e2m1 pairs, see the doc string)
        for ln in tt_lines:
            e(ln)
        e("s_lshl_b32 s44, s44, 2")
        e("v_add_u32 v%d, s44, %%[vtab]" % TMP)
        e("ds_read_b32 v%d, v%d" % (BXR, TMP))

    def nt_mfma(dst):
        e("%s %s, %%[ax], %s, 0, %%[scl5], v%d %s" % (MF, rng(dst, 16), rng(BXR, 4), SCB, TAIL))

    def half_step(half):
        acc = ACCP if half == 0 else ACCQ
        other = ACCQ if half == 0 else ACCP
        nt = NTP if half == 0 else NTQ
        nt_next = NTQ if half == 0 else NTP
        vb = "v%d" % (VBC if half == 0 else VBN)
        roff = 512 if half == 0 else 0
        # the NEXT half step's block number (tt + 1) is fetched first: it is the oldest LDS read at the end of the step
        if half == 0:
            t_read(["s_lshl_b32 s44, s40, 1", "s_or_b32 s44, s44, 1"])
        else:
            t_read(["s_lshl_b32 s44, s40, 1", "s_add_u32 s44, s44, 2"])
        if gated:
            if half == 0:
                e("s_waitcnt lgkmcnt(%d)" % lgk0)
            for b in range(nrb):
                e("%s %s, %%[axc%d], %s, %s, v%d, v%d %s" % (MF, rng(acc + 16 * b, 16), b, rng(BXC, 4), rng(nt, 16), SCC, SCB, TAIL))
            e("ds_read_b128 %s, v%d offset:%d" % (rng(BXC, 4), VCC if half == 0 else VCN, 1024 if half == 0 else 0))
        for s in range(4):
            if half == 0:
                e("s_waitcnt lgkmcnt(%d)" % lgk0)
            for b in range(nrb):
                c = rng(nt, 16) if (s == 0 and not gated) else rng(acc + 16 * b, 16)
                e("%s %s, %%[a%d%d], %s, %s, v%d, v%d %s" % (MF, rng(acc + 16 * b, 16), b, s, rng(BF + 4 * s, 4), c, SCA, SCB, TAIL))
            e("ds_read_b128 %s, %s offset:%d" % (rng(BF + 4 * s, 4), vb, s * 2048 + roff))
            for j in range(NA // 4):
                i = (NA // 4) * s + j
                e("v_max_f32 v%d, v%d, v%d" % (RUN + i, RUN + i, other + i))
            if half == 0 and s == 1:
                tile_address(nbuf)
            if half == 1 and s in (0, 2):
                dma_piece(None, s // 2)
            if half == 1 and gated and s == 3:
                dma_piece(None, 2)
        # the block counter of the NEXT half step, into the other nt set
        e("s_waitcnt lgkmcnt(%d)" % (0 if half == 0 else lgk0))
        nt_mfma(nt_next)

    # ---------------- prologue
    e("s_mov_b64 s[50:51], %[mbase]")
    if gated:
        e("s_mov_b64 s[52:53], %[cbase]")
        e("v_mov_b32 v%d, 0x8b8b8b8b" % SCC)
        e("v_mov_b32 v%d, %%[vcfrag]" % VCC)
        e("v_add_u32 v%d, 2048, %%[vcfrag]" % VCN)
    for i in range(NA):
        e("v_mov_b32 v%d, 0xff800000" % (RUN + i))
        e("v_mov_b32 v%d, 0xff800000" % (ACCQ + i))
    for r in range(1, 4):
        e("v_mov_b32 v%d, 0" % (BXR + r))
    e("v_mov_b32 v%d, 0x88888888" % SCA)
    e("v_mov_b32 v%d, 0x7f7f7f7f" % SCB)
    e("v_mov_b32 v%d, %%[vfrag]" % VBC)
    e("v_add_u32 v%d, 8192, %%[vfrag]" % VBN)
    e("s_mov_b32 s40, 0")
    e("s_add_i32 s41, %[ntiles], -1")
    e("s_mov_b32 s54, 0")
    e("s_mov_b32 s55, 16384")               # s54 / s55 count in the int8 loop's units (16 KB per buffer); halved where used
    for t in range(nbuf):
        tile_address(t)
        for p in range(npw):
            dma_piece(t, p)
    t_read(["s_mov_b32 s44, 0"])
    e("s_waitcnt lgkmcnt(0)")
    nt_mfma(NTP)
    e("s_waitcnt vmcnt(%d)" % (npw * (nbuf - 1)))
    e("s_barrier")
    if gated:
        e("ds_read_b128 %s, %%[vcfrag]" % rng(BXC, 4))
    for s in range(4):
        e("ds_read_b128 %s, %%[vfrag] offset:%d" % (rng(BF + 4 * s, 4), s * 2048))
    e("LT_%=:")
    half_step(0)
    e("s_waitcnt vmcnt(%d)" % (npw * (nbuf - 2)))
    e("s_barrier")
    half_step(1)
    e("v_mov_b32 v%d, v%d" % (VBC, VBN))
    e("s_mov_b32 s54, s55")
    e("s_add_u32 s55, s55, 16384")
    e("s_cmp_eq_u32 s55, %d" % (nbuf * 16384))
    e("s_cselect_b32 s55, 0, s55")
    e("s_lshr_b32 s56, s55, 1")
    e("v_add_u32 v%d, s56, %%[vfrag]" % VBN)
    if gated:
        e("v_mov_b32 v%d, v%d" % (VCC, VCN))
        e("s_lshr_b32 s56, s55, 3")
        e("v_add_u32 v%d, s56, %%[vcfrag]" % VCN)
    e("s_add_i32 s40, s40, 1")
    e("s_cmp_lt_i32 s40, %[ntiles]")
    e("s_cbranch_scc1 LT_%=")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_nop 7")
    e("s_nop 7")
    for i in range(NA):
        e("v_max_f32 v%d, v%d, v%d" % (RUN + i, RUN + i, ACCQ + i))
    e("s_barrier")
    for i in range(NA // 4):
        e("ds_write_b128 %%[vdump], %s offset:%d" % (rng(RUN + 4 * i, 4), i * 1024))
    e("s_waitcnt lgkmcnt(0)")
    return L


def c_string(lines):
    return "\n".join('    "%s\\n\\t"' % ln for ln in lines)


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "lane_slam_amd", "csrc", "k_assoc_loop.inc")
    sclob = ", ".join('"s%d"' % i for i in range(40, 58)) + ', "m0", "memory", "scc"'
    clob_plain = ", ".join('"v%d"' % i for i in range(104, 252)) + ", " + sclob
    clob_gated = ", ".join('"v%d"' % i for i in list(range(104, 232)) + list(range(240, 252))) + ", " + sclob
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_assoc_loop.py -- do not edit; see that file for the register map and the schedule.\n")
        f.write("#define LF_ASSOC_LOOP_PLAIN \\\n" + c_string(gen(False)).replace("\n", " \\\n") + "\n\n")
        f.write("#define LF_ASSOC_LOOP_GATED \\\n" + c_string(gen(True)).replace("\n", " \\\n") + "\n\n")
        f.write("#define LF_ASSOC_LOOP_FP4 \\\n" + c_string(gen_fp4()).replace("\n", " \\\n") + "\n\n")
        f.write("#define LF_ASSOC_LOOP_FP4_GATED \\\n" + c_string(gen_fp4(gated=True)).replace("\n", " \\\n") + "\n\n")
        clob_fp4 = ", ".join('"v%d"' % i for i in list(range(76, 200)) + list(range(200, 239))) + ", " + ", ".join('"s%d"' % i for i in range(40, 60)) + ', "m0", "memory", "scc"'
        f.write("#define LF_ASSOC_LOOP_CLOBBERS_FP4 " + clob_fp4 + "\n")
        # the small shape (round 6): one row block per wave, three tile buffers, registers v48 .. v160
        f.write("#define LF_ASSOC_LOOP_FP4_S \\\n" + c_string(gen_fp4(nbuf=3, nrb=1)).replace("\n", " \\\n") + "\n\n")
        f.write("#define LF_ASSOC_LOOP_FP4_GATED_S \\\n" + c_string(gen_fp4(nbuf=3, gated=True, nrb=1)).replace("\n", " \\\n") + "\n\n")
        clob_fp4_s = ", ".join('"v%d"' % i for i in range(48, 161)) + ", " + ", ".join('"s%d"' % i for i in range(40, 60)) + ', "m0", "memory", "scc"'
        f.write("#define LF_ASSOC_LOOP_CLOBBERS_FP4_S " + clob_fp4_s + "\n")
        f.write("#define LF_ASSOC_LOOP_CLOBBERS_PLAIN " + clob_plain + "\n")
        f.write("#define LF_ASSOC_LOOP_CLOBBERS_GATED " + clob_gated + "\n")
    print("wrote", os.path.normpath(out), "plain", len(gen(False)), "gated", len(gen(True)), "instructions")


if __name__ == "__main__":
    main()
