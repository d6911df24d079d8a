#!/usr/bin/env python3
"""Writes tools/micro/mfma_mix_asm.inc: hand-scheduled loop bodies for mfma_mix.hip (modes 1-3), one asm statement each.
Registers: a[0:31] accumulators of the running block (two tiles), a[32:63] the finished block; v[0:15] A-operand slots (4),
v[16:31] / v[32:47] B operands of tile 0 / 1 (4 K-steps each, cycled), v48 / v49 LDS address of the current / next ring chunk,
v[50:53] conversion temporaries, v[54:69] staging of the ring refill, v70 the wave's LDS refill address, v[72:73] global address."""
import sys


def body(mode):
    L = []
    e = L.append
    e("s_mov_b32 s40, %[iters]")
    e("s_mov_b32 s41, 0")                       # chunk index v48 points at
    e("v_readfirstlane_b32 s43, %[lds]")
    e("v_mov_b32 v48, %[lds]")
    e("v_add_u32 v49, 0x4000, v48")
    e("v_add_u32 v70, 0xc000, %[lds]")
    e("v_add_u32 v70, %[roff], v70")
    for i in range(8):
        e("global_load_dwordx4 v[%d:%d], %%[loff], %%[bsrc%s] offset:%d" % (16 + 4 * i, 19 + 4 * i, "2" if i >= 4 else "", 1024 * (i % 4)))
    for i in range(64):
        e("v_accvgpr_write_b32 a%d, 0" % i)
    e("s_waitcnt vmcnt(0)")
    if mode == 16:
        e("s_mov_b64 s[46:47], %[src]")
        e("s_mov_b64 s[48:49], %[src]")
        e("s_mov_b32 s45, 0")
    if mode == 8 or mode >= 13:
        e("s_add_u32 s44, s43, 0xc000")
        e("s_mov_b32 m0, s44")
    if mode == 3:
        for k in range(4):
            e("global_load_dwordx4 v[%d:%d], %%[roff], %%[src] offset:%d" % (54 + 4 * k, 57 + 4 * k, 1024 * k))
    for sl in range(3):
        e("ds_read_b128 v[%d:%d], v48 offset:%d" % (4 * sl, 4 * sl + 3, 1024 * sl))
    for blk in range(2):                        # two blocks per loop trip: accumulator sets swap
        cur = 0 if blk == 0 else 32
        old = 32 - cur
        if blk == 0:
            e("1:")
            # first three A operands of the trip
        if blk == 0:
            pass
        fill = []                               # VALU / memory work to spread over the 32 MFMAs of the block
        if mode in (2, 3):
            for t in range(2):
                for r in range(0, 16, 2):
                    fill.append("v_accvgpr_read_b32 v50, a%d" % (old + 16 * t + r))
                    fill.append("v_accvgpr_read_b32 v51, a%d" % (old + 16 * t + r + 1))
                    fill.append("v_cvt_pkrtz_f16_f32 v52, v50, v51")
                    fill.append("v_pk_max_f16 v53, v52, v52")
        mem = {}
        if 4 <= mode <= 11:
            op = {4: "ds_write_b128 v70, v[54:57]", 5: "ds_write_b128 v70, a[64:67]", 6: "global_load_dwordx4 v[54:57], %[roff], %[src]",
                  7: "global_load_dwordx4 a[64:67], %[roff], %[src]", 8: "global_load_lds_dwordx4 %[roff], %[src]",
                  9: "ds_write_b128 v70, v[54:57]", 10: "ds_write_b64 v70, v[54:55]", 11: "ds_write_b32 v70, v54"}[mode]
            for k in range(4):
                mem[8 * k + (2 if mode == 9 else 1)] = (op + " offset:%d" % (1024 * k),)
        elif mode >= 12:
            # the shipped ring protocol in parts: 12 = rendezvous only, 13 = four LDS-DMA pieces per chunk only, 14 = DMA + the counted wait,
            # 15 = DMA + wait + rendezvous (the kernel's pattern)
            if mode in (12, 15, 16):
                mem[23] = ("s_barrier",)
            if mode in (14, 15, 16):
                mem[22] = ("s_waitcnt vmcnt(0)",)
            if mode >= 13:
                for k in range(4):
                    mem[24 + 2 * k] = ("global_load_lds_dwordx4 %%[roff], %s offset:%d" % ("s[46:47]" if mode == 16 else "%[src]", 1024 * k),)
            if mode == 16:      # ... and the source walks a 1.44 MB stream (90 chunks of 16 KB), as the kernel's does: real L2 traffic
                mem[31] = ("s_add_u32 s46, s46, 0x4000", "s_addc_u32 s47, s47, 0", "s_add_u32 s45, s45, 1", "s_cmp_eq_u32 s45, 90",
                           "s_cselect_b32 s45, 0, s45", "s_cselect_b32 s46, s48, s46", "s_cselect_b32 s47, s49, s47")
        elif mode >= 3:
            # the rendezvous of the planned kernel: barrier in front of step 12, then one (publish, refill) pair per step in steps 12..15
            mem[23] = ("s_barrier",)
            for k in range(4):
                mem[24 + 2 * k] = ("s_waitcnt vmcnt(3)", "ds_write_b128 v70, v[%d:%d] offset:%d" % (54 + 4 * k, 57 + 4 * k, 1024 * k),
                                   "global_load_dwordx4 v[%d:%d], %%[roff], %%[src] offset:%d" % (54 + 4 * k, 57 + 4 * k, 1024 * k))
        fi = 0
        for s in range(16):
            slot = (s + 3) % 4
            if s < 13:
                e("ds_read_b128 v[%d:%d], v48 offset:%d" % (4 * slot, 4 * slot + 3, (s + 3) * 1024))
            else:
                e("ds_read_b128 v[%d:%d], v49 offset:%d" % (4 * slot, 4 * slot + 3, (s + 3 - 16) * 1024))
            e("s_waitcnt lgkmcnt(3)")
            for t in range(2):
                m = 2 * s + t
                e("v_mfma_f32_32x32x16_f16 a[%d:%d], v[%d:%d], v[%d:%d], a[%d:%d]" % (cur + 16 * t, cur + 16 * t + 15, 4 * (s % 4), 4 * (s % 4) + 3,
                                                                                      16 + 16 * t + 4 * (s % 4), 19 + 16 * t + 4 * (s % 4), cur + 16 * t, cur + 16 * t + 15))
                if m in mem:
                    x = mem[m]
                    for q in (x if isinstance(x, tuple) else (x,)):
                        e(q)
                for _ in range(2):
                    if fi < len(fill):
                        e(fill[fi]); fi += 1
        assert fi == len(fill)
        # next chunk
        e("s_add_u32 s41, s41, 1")
        e("s_cmp_eq_u32 s41, 3")
        e("s_cselect_b32 s41, 0, s41")
        e("s_add_u32 s42, s41, 1")
        e("s_cmp_eq_u32 s42, 3")
        e("s_cselect_b32 s42, 0, s42")
        e("s_lshl_b32 s44, s41, 14")
        e("s_add_u32 s44, s44, s43")
        e("v_mov_b32 v48, s44")
        e("s_lshl_b32 s44, s42, 14")
        e("s_add_u32 s44, s44, s43")
        e("v_mov_b32 v49, s44")
    e("s_sub_u32 s40, s40, 2")
    e("s_cmp_gt_i32 s40, 0")
    e("s_cbranch_scc1 1b")
    e("s_waitcnt lgkmcnt(0) vmcnt(0)")
    return L


def main():
    out = open(sys.argv[1], "w")
    for mode in range(1, 17):
        out.write("#define MIX_BODY_%d \\\n" % mode)
        out.write(" \\\n".join('    "%s\\n"' % l for l in body(mode)))
        out.write("\n\n")


if __name__ == "__main__":
    main()
