#!/usr/bin/env python3
"""Generator of the hand-scheduled instruction stream of k_mlp_fwd_ha (nf_mlp_ha.hip): the fp16-MFMA NeRF MLP forward
(/root/reference/models/nerf.py:83-124) of nf_mlp_h2.hip — its weight blocks (nf_nerf_pack_h2's 1 KB A blocks in consumption order, re-packed
by nf_nerf_pack_ha WITHOUT the 78 bias K-steps: their exact value fp32(hi) + fp32(lo) is the C operand of each output block's first MFMAs, read
from a table behind the stream), the same X layout, the same arithmetic in the same order (v_mfma_f32_32x32x16_f16, fp32 accumulate,
out-block-major, two 32-row tiles per wave, every finished block rounded to packed fp16 once) — bit-identical results, with every instruction of
the pair body placed by this script instead of by the compiler:

  * the whole pair (1 336 steps of one A block x two tiles + 8 padding steps without arithmetic = 84 chunks of 16 = 28 rings) is straight-line
    code: ring positions, register names, wait counts and the rendezvous steps are constants;
  * per step: the ds_read_b128 of the A block three steps ahead, (X steps) the two stash reads two steps ahead, ONE counted s_waitcnt, two
    MFMAs; everything else is a `piece` placed by a budgeted list scheduler into the shadow of a named MFMA: the conversion pieces of the
    previous block, the ring refill (s_barrier in front of step 12 of every chunk, then four LDS-DMA pieces; waited for by vmcnt in front of
    the next rendezvous), the next block's bias-table loads, the staging of the next pair's X into the LDS stash (layers 5-6), the
    direction-feature loads (layer 8), the row_sample loads, the per-pair scalar setup and the PREVIOUS pair's sigmoid + store;
  * registers: four accumulators v[0:63] (VGPR form: the VALU converts them directly), activation bank 0 v[64:191], the bias C operand
    v[192:207]; activation bank 1 a[0:127] (MFMA A / B operands are read from AGPRs directly), X staging a[144:159], direction operands
    a[160:191], stash operand slots a[192:223], A-operand slots a[224:239].

Output: a C string literal (one "...\\n" line per instruction) included by nf_mlp_ha.hip as the body of ONE asm statement.
Usage: python gen_mlp_ha.py [out.inc]      (neurofluid_amd/build.py runs it before compiling nf_mlp_ha.hip)
Dev switches: NF_HA_* in the environment (tools/ab_mlp_ha.py builds A/B libraries from them); NF_HA_TIMING=1 leaves block 0 / wave 0's
shader cycles and 100 MHz ticks in the stream's first padding block.
"""
import os
import sys


def knob(name, default=0):
    """A/B switches of the schedule; the shipped kernel is the all-defaults build."""
    return int(os.environ.get("NF_HA_" + name, default))


CHUNK, RING, XS = 16, 48, 13
RING_B = RING * 1024
STASH_W = 2 * XS * 1024          # per wave: [tile A: 13 K-steps][tile B: 13 K-steps] of 1 KB
LDS_BYTES = RING_B + 4 * STASH_W
PF = 3                           # A-operand prefetch distance (steps)
DMA = knob("DMA", 1)               # ring refill by LDS-DMA (0: global_load -> registers -> ds_write)
XPF = 2                          # stash-operand prefetch distance (steps)

IN = dict(stream="%0", X="%1", n_rows="%2", row_sample="%3", out="%4", max_rows="%5", wave="%6", block="%7", nblocks="%8", btab="%9")

_s = 40


def _salloc(n=1, align=1):
    global _s
    _s = (_s + align - 1) // align * align
    r = _s
    _s += n
    return r


S_WBASE = _salloc(2, 2); S_WCUR = _salloc(2, 2); S_XH = _salloc(2, 2); S_XD = _salloc(2, 2); S_XDB = _salloc(2, 2)
S_XN = _salloc(2, 2); S_XNB = _salloc(2, 2); S_RS = _salloc(2, 2); S_OUT = _salloc(2, 2); S_TMP = _salloc(2, 2)
S_SAVE = _salloc(2, 2); S_EXA = _salloc(2, 2); S_EXB = _salloc(2, 2); S_DV = _salloc(2, 2)
S_NROWS = _salloc(); S_NPAIRS = _salloc(); S_NGROUPS = _salloc(); S_TG = _salloc(); S_WAVE = _salloc(); S_NBLK = _salloc()
S_PAIR = _salloc(); S_LASTP = _salloc(); S_NL2E = _salloc(); S_EHI = _salloc(); S_ELO = _salloc(); S_T1 = _salloc(); S_OWNER = _salloc(); S_WAVE4K = _salloc(); S_PEXA = _salloc(2, 2); S_PEXB = _salloc(2, 2); S_BTBASE = _salloc(2, 2); S_BTCUR = _salloc(2, 2)
assert _s <= 100


def sp(r):
    return f"s[{r}:{r + 1}]"


# ---- vector registers -----------------------------------------------------------------------------------------------
def ACC(buf, tile):
    b = (2 * buf + tile) * 16
    return f"v[{b}:{b + 15}]"


def acc_reg(buf, tile, r):
    return (2 * buf + tile) * 16 + r


def bank_reg(bank, tile, k, c=0):
    """register NAME of component c of K-step k of a tile's packed activations"""
    if bank == 0:
        return f"v{64 + tile * 64 + 4 * k + c}"
    return f"a{tile * 64 + 4 * k + c}"


def bank_quad(bank, tile, k):
    if bank == 0:
        b = 64 + tile * 64 + 4 * k
        return f"v[{b}:{b + 3}]"
    b = tile * 64 + 4 * k
    return f"a[{b}:{b + 3}]"


def ASL(i):
    return f"a[{224 + 4 * (i & 3)}:{227 + 4 * (i & 3)}]"


BIASC = 192                                  # v[192:207]: the C operand of a block's first MFMAs (the block's biases, from the table)
V_H64 = 208                                  # (lane >> 5) * 64: the lane half's row of a bias-table entry
V_T = [212, 213, 214, 215]
V_LANE16, V_STASH, V_PUB = 216, 217, 218
V_SIG = [220, 221]
V_PRGB = [222, 223, 224, 225, 226, 227]     # the PREVIOUS pair's raw rgb head outputs: tile A r, g, b; tile B r, g, b
V_ROW, V_IDXOFF = 228, 229
V_IDX = [230, 231]
V_PADDR = [232, 234]                         # the previous pair's two store addresses (2 registers each)
V_O = 236                                    # 4 registers
V_E = list(range(240, 256))
V_PSIG = [240, 241]                          # the previous pair's sigma (E[0], E[1]: the sigmoid uses E[3..14])


def STAGE(q):
    return f"a[{128 + 4 * q}:{131 + 4 * q}]"


def XF(s_, tile):
    b = 144 + 8 * (s_ & 1) + 4 * tile
    return f"a[{b}:{b + 3}]"


def DIR(tile, t):
    b = 160 + 16 * tile + 4 * t
    return f"a[{b}:{b + 3}]"


def XR(tile, i):
    b = 192 + 16 * tile + 4 * (i & 3)
    return f"a[{b}:{b + 3}]"


class Prog:
    def __init__(self):
        self.lines = []
        self.vm = []            # outstanding vector-memory operations, in issue order (tags)
        self.lq = []            # outstanding LDS operations, in issue order (tags)

    def i(self, s):
        self.lines.append(s)

    def vmem(self, text, tag):
        self.i(text)
        self.vm.append(tag)
        assert len(self.vm) < 60, "vmcnt is a 6-bit counter"

    def lds(self, text, tag):
        self.i(text)
        self.lq.append(tag)
        assert len(self.lq) <= 15, "lgkmcnt is a 4-bit counter"

    def wait_vm(self, tag):
        if tag not in self.vm:
            return
        idx = len(self.vm) - 1 - self.vm[::-1].index(tag)
        n_after = len(self.vm) - 1 - idx
        self.vm = self.vm[idx + 1:]
        if not knob("NOWAITVM"):
            self.i(f"s_waitcnt vmcnt({n_after})")

    def wait_lds(self, tags):
        idx = -1
        for t in tags:
            if t in self.lq:
                idx = max(idx, len(self.lq) - 1 - self.lq[::-1].index(t))
        if idx < 0:
            return
        n_after = len(self.lq) - 1 - idx
        self.lq = self.lq[idx + 1:]
        if not knob("NOWAITLDS"):
            self.i(f"s_waitcnt lgkmcnt({n_after})")


_lb = [0]


def long_branch_scc1(p, target):
    n = _lb[0]
    _lb[0] += 1
    p.i(f"s_cbranch_scc0 .Lnf_ha_skip{n}_%=")
    p.i(f"s_getpc_b64 {sp(S_TMP)}")
    p.i(f".Lnf_ha_pc{n}_%=:")
    p.i(f"s_add_u32 s{S_TMP}, s{S_TMP}, ({target}-.Lnf_ha_pc{n}_%=)&4294967295")
    p.i(f"s_addc_u32 s{S_TMP + 1}, s{S_TMP + 1}, ({target}-.Lnf_ha_pc{n}_%=)>>32")
    p.i(f"s_setpc_b64 {sp(S_TMP)}")
    p.i(f".Lnf_ha_skip{n}_%=:")


# ------------------------------------------------------------------------------------------------------------------------
# the block list of one pair (nf_mlp_h2.hip: k_mlp_fwd_h2's body, block by block)
# ------------------------------------------------------------------------------------------------------------------------
class Block:
    __slots__ = ("name", "layer", "blk", "acc", "inb", "steps", "cvt", "first")

    def __init__(self, name, layer, blk, acc, inb, steps, cvt):
        self.name, self.layer, self.blk, self.acc, self.inb, self.steps, self.cvt = name, layer, blk, acc, inb, steps, cvt
        self.first = 0


def build_blocks():
    B = []
    bias = []                                               # (nf_mlp_h2.hip's bias K-step: here the C operand of the block's first MFMAs)
    xs = [("xs", t) for t in range(XS)]
    h16 = [("h", k) for k in range(16)]
    for b in range(8):                                       # layer 0: X (stash) -> bank 1
        B.append(Block("L0", 0, b, b & 1, None, bias + xs, None if b == 0 else ((b - 1) & 1, 1, b - 1, True)))

    def hidden(name, layer, inb, outb, relu_out, x_part):
        B.append(Block(name, layer, 0, 0, inb, bias + x_part + h16, (1, inb, 7, True)))
        for b in range(1, 8):
            B.append(Block(name, layer, b, b & 1, inb, bias + x_part + h16, ((b - 1) & 1, outb, b - 1, relu_out)))

    hidden("L1", 1, 1, 0, True, [])
    hidden("L2", 2, 0, 1, True, [])
    hidden("L3", 3, 1, 0, True, [])
    hidden("L4", 4, 0, 1, True, xs)                          # skip layer: X from the stash, then the hidden part
    hidden("L5", 5, 1, 0, True, [])
    hidden("L6", 6, 0, 1, True, [])
    hidden("L7", 7, 1, 0, True, [])                          # bank 0 = relu(h8): input of xyz_encoding_final AND of sigma
    hidden("L8", 8, 0, 1, False, [])                         # xyz_encoding_final (no activation) -> bank 1
    B.append(Block("SIG", 8, 8, 0, 0, bias + h16, (1, 1, 7, False)))
    xd = [("xd", t) for t in range(4)]
    B.append(Block("V", 9, 0, 1, 1, bias + xd + h16, None))
    for b in range(1, 4):
        B.append(Block("V", 9, b, (b + 1) & 1, 1, bias + xd + h16, (b & 1, 0, b - 1, True)))
    B.append(Block("RGB", 10, 0, 1, 0, bias + [("h", k) for k in range(8)], (0, 0, 3, True)))
    n = 0
    for blk in B:
        blk.first = n
        n += len(blk.steps)
    return B, n


def cvt_piece(p_, cv):
    """conversion piece p_ (0..7) of a finished block (nf_mlp_h2.hip h2_cvt_piece): tile p_ >> 2, quarter q: accumulator floats
    4q..4q+3 -> packed registers of K-step 2 pblk + (q >> 1), components 2 (q & 1), + 1."""
    prev, outb, pblk, relu = cv
    tile, q = p_ >> 2, p_ & 3
    k = 2 * pblk + (q >> 1)
    a0 = acc_reg(prev, tile, 4 * q)
    ins = []
    for j in range(2):
        dst = bank_reg(outb, tile, k, 2 * (q & 1) + j)
        src = f"v{a0 + 2 * j}, v{a0 + 2 * j + 1}"
        t = f"v{V_T[2 * (p_ & 1) + j]}"
        if outb == 0:
            if relu:
                ins += [f"v_cvt_pk_f16_f32 {t}, {src}", f"v_pk_max_f16 {dst}, {t}, 0"]
            else:
                ins += [f"v_cvt_pk_f16_f32 {dst}, {src}"]
        else:
            ins += [f"v_cvt_pk_f16_f32 {t}, {src}"]
            if relu:
                ins += [f"v_pk_max_f16 {t}, {t}, 0"]
            ins += [f"v_accvgpr_write_b32 {dst}, {t}"]
    return ins


def sigmoid_pieces(x, out):
    """1 / (1 + expf(-x)) as the compiler expands it in k_mlp_fwd_h2 (same operations, same order: bit-identical), cut into pieces that
    keep every VCC producer with its consumer (v_div_fmas reads VCC implicitly)."""
    E = V_E
    t, ph, r, ri, e = E[3], E[4], E[5], E[6], E[7]
    d, ds, rc, e0, ns, q_, e1 = e, E[9], E[10], E[11], E[12], E[13], E[14]
    return [
        [f"v_mul_f32 v{t}, 0xbfb8aa3b, v{x}", f"v_fma_f32 v{ph}, v{x}, s{S_NL2E}, -v{t}", f"v_rndne_f32 v{r}, v{t}",
         f"v_fmac_f32 v{ph}, 0xb2a5705f, v{x}", f"v_sub_f32 v{t}, v{t}, v{r}", f"v_add_f32 v{t}, v{t}, v{ph}"],
        [f"v_cvt_i32_f32 v{ri}, v{r}", f"v_exp_f32 v{e}, v{t}", "s_nop 0", f"v_ldexp_f32 v{e}, v{e}, v{ri}"],
        [f"v_cmp_nlt_f32 vcc, s{S_EHI}, v{x}", "s_nop 1", f"v_cndmask_b32 v{e}, 0, v{e}, vcc", f"v_mov_b32 v{E[8]}, 0x7f800000"],
        [f"v_cmp_ngt_f32 vcc, s{S_ELO}, v{x}", "s_nop 1", f"v_cndmask_b32 v{e}, v{E[8]}, v{e}, vcc", f"v_add_f32 v{e}, 1.0, v{e}"],
        [f"v_div_scale_f32 v{ds}, {sp(S_DV)}, v{d}, v{d}, 1.0", f"v_rcp_f32 v{rc}, v{ds}", "s_nop 0", f"v_fma_f32 v{e0}, -v{ds}, v{rc}, 1.0"],
        [f"v_div_scale_f32 v{ns}, vcc, 1.0, v{d}, 1.0", f"v_fmac_f32 v{rc}, v{e0}, v{rc}", f"v_mul_f32 v{q_}, v{ns}, v{rc}",
         f"v_fma_f32 v{e1}, -v{ds}, v{q_}, v{ns}", f"v_fmac_f32 v{q_}, v{e1}, v{rc}", f"v_fma_f32 v{e1}, -v{ds}, v{q_}, v{ns}", "s_nop 1",
         f"v_div_fmas_f32 v{e1}, v{e1}, v{rc}, v{q_}"],
        [f"v_div_fixup_f32 v{out}, v{e1}, v{d}, 1.0"],
    ]


def epilogue_pieces():
    """the PREVIOUS pair's outputs (lanes h == 0 hold one row of tile A and one of tile B): rgb = sigmoid(raw head outputs carried over in
    V_PRGB), sigma carried in V_PSIG, the store addresses in V_PADDR, the store masks in S_PEXA / S_PEXB."""
    out = []
    for tile in range(2):
        for c in range(3):
            out += sigmoid_pieces(V_PRGB[3 * tile + c], V_O + c)
        out.append([f"v_mov_b32 v{V_O + 3}, v{V_PSIG[tile]}", f"s_mov_b64 {sp(S_TMP)}, exec", f"s_mov_b64 exec, {sp(S_PEXA if tile == 0 else S_PEXB)}",
                    ("vmem", f"global_store_dwordx4 v[{V_PADDR[tile]}:{V_PADDR[tile] + 1}], v[{V_O}:{V_O + 3}], off", "st"),
                    "s_nop 1", f"s_mov_b64 exec, {sp(S_TMP)}"])          # (S_TMP: only ever live inside one piece)
    return out


def emit_pair_select(p, s_tg, s_pair):
    """s_pair = min(tg * 4 + wave, last_pair)"""
    p.i(f"s_lshl_b32 s{s_pair}, s{s_tg}, 2")
    p.i(f"s_add_u32 s{s_pair}, s{s_pair}, s{S_WAVE}")
    p.i(f"s_min_i32 s{s_pair}, s{s_pair}, s{S_LASTP}")


def emit_x_base(p, s_pair, s_out, extra=0):
    """s_out(2) = Xh + pair * 32 KB + extra"""
    p.i(f"s_lshl_b32 s{S_TMP}, s{s_pair}, 15")
    p.i(f"s_lshr_b32 s{S_TMP + 1}, s{s_pair}, 17")
    p.i(f"s_add_u32 s{s_out}, s{S_XH}, s{S_TMP}")
    p.i(f"s_addc_u32 s{s_out + 1}, s{S_XH + 1}, s{S_TMP + 1}")
    if extra:
        p.i(f"s_add_u32 s{s_out}, s{s_out}, {extra}")
        p.i(f"s_addc_u32 s{s_out + 1}, s{s_out + 1}, 0")


def gen():
    p = Prog()
    blocks, nreal = build_blocks()
    NSLOT = (nreal + RING - 1) // RING * RING
    nchunks = NSLOT // CHUNK
    assert nchunks % 3 == 0
    step_of = []                         # global step -> (block, index in block)
    for b in blocks:
        for s in range(len(b.steps)):
            step_of.append((b, s))

    # ------------------------------------------------------------------ prologue
    p.i(f"s_mov_b64 {sp(S_WBASE)}, {IN['stream']}")
    if knob("TIMING"):          # dev: block 0 / wave 0 leaves (shader cycles, 100 MHz ticks) of its residency in the stream's first padding block
        p.i("s_memtime s[92:93]")
        p.i("s_memrealtime s[94:95]")
        p.i("s_waitcnt lgkmcnt(0)")
        p.i("s_mov_b32 s90, s92")
        p.i("s_mov_b32 s91, s94")
    p.i(f"s_mov_b64 {sp(S_XH)}, {IN['X']}")
    p.i(f"s_mov_b64 {sp(S_RS)}, {IN['row_sample']}")
    p.i(f"s_mov_b64 {sp(S_OUT)}, {IN['out']}")
    p.i(f"s_mov_b32 s{S_WAVE}, {IN['wave']}")
    p.i(f"s_mov_b32 s{S_TG}, {IN['block']}")
    p.i(f"s_mov_b32 s{S_NBLK}, {IN['nblocks']}")
    p.i(f"s_mov_b32 s{S_NROWS}, {IN['max_rows']}")
    p.i(f"s_load_dword s{S_TMP}, {IN['n_rows']}, 0x0")
    p.i(f"s_mov_b32 s{S_NL2E}, 0xbfb8aa3b")
    p.i(f"s_mov_b32 s{S_EHI}, 0x42ce8ed0")
    p.i(f"s_mov_b32 s{S_ELO}, 0xc2b17218")
    p.i(f"v_mbcnt_lo_u32_b32 v{V_E[0]}, -1, 0")
    p.i(f"v_mbcnt_hi_u32_b32 v{V_E[0]}, -1, v{V_E[0]}")                 # lane
    p.i(f"v_lshlrev_b32 v{V_LANE16}, 4, v{V_E[0]}")
    p.i(f"v_and_b32 v{V_ROW}, 31, v{V_E[0]}")                            # j
    p.i(f"v_lshrrev_b32 v{V_H64}, 5, v{V_E[0]}")
    p.i(f"v_lshlrev_b32 v{V_H64}, 6, v{V_H64}")                          # h * 64
    p.i(f"s_mov_b64 {sp(S_BTBASE)}, {IN['btab']}")
    p.i(f"s_lshl_b32 s{S_T1}, s{S_WAVE}, 12")
    p.i(f"v_add_u32 v{V_PUB}, s{S_T1}, v{V_LANE16}")                     # wave * 4096 + lane * 16 (ring write address AND stream offset)
    p.i(f"s_mul_i32 s{S_T1}, s{S_WAVE}, {STASH_W}")
    p.i(f"s_add_u32 s{S_T1}, s{S_T1}, {RING_B}")
    p.i(f"v_add_u32 v{V_STASH}, s{S_T1}, v{V_LANE16}")
    p.i("s_waitcnt lgkmcnt(0)")
    p.i(f"s_min_i32 s{S_NROWS}, s{S_TMP}, s{S_NROWS}")
    p.i(f"s_add_i32 s{S_NPAIRS}, s{S_NROWS}, 63")
    p.i(f"s_ashr_i32 s{S_NPAIRS}, s{S_NPAIRS}, 6")                        # ((nrows + 31) / 32 + 1) / 2 = (nrows + 63) / 64
    p.i(f"s_add_i32 s{S_NGROUPS}, s{S_NPAIRS}, 3")
    p.i(f"s_ashr_i32 s{S_NGROUPS}, s{S_NGROUPS}, 2")
    p.i(f"s_add_i32 s{S_LASTP}, s{S_NPAIRS}, -1")
    p.i(f"s_max_i32 s{S_LASTP}, s{S_LASTP}, 0")
    p.i(f"s_cmp_ge_i32 s{S_TG}, s{S_NGROUPS}")
    long_branch_scc1(p, ".Lnf_ha_done_%=")
    # ring: chunks 0 and 1 in place.  The refill is LDS-DMA (global_load_lds_dwordx4: L2 -> LDS without a register hop; destination =
    # M0 + instruction offset + lane * 16, the instruction offset also moves the source): every wave brings in its quarter (4 KB) of a chunk
    p.i(f"s_lshl_b32 s{S_WAVE4K}, s{S_WAVE}, 12")
    p.i(f"s_mov_b64 {sp(S_WCUR)}, {sp(S_WBASE)}")
    if DMA:
        for c in range(2):
            p.i(f"s_add_u32 m0, s{S_WAVE4K}, {c * CHUNK * 1024}")
            p.i("s_nop 0")
            for q in range(4):
                p.i(f"global_load_lds_dwordx4 v{V_PUB}, {sp(S_WCUR)} offset:{1024 * q}")
            p.i(f"s_add_u32 s{S_WCUR}, s{S_WCUR}, {CHUNK * 1024}")
            p.i(f"s_addc_u32 s{S_WCUR + 1}, s{S_WCUR + 1}, 0")
        p.i("s_waitcnt vmcnt(0)")
    else:
        for c in range(3):
            for q in range(4):
                p.i(f"global_load_dwordx4 {STAGE(q)}, v{V_PUB}, {sp(S_WCUR)} offset:{1024 * q}")
            p.i(f"s_add_u32 s{S_WCUR}, s{S_WCUR}, {CHUNK * 1024}")
            p.i(f"s_addc_u32 s{S_WCUR + 1}, s{S_WCUR + 1}, 0")
            if c < 2:
                p.i("s_waitcnt vmcnt(0)")
                for q in range(4):
                    p.i(f"ds_write_b128 v{V_PUB}, {STAGE(q)} offset:{c * CHUNK * 1024 + 1024 * q}")
                p.i("s_waitcnt lgkmcnt(0)")
    for q in range(4):
        p.vm.append(f"w{q}")
    # X of this wave's first pair -> stash (through bank 1's registers)
    emit_pair_select(p, S_TG, S_PAIR)
    emit_x_base(p, S_PAIR, S_XN)
    emit_x_base(p, S_PAIR, S_XNB, 16384)
    for half, base in ((0, S_XN), (1, S_XNB)):
        for t in range(XS):
            if t and t % 4 == 0:
                p.i(f"s_add_u32 s{base}, s{base}, 4096")
                p.i(f"s_addc_u32 s{base + 1}, s{base + 1}, 0")
            r = 4 * (half * XS + t)
            p.i(f"global_load_dwordx4 a[{r}:{r + 3}], v{V_LANE16}, {sp(base)} offset:{1024 * (t % 4)} nt")
    p.i("s_waitcnt vmcnt(0)")
    for i in range(2 * XS):
        p.i(f"ds_write_b128 v{V_STASH}, a[{4 * i}:{4 * i + 3}] offset:{1024 * i}")
    p.i("s_waitcnt lgkmcnt(0)")
    p.i("s_barrier")
    for s in range(PF):
        p.lds(f"ds_read_b128 {ASL(s)}, v{V_LANE16} offset:{1024 * s}", f"A{s}")
    for t in range(XPF):
        p.lds(f"ds_read_b128 {XR(0, t)}, v{V_STASH} offset:{1024 * t}", f"X{t}")          # (steps 0, 1 of the pair = K-steps 0, 1 of layer 0)
        p.lds(f"ds_read_b128 {XR(1, t)}, v{V_STASH} offset:{1024 * (XS + t)}", f"X{t}")
    p.i(f"s_mov_b64 {sp(S_BTCUR)}, {sp(S_BTBASE)}")
    for q in range(4):
        p.vmem(f"global_load_dwordx4 v[{BIASC + 4 * q}:{BIASC + 4 * q + 3}], v{V_H64}, {sp(S_BTCUR)} offset:{16 * q}", "b0")
    p.i(f"s_mov_b64 {sp(S_PEXA)}, 0")                  # no previous pair yet
    p.i(f"s_mov_b64 {sp(S_PEXB)}, 0")
    p.i(".Lnf_ha_pair_%=:")

    # ------------------------------------------------------------------ everything that is not an MFMA or an operand read
    # A `piece` is a short run of instructions that must stay together (a conversion piece, an SCC / VCC producer with its
    # consumers, a store + the load that refills its registers ...).  Pieces are queued in STREAMS (in-order inside a stream), each
    # with the first gap it may sit in and the last one; a gap is the shadow of one MFMA (index 2 * step + tile).  The placement
    # pass below fills every gap up to a budget of issue cycles (cost model: COST_*), most urgent deadline first.
    streams = {}

    def piece(stream, earliest, deadline, items):
        streams.setdefault(stream, []).append([earliest, deadline, items])

    def xbase(s_pair, s_out, extra=0):
        q_ = Prog()
        emit_x_base(q_, s_pair, s_out, extra)
        return q_.lines

    def psel(s_tg, s_pair):
        q_ = Prog()
        emit_pair_select(q_, s_tg, s_pair)
        return q_.lines

    # ---- per-pair scalar setup (S_PAIR = this pair on entry): every SCC / VCC producer sits in one piece with its consumers
    groups = [
        [f"s_lshl_b32 s{S_T1}, s{S_TG}, 2", f"s_add_u32 s{S_T1}, s{S_T1}, s{S_WAVE}", f"s_cmp_lt_i32 s{S_T1}, s{S_NPAIRS}",
         f"s_cselect_b32 s{S_OWNER}, -1, 0"],
        # rows of tile A / B held by the lanes h == 0
        [f"s_lshl_b32 s{S_T1}, s{S_PAIR}, 6", f"v_and_b32 v{V_ROW}, 31, v{V_ROW}", f"v_add_u32 v{V_ROW}, s{S_T1}, v{V_ROW}",
         f"v_add_u32 v{V_IDXOFF}, 32, v{V_ROW}"],
        [f"s_mov_b32 s{S_SAVE}, -1", f"s_mov_b32 s{S_SAVE + 1}, 0", f"v_cmp_gt_i32 vcc, s{S_NROWS}, v{V_ROW}",
         f"s_and_b64 {sp(S_EXA)}, vcc, {sp(S_SAVE)}", f"s_cmp_lg_u32 s{S_OWNER}, 0", f"s_cselect_b64 {sp(S_EXA)}, {sp(S_EXA)}, 0"],
        [f"v_cmp_gt_i32 vcc, s{S_NROWS}, v{V_IDXOFF}", f"s_and_b64 {sp(S_EXB)}, vcc, {sp(S_SAVE)}", f"s_cmp_lg_u32 s{S_OWNER}, 0",
         f"s_cselect_b64 {sp(S_EXB)}, {sp(S_EXB)}, 0"],
        # row_sample[rowA], row_sample[rowB] (clamped rows: the stores are masked)
        [f"s_add_i32 s{S_T1}, s{S_NROWS}, -1", f"v_min_i32 v{V_IDX[0]}, s{S_T1}, v{V_ROW}", f"v_min_i32 v{V_IDX[1]}, s{S_T1}, v{V_IDXOFF}",
         f"v_lshlrev_b32 v{V_IDX[0]}, 2, v{V_IDX[0]}", f"v_lshlrev_b32 v{V_IDX[1]}, 2, v{V_IDX[1]}"],
        [("vmem", f"global_load_dword v{V_IDX[0]}, v{V_IDX[0]}, {sp(S_RS)}", "idx")],
        [("vmem", f"global_load_dword v{V_IDX[1]}, v{V_IDX[1]}, {sp(S_RS)}", "idx")],
        # direction operands of this pair: K-steps 12..15 of both tiles
        xbase(S_PAIR, S_XD, 12288),
        xbase(S_PAIR, S_XDB, 16384 + 12288),
        # the next pair of this wave (clamped: its loads re-read valid rows and are simply not used); S_PAIR = NEXT pair from here on
        [f"s_add_u32 s{S_T1}, s{S_TG}, s{S_NBLK}"] + psel(S_T1, S_PAIR),
        xbase(S_PAIR, S_XN),
        xbase(S_PAIR, S_XNB, 16384),
    ]
    for i, g in enumerate(groups):
        piece("setup", 2 * (1 + i), 2 * 100, g)
    # ---- the previous pair's epilogue (sigmoid + store), spread over layer 0 and layer 1
    if not knob("NOEPI"):
        for g in epilogue_pieces():
            piece("epi", 2 * 4, 2 * 600, g)

    # ---- X of the next pair -> stash: batch t loaded from block t of layers 5-6 on, stored two blocks later (the stash is free
    # once the skip layer has run)
    bl = [b for b in blocks if b.name in ("L5", "L6")]
    for t in range(XS):
        if t >= 2:
            b2 = bl[t].first
            piece("xf", 2 * (b2 + 8), 2 * (b2 + 16), [("waitvm", f"xf{t - 2}"),
                                                       ("lds", f"ds_write_b128 v{V_STASH}, {XF(t, 0)} offset:{1024 * (t - 2)}", "st")])
            piece("xf", 2 * (b2 + 8), 2 * (b2 + 16), [("lds", f"ds_write_b128 v{V_STASH}, {XF(t, 1)} offset:{1024 * (XS + t - 2)}", "st")])
        b0 = bl[t].first
        piece("xf", 2 * (b0 + 9), 2 * (b0 + 17), [("vmem", f"global_load_dwordx4 {XF(t, 0)}, v{V_LANE16}, {sp(S_XN)} nt", f"xf{t}"),
                                                   f"s_add_u32 s{S_XN}, s{S_XN}, 1024", f"s_addc_u32 s{S_XN + 1}, s{S_XN + 1}, 0"])
        piece("xf", 2 * (b0 + 9), 2 * (b0 + 17), [("vmem", f"global_load_dwordx4 {XF(t, 1)}, v{V_LANE16}, {sp(S_XNB)} nt", f"xf{t}"),
                                                   f"s_add_u32 s{S_XNB}, s{S_XNB}, 1024", f"s_addc_u32 s{S_XNB + 1}, s{S_XNB + 1}, 0"])
    for t in (XS, XS + 1):                           # the last two batches' stores
        b2 = bl[t].first
        piece("xf", 2 * (b2 + 8), 2 * (b2 + 16), [("waitvm", f"xf{t - 2}"),
                                                   ("lds", f"ds_write_b128 v{V_STASH}, {XF(t, 0)} offset:{1024 * (t - 2)}", "st")])
        piece("xf", 2 * (b2 + 8), 2 * (b2 + 16), [("lds", f"ds_write_b128 v{V_STASH}, {XF(t, 1)} offset:{1024 * (XS + t - 2)}", "st")])
    # ---- direction operands: two loads per block in blocks 2..5 of layer 8
    l8 = [b for b in blocks if b.name == "L8"]
    for t in range(4):
        b0 = l8[2 + t].first
        piece("dir", 2 * (b0 + 9), 2 * (b0 + 17), [("vmem", f"global_load_dwordx4 {DIR(0, t)}, v{V_LANE16}, {sp(S_XD)} offset:{1024 * t} nt", f"d{t}")])
        piece("dir", 2 * (b0 + 9), 2 * (b0 + 17), [("vmem", f"global_load_dwordx4 {DIR(1, t)}, v{V_LANE16}, {sp(S_XDB)} offset:{1024 * t} nt", f"d{t}")])
    # ---- sigma: register 0 of the sigma block's accumulators, saved while view block 0 runs in the other buffer
    v0 = [b for b in blocks if b.name == "V"][0].first
    piece("sig", 2 * (v0 + 2), 2 * (v0 + 18), [f"v_mov_b32 v{V_SIG[0]}, v{acc_reg(0, 0, 0)}", f"v_mov_b32 v{V_SIG[1]}, v{acc_reg(0, 1, 0)}"])
    # ---- ring: behind the rendezvous in front of step 12 of chunk k, the third chunk k - 1 lived in is free: four LDS-DMA pieces bring
    # chunk k + 2 in; they are waited for (vmcnt) in front of the NEXT rendezvous, behind which chunk k + 2 is read
    for chunk in range(nchunks):
        third = (chunk + 2) % 3
        g0 = 2 * (CHUNK * chunk + 12)
        for qq in range(4):
            if DMA:
                if knob("NOFETCH"):
                    continue
                pre_ = [f"s_add_u32 m0, s{S_WAVE4K}, {third * CHUNK * 1024}", "s_nop 0"] if qq == 0 else []
                adv = []
                if qq == 3:
                    if (chunk + 3) % nchunks == 0:
                        adv = [f"s_mov_b64 {sp(S_WCUR)}, {sp(S_WBASE)}"]
                    else:
                        adv = [f"s_add_u32 s{S_WCUR}, s{S_WCUR}, {CHUNK * 1024}", f"s_addc_u32 s{S_WCUR + 1}, s{S_WCUR + 1}, 0"]
                piece("ring", g0, g0 + 16, pre_ + [("vmem", f"global_load_lds_dwordx4 v{V_PUB}, {sp(S_WCUR)} offset:{1024 * qq}", f"w{qq}")] + adv)
                continue
            if not knob("NOPUB"):
                piece("ring", g0, g0 + 14, [("waitvm", f"w{qq}")] + ([] if knob("NOPUBW") else
                                            [("lds", f"ds_write_b128 v{V_PUB}, {STAGE(qq)} offset:{third * CHUNK * 1024 + 1024 * qq}", "pub")]))
            if not knob("NOFETCH"):
                adv = []
                if qq == 3:
                    if (chunk + 4) % nchunks == 0:
                        adv = [f"s_mov_b64 {sp(S_WCUR)}, {sp(S_WBASE)}"]
                    else:
                        adv = [f"s_add_u32 s{S_WCUR}, s{S_WCUR}, {CHUNK * 1024}", f"s_addc_u32 s{S_WCUR + 1}, s{S_WCUR + 1}, 0"]
                piece("ring", g0, g0 + 16, [("vmem", f"global_load_dwordx4 {STAGE(qq)}, v{V_PUB}, {sp(S_WCUR)} offset:{1024 * qq}", f"w{qq}")] + adv)
    # ---- conversion pieces of every block's predecessor
    for bi, blk in enumerate(blocks):
        if blk.cvt is None or knob("NOCVT"):
            continue
        last = 2 * (blk.first + len(blk.steps) - 1) + 1
        prev, outb, pblk, relu = blk.cvt
        for pc in range(8):
            tile, q4 = pc >> 2, pc & 3
            k = 2 * pblk + (q4 >> 1)
            dl = last                                    # the next block's bias step overwrites the accumulators being converted
            if outb == blk.inb:                          # ... and this block itself reads what the piece writes (K-step k of its input bank)
                cons = [i for i, st in enumerate(blk.steps) if st == ("h", k)][0]
                dl = 2 * (blk.first + cons) - 1
            # tile A's last MFMA is two MFMAs old when the bias step's first one has issued; tile B's one more gap later
            piece(f"cvt{bi}", 2 * blk.first + (0 if tile == 0 else 2), dl, cvt_piece(pc, blk.cvt))

    # ---- the NEXT block's biases (16 floats per lane: the lane half's row of the table entry) into the C-operand registers, behind this
    # block's first two MFMAs (which read them)
    def bias_piece(j):
        adv = [f"s_mov_b64 {sp(S_BTCUR)}, {sp(S_BTBASE)}"] if j == 0 else [f"s_add_u32 s{S_BTCUR}, s{S_BTCUR}, 128", f"s_addc_u32 s{S_BTCUR + 1}, s{S_BTCUR + 1}, 0"]
        return [adv + [("vmem", f"global_load_dwordx4 v[{BIASC + 4 * q}:{BIASC + 4 * q + 3}], v{V_H64}, {sp(S_BTCUR)} offset:{16 * q}", f"b{j}") for q in range(2)],
                [("vmem", f"global_load_dwordx4 v[{BIASC + 4 * q}:{BIASC + 4 * q + 3}], v{V_H64}, {sp(S_BTCUR)} offset:{16 * q}", f"b{j}") for q in range(2, 4)]]

    for bi, blk in enumerate(blocks):
        j = (bi + 1) % len(blocks)
        last = 2 * (blk.first + len(blk.steps) - 1) + 1
        for pc_ in bias_piece(j):
            piece("bias", 2 * blk.first + 2, last - 2, pc_)

    # ------------------------------------------------------------------ placement
    C_VALU, C_LDSW, C_LDSR, C_VMEM, BUDGET = knob("C_VALU", 4), knob("C_LDSW", 26), knob("C_LDSR", 8), knob("C_VMEM", 16), knob("BUDGET", 24)

    def cost(items):
        c = 0
        for it in items:
            if isinstance(it, tuple):
                c += {"vmem": C_VMEM, "lds": C_LDSW if "ds_write" in it[1] else C_LDSR, "waitvm": 1}[it[0]]
            else:
                c += C_VALU if it.startswith("v_") else 1
        return c

    def reads_of(n):
        """the operand reads emitted in front of step n's first MFMA (they sit in the shadow of step n - 1's second MFMA)"""
        r = []
        m = (n + PF) % NSLOT
        if m < nreal and not knob("NOAREAD"):
            r.append((f"ds_read_b128 {ASL(m)}, v{V_LANE16} offset:{1024 * (m % RING)}", f"A{m}"))
        mx = (n + XPF) % NSLOT                           # the stash operands of the step XPF ahead (into the next pair's layer 0 at the very end)
        if mx < nreal and not knob("NOXS"):
            bx, sx = step_of[mx]
            kx, ix = bx.steps[sx]
            if kx == "xs":
                # (operand slot = the STEP's index mod 4, not the K-step's: the next block's K-step 0 is requested while this block's
                # K-step 12 — the same slot by K-step index — is still waiting for its MFMAs)
                r.append((f"ds_read_b128 {XR(0, mx)}, v{V_STASH} offset:{1024 * ix}", f"X{mx}"))
                r.append((f"ds_read_b128 {XR(1, mx)}, v{V_STASH} offset:{1024 * (XS + ix)}", f"X{mx}"))
        return r

    ngaps = 2 * nreal
    gaps = [[] for _ in range(ngaps)]
    heads = {k: 0 for k in streams}
    for G in range(ngaps):
        n, g = divmod(G, 2)
        used = 0
        if g == 1 and n + 1 < NSLOT:
            used = C_LDSR * len(reads_of(n + 1)) + 1
        while True:
            cand = []
            for k, lst in streams.items():
                h = heads[k]
                if h < len(lst) and lst[h][0] <= G:
                    cand.append((lst[h][1], k))
            if not cand:
                break
            cand.sort()
            placed = False
            for dl, k in cand:
                pc_ = streams[k][heads[k]]
                c = cost(pc_[2])
                if used + c <= BUDGET or (not gaps[G] and used <= C_LDSR + 1) or dl <= G:
                    assert dl >= G, f"piece of stream {k} misses its deadline ({dl} < {G})"
                    gaps[G] += pc_[2]
                    used += c
                    heads[k] += 1
                    placed = True
                    break
            if not placed:
                break
    for k, lst in streams.items():
        for pc_ in lst[heads[k]:]:
            assert pc_[1] >= ngaps, f"stream {k}: a piece with a deadline inside the pair was not placed"
    # what is left belongs to the padding steps: ring pieces of the last chunks, in order
    pad_items = {}
    for k, lst in streams.items():
        for pc_ in lst[heads[k]:]:
            pad_items.setdefault(max(pc_[0] // 2, nreal), []).extend(pc_[2])

    # ------------------------------------------------------------------ the steps
    bar = not knob("NOBAR")
    for n in range(NSLOT):
        real = n < nreal
        if n % CHUNK == 12:
            if DMA:
                p.wait_vm("w3")              # this wave's quarter of the chunk that becomes readable behind the rendezvous
            if bar:
                p.i("s_barrier")
        for text, tag in reads_of(n):
            p.lds(text, tag)
        if not real:
            if n == nreal:
                emit_handover(p)
            emit_items(p, pad_items.get(n, []))
            continue
        blk, s = step_of[n]
        kind, idx = blk.steps[s]
        p.wait_lds([f"A{n}"] + ([f"X{n}"] if kind == "xs" else []))
        if kind == "xd" and idx == 0 and blk.blk == 0:
            p.wait_vm("d3")
        if s == 0:
            p.wait_vm(f"b{blocks.index(blk)}")            # the block's biases: the C operand of its first two MFMAs
        for tile in range(2):
            if kind == "xs":
                bop = XR(tile, n)
            elif kind == "xd":
                bop = DIR(tile, idx)
            else:
                bop = bank_quad(blk.inb, tile, idx)
            acc = ACC(blk.acc, tile)
            p.i(f"v_mfma_f32_32x32x16_f16 {acc}, {ASL(n)}, {bop}, {f'v[{BIASC}:{BIASC + 15}]' if s == 0 else acc}")
            emit_items(p, gaps[2 * n + tile])

    # ------------------------------------------------------------------ loop back
    p.i(f"s_add_u32 s{S_TG}, s{S_TG}, s{S_NBLK}")
    p.i(f"s_cmp_lt_i32 s{S_TG}, s{S_NGROUPS}")
    long_branch_scc1(p, ".Lnf_ha_pair_%=")
    # the last pair's epilogue
    for g in epilogue_pieces():
        emit_items(p, g)
    p.i(".Lnf_ha_done_%=:")
    p.i("s_waitcnt vmcnt(0) lgkmcnt(0)")
    if knob("TIMING"):
        p.i("s_memtime s[92:93]")
        p.i("s_memrealtime s[94:95]")
        p.i("s_waitcnt lgkmcnt(0)")
        p.i("s_sub_u32 s92, s92, s90")
        p.i("s_sub_u32 s94, s94, s91")
        p.i(f"s_or_b32 s90, {IN['block']}, s{S_WAVE}")
        p.i("s_cmp_eq_u32 s90, 0")
        p.i("s_cbranch_scc0 .Lnf_ha_notime_%=")
        p.i("v_mov_b32 v0, s92")
        p.i("v_mov_b32 v1, s94")
        p.i(f"v_mov_b32 v2, {nreal * 1024}")
        p.i("s_mov_b64 exec, 1")
        p.i(f"global_store_dwordx2 v2, v[0:1], {sp(S_WBASE)}")
        p.i("s_waitcnt vmcnt(0)")
        p.i(".Lnf_ha_notime_%=:")
    return p.lines, NSLOT, nreal


def emit_items(p, items):
    for it in items:
        if isinstance(it, tuple):
            if it[0] == "vmem":
                p.vmem(it[1], it[2])
            elif it[0] == "lds":
                p.lds(it[1], it[2])
            elif it[0] == "waitvm":
                p.wait_vm(it[1])
        else:
            p.i(it)


def emit_handover(p):
    """behind the pair's last MFMA: what the deferred epilogue (inside the NEXT pair, or behind the loop) needs is moved out of the registers
    the next pair overwrites — the raw rgb head outputs (registers 0..2 of the rgb block's accumulators), sigma, the two store addresses
    (row_sample[row] * 16 + out) and the store masks."""
    p.i("s_nop 15")
    for tile in range(2):
        for c in range(3):
            p.i(f"v_mov_b32 v{V_PRGB[3 * tile + c]}, v{acc_reg(1, tile, c)}")
        p.i(f"v_mov_b32 v{V_PSIG[tile]}, v{V_SIG[tile]}")
    p.wait_vm("idx")
    for tile in range(2):
        a = V_PADDR[tile]
        p.i(f"v_ashrrev_i32 v{a + 1}, 31, v{V_IDX[tile]}")
        p.i(f"v_mov_b32 v{a}, v{V_IDX[tile]}")
        p.i(f"v_lshl_add_u64 v[{a}:{a + 1}], v[{a}:{a + 1}], 4, {sp(S_OUT)}")
    p.i(f"s_mov_b64 {sp(S_PEXA)}, {sp(S_EXA)}")
    p.i(f"s_mov_b64 {sp(S_PEXB)}, {sp(S_EXB)}")


def main():
    lines, nslot, nreal = gen()
    out = sys.argv[1] if len(sys.argv) > 1 else None
    text = "".join('"%s\\n"\n' % ln for ln in lines)
    if out:
        with open(out, "w") as f:
            f.write("// generated by gen_mlp_ha.py — do not edit\n")
            f.write("#define NF_HA_LDS_BYTES %d\n" % LDS_BYTES)
            f.write("#define NF_HA_SLOTS %d\n" % nslot)
            f.write("#define NF_HA_STEPS %d\n" % nreal)
            f.write(text)
    else:
        sys.stdout.write(text)
    sys.stderr.write("[gen_mlp_ha] %d instructions, %d MFMAs, %d + %d steps, LDS %d bytes\n" %
                     (len(lines), sum("v_mfma" in ln for ln in lines), nreal, nslot - nreal, LDS_BYTES))


if __name__ == "__main__":
    main()
