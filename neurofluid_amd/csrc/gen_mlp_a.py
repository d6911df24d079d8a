#!/usr/bin/env python3
"""Generator of the hand-scheduled instruction stream of k_mlp_fwd_a (nf_mlp_a.hip): the NeRF MLP forward
(/root/reference/models/nerf.py:83-124) on v_mfma_f32_32x32x2_f32, one 32-row tile per wave, the weight stream shared by the
four waves of a workgroup through an LDS ring — the arithmetic and the summation order of k_mlp_fwd_l (nf_mlp_l.hip), with
every instruction of the tile body placed by this script instead of by the compiler:

  * the whole tile (1 312 weight slots of 2 KB = 164 chunks of 8; no padding slots: the bias K-step of a layer is simply the
    slot after its last hidden K-step) is straight-line code, so ring positions, register names and wait counts are constants;
  * every non-MFMA instruction sits in the shadow of a named MFMA (a `gap`): the two ds_read_b128 that fetch the NEXT slot's
    A operands, the ReLU of the next B operand, one ring refill load per slot in the first four slots of a chunk, the
    rendezvous + publish on a chunk's last slot, the X loads / stash traffic, and the two VALU heads (sigma inside the view
    branch of the same tile, rgb + sigmoid + store inside layer 0 of the NEXT tile);
  * registers: accA v[0:127] (VGPR form, so the VALU reads it directly), accB a[0:127], view-branch accumulators a[128:191],
    two A-operand sets that alternate with the slot parity, two B-operand registers, two X quads, one staging set.

Output: a C string literal (one "...\\n" line per instruction) included by nf_mlp_a.hip as the body of ONE asm statement.
Usage: python gen_mlp_a.py [out.inc]      (neurofluid_amd/build.py runs it before compiling nf_mlp_a.hip)
"""
import os
import sys


def knob(name, default=0):
    """A/B switches of the schedule (tools/ab_mlp_a.py); the shipped kernel is the all-defaults build."""
    return int(os.environ.get("NF_A_" + name, default))


# feature groups of 8: (25, 7) = the default 198 + 54 row; the other encodings of models/renderer.py:30-44 give qx in {8, 9, 16, 17, 24, 25},
# qd in {4, 7} (argv[2], argv[3]; nf_mlp_a.hip instantiates all twelve)
QX, QD = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (25, 7)
assert 1 <= QX <= 25 and 1 <= QD <= 7
CH = 8                         # slots per chunk
SLOT_B, CHUNK_B = 2048, 16384
# PIPEPUB (default): the four 1 KB pieces a wave owns of chunk k + 1 are published ONE PER SLOT in slots 0..3 of chunk k (each
# behind the load that brought it in during chunk k - 1, and in front of the load that refills its staging quad with chunk
# k + 2), instead of four stores per wave right behind the rendezvous.  The chunk being published is the NEXT one, so two
# chunks of LDS are enough, the ring phase is the chunk parity (164 chunks per tile: even) and no address register rotates.
PIPEPUB = knob("PIPEPUB", 1)
RING_B = (2 if PIPEPUB else 3) * CHUNK_B
STASH_B = QX * 1024            # per wave
HW_BASE = RING_B + 4 * STASH_B  # head weights in LDS: sigma [2][128] floats, then rgb [3][2][64]
LDS_BYTES = HW_BASE + 1024 + 1536

# ---- inputs of the asm statement (operand numbers in nf_mlp_a.hip) -------------------------------------------------
IN = dict(wstream="%0", wsig="%1", wrgb="%2", bias="%3", X="%4", n_rows="%5", row_sample="%6", out="%7", max_rows="%8",
          wave="%9", block="%10", nblocks="%11")

# ---- scalar registers -------------------------------------------------------------------------------------------------
_s = 40


def _salloc(n=1, align=1):
    global _s
    _s = (_s + align - 1) // align * align
    r = _s
    _s += n
    return r


S_WBASE = _salloc(2, 2); S_WCUR = _salloc(2, 2); S_X = _salloc(2, 2); S_XTILE = _salloc(2, 2); S_XNEXT = _salloc(2, 2)
S_XBASE = _salloc(2, 2); S_RS = _salloc(2, 2); S_OUT = _salloc(2, 2); S_TMP = _salloc(2, 2); S_PEXEC = _salloc(2, 2)
S_CEXEC = _salloc(2, 2); S_SAVE = _salloc(2, 2); S_T2 = _salloc(2, 2)
S_NROWS = _salloc(); S_NTILES = _salloc(); S_NGROUPS = _salloc(); S_TG = _salloc(); S_WAVE = _salloc(); S_TILE = _salloc()
S_NBLK = _salloc(); S_BSIG = _salloc(); S_BRGB = _salloc(3); S_NL2E = _salloc(); S_EHI = _salloc(); S_ELO = _salloc()
S_TN = _salloc()
S_LAST = _s
assert S_LAST <= 100


def sp(r):
    return f"s[{r}:{r + 1}]"


# ---- vector registers ------------------------------------------------------------------------------------------------
ACC_A = lambda b: f"v[{16 * b}:{16 * b + 15}]"          # noqa: E731
ACC_B = lambda b: f"a[{16 * b}:{16 * b + 15}]"          # noqa: E731
HD = lambda b: f"a[{128 + 16 * b}:{128 + 16 * b + 15}]"  # noqa: E731
P = [128, 136]                  # A-operand sets: [set] + 0..3 = blocks 0-3 (or first half-step), + 4..7 = blocks 4-7 (second)
VB = [144, 145]                 # ReLU'd B operand of the current / next K-step
V_ONE = 146
XV = [148, 152]                 # X quads
ST = 156                        # 16 staging registers (this wave's quarter of the chunk in flight)
V_LANE16, V_T0, V_T1, V_T2, V_WAVE4K, V_WOFF, V_STASH, V_PUB, V_TMP = 172, 173, 174, 175, 176, 177, 178, 179, 180
V_HS, V_HR, V_BP = 181, 182, 183       # head-weight LDS addresses (sigma / rgb, incl. the lane's half), bpermute address (lane ^ 32)
V_SIG = 184                     # sigma partial -> sigma of the tile
V_C = [185, 186, 187]           # rgb partials
V_PSIG, V_PROW = 188, 189       # previous tile: sigma, row index (the rgb head of a tile runs inside the next tile's layer 0)
V_ROW = 190
HWQ = [192, 196]                # head-weight quads (sigma), double-buffered
HWR = [200, 212]                # rgb: 3 quads per set
DG = [148, 152, 200, 204, 208, 212, 216]      # direction-feature quads of the view branch (7 groups)
V_H = [224, 225]                # head temporaries (value, product)
V_E = list(range(226, 244))     # sigmoid temporaries
V_IDX = 244                     # row_sample[row] of the previous tile
V_ADDR = 246                    # 64-bit store address (2 registers)
V_O = 248                       # output quad


class Prog:
    def __init__(self):
        self.lines = []
        self.vm = []            # outstanding vector-memory operations, in issue order (tags)

    def i(self, s):
        self.lines.append(s)

    def vmem(self, text, tag):
        self.i(text)
        self.vm.append(tag)
        assert len(self.vm) < 60, "vmcnt is a 6-bit counter"

    def wait_vm(self, tag):
        """Instruction text that waits for the LAST operation tagged `tag` (and, vmcnt being an in-order count, everything
        issued before it) — or None when nothing with that tag is outstanding."""
        if tag not in self.vm:
            return None
        idx = len(self.vm) - 1 - self.vm[::-1].index(tag)
        n_after = len(self.vm) - 1 - idx
        self.vm = self.vm[idx + 1:]
        return f"s_waitcnt vmcnt({n_after})"


_lb = [0]


def long_branch_scc1(p, target):
    """s_cbranch_scc1 over more than the 16-bit word offset of a branch (the tile body is ~150 KB): the compiler's own
    long-branch sequence — pc of the next instruction + (target - that) — behind an inverted short branch."""
    n = _lb[0]
    _lb[0] += 1
    p.i(f"s_cbranch_scc0 .Lnf_a_skip{n}_%=")
    p.i(f"s_getpc_b64 {sp(S_TMP)}")
    p.i(f".Lnf_a_pc{n}_%=:")
    p.i(f"s_add_u32 s{S_TMP}, s{S_TMP}, ({target}-.Lnf_a_pc{n}_%=)&4294967295")
    p.i(f"s_addc_u32 s{S_TMP + 1}, s{S_TMP + 1}, ({target}-.Lnf_a_pc{n}_%=)>>32")
    p.i(f"s_setpc_b64 {sp(S_TMP)}")
    p.i(f".Lnf_a_skip{n}_%=:")


VBR = list(range(192, 200)) + list(range(226, 234))      # B-operand registers of the hidden parts: the sigma head's weight quads
#                                                           and the sigmoid temporaries, both idle between layer 0 and the view branch


def vb_reg(k):
    rb = knob("RELU_BATCH", 4)
    return VBR[((k // rb) & 1) * rb + k % rb]


def mfma(dst, a, b, c):
    return f"v_mfma_f32_32x32x2_f32 {dst}, v{a}, {b}, {c}"


# ------------------------------------------------------------------------------------------------------------------------
# the slot list of one tile
# ------------------------------------------------------------------------------------------------------------------------
class Slot:
    """One 2 KB slot of the weight stream = one K-step of an 8-block layer ("x": X operand, "h": hidden operand, "b": bias), two half-steps of the
    4-block view branch ("vx", "vh", "vb") or a padding slot ("pad"); `k` = its index inside its part, `idx` = its index inside the tile."""
    __slots__ = ("kind", "layer", "k", "idx")

    def __init__(self, kind, layer, k):
        self.kind, self.layer, self.k = kind, layer, k


def layer_dst(layer):
    """Accumulator set the 8-block layer `layer` writes: 0 -> accA, 1 -> accB, 2 -> accA, ... (its hidden operand is the other set)."""
    return ACC_A if layer % 2 == 0 else ACC_B


def build_tile():
    slots = []
    for layer in range(9):
        if layer in (0, 4):
            for k in range(4 * QX):
                slots.append(Slot("x", layer, k))
        if layer > 0:
            for k in range(128):
                slots.append(Slot("h", layer, k))
        slots.append(Slot("b", layer, 0))
    for k in range(2 * QD):
        slots.append(Slot("vx", 9, k))
    for k in range(64):
        slots.append(Slot("vh", 9, k))
    slots.append(Slot("vb", 9, 0))
    assert len(slots) == 8 * QX + 2 * QD + 1098
    # the ring's phase is the chunk parity and the A-operand sets alternate with the slot parity: a tile is a whole, EVEN number of chunks.
    # Feature rows that do not give one are padded with slots that hold no MFMA (ring bookkeeping only; the stream carries zeros there)
    while len(slots) % (2 * CH):
        slots.append(Slot("pad", 10, 0))
    for n, s in enumerate(slots):
        s.idx = n
    return slots


def gen():
    p = Prog()
    slots = build_tile()
    NS = len(slots)
    nchunks = NS // CH

    # ------------------------------------------------------------------ prologue
    p.i(f"s_mov_b64 {sp(S_WBASE)}, {IN['wstream']}")
    p.i(f"s_mov_b64 {sp(S_XBASE)}, {IN['X']}")
    p.i(f"s_mov_b64 {sp(S_RS)}, {IN['row_sample']}")
    p.i(f"s_mov_b64 {sp(S_OUT)}, {IN['out']}")
    p.i(f"s_mov_b32 s{S_WAVE}, {IN['wave']}")
    p.i(f"s_mov_b32 s{S_TG}, {IN['block']}")
    p.i(f"s_mov_b32 s{S_NBLK}, {IN['nblocks']}")
    p.i(f"s_mov_b32 s{S_NROWS}, {IN['max_rows']}")
    p.i(f"s_load_dword s{S_TMP}, {IN['n_rows']}, 0x0")
    p.i(f"s_load_dword s{S_BSIG}, {IN['bias']}, 0x0")
    p.i(f"s_load_dword s{S_BRGB}, {IN['bias']}, 0x10")           # NfMlpLayout: off_brgb = off_bsig + 4 floats
    p.i(f"s_load_dword s{S_BRGB + 1}, {IN['bias']}, 0x14")
    p.i(f"s_load_dword s{S_BRGB + 2}, {IN['bias']}, 0x18")
    p.i(f"s_mov_b64 {sp(S_T2)}, {IN['wsig']}")
    p.i(f"s_mov_b64 {sp(S_SAVE)}, {IN['wrgb']}")
    p.i(f"s_mov_b32 s{S_NL2E}, 0xbfb8aa3b")
    p.i(f"s_mov_b32 s{S_EHI}, 0x42ce8ed0")
    p.i(f"s_mov_b32 s{S_ELO}, 0xc2b17218")
    # lane ids
    p.i(f"v_mbcnt_lo_u32_b32 v{V_E[3]}, -1, 0")
    p.i(f"v_mbcnt_hi_u32_b32 v{V_E[3]}, -1, v{V_E[3]}")           # lane 0..63
    p.i(f"v_lshlrev_b32 v{V_LANE16}, 4, v{V_E[3]}")
    p.i(f"v_mov_b32 v{V_ONE}, 1.0")
    p.i(f"v_xor_b32 v{V_BP}, 32, v{V_E[3]}")
    p.i(f"v_lshlrev_b32 v{V_BP}, 2, v{V_BP}")                     # ds_bpermute address of lane ^ 32
    p.i(f"v_lshrrev_b32 v{V_E[3 + 1]}, 5, v{V_E[3]}")               # h
    p.i(f"v_lshlrev_b32 v{V_HS}, 9, v{V_E[3 + 1]}")                # h * 512 B
    p.i(f"v_add_u32 v{V_HS}, {HW_BASE}, v{V_HS}")
    p.i(f"v_lshlrev_b32 v{V_HR}, 8, v{V_E[3 + 1]}")                # h * 256 B
    p.i(f"v_add_u32 v{V_HR}, {HW_BASE + 1024}, v{V_HR}")
    p.i(f"v_and_b32 v{V_ROW}, 31, v{V_E[3]}")                      # j = lane & 31 (row inside the tile)
    p.i(f"s_lshl_b32 s{S_TMP + 1}, s{S_WAVE}, 12")
    p.i(f"v_mov_b32 v{V_WAVE4K}, s{S_TMP + 1}")
    p.i(f"v_add_u32 v{V_WOFF}, v{V_WAVE4K}, v{V_LANE16}")         # wave * 4096 + lane * 16
    p.i(f"s_mul_i32 s{S_TMP + 1}, s{S_WAVE}, {STASH_B}")
    p.i(f"s_add_u32 s{S_TMP + 1}, s{S_TMP + 1}, {RING_B}")
    p.i(f"v_add_u32 v{V_STASH}, s{S_TMP + 1}, v{V_LANE16}")
    p.i(f"v_mov_b32 v{V_T0}, v{V_LANE16}")
    p.i(f"v_add_u32 v{V_T1}, {CHUNK_B}, v{V_LANE16}")
    p.i(f"v_add_u32 v{V_T2}, {2 * CHUNK_B}, v{V_LANE16}")
    # head weights -> LDS: thread t (= wave * 64 + lane) of 256
    p.i(f"s_lshl_b32 s{S_TMP + 1}, s{S_WAVE}, 6")
    p.i(f"v_add_u32 v{V_E[3 + 1]}, s{S_TMP + 1}, v{V_E[3]}")        # t
    # sigma: LDS float t = h * 128 + k  <-  wsig[k * 2 + h]
    p.i(f"v_and_b32 v{V_E[3 + 2]}, 127, v{V_E[3 + 1]}")
    p.i(f"v_lshrrev_b32 v{V_E[3 + 3]}, 7, v{V_E[3 + 1]}")
    p.i(f"v_lshl_add_u32 v{V_E[3 + 2]}, v{V_E[3 + 2]}, 1, v{V_E[3 + 3]}")
    p.i(f"v_lshlrev_b32 v{V_E[3 + 2]}, 2, v{V_E[3 + 2]}")
    p.i(f"global_load_dword v{V_E[0]}, v{V_E[3 + 2]}, {sp(S_T2)}")
    # rgb: LDS float i = c * 128 + h * 64 + k (k < 64)  <-  wrgb[c * 128 + k * 2 + h],  i = t and t + 256 (< 384)
    for rep in range(2):
        p.i(f"v_add_u32 v{V_E[3 + 4]}, {256 * rep}, v{V_E[3 + 1]}")                 # i
        p.i(f"v_and_b32 v{V_E[3 + 5]}, 63, v{V_E[3 + 4]}")                           # k
        p.i(f"v_bfe_u32 v{V_E[3 + 6]}, v{V_E[3 + 4]}, 6, 1")                         # h
        p.i(f"v_lshrrev_b32 v{V_E[3 + 7]}, 7, v{V_E[3 + 4]}")                        # c
        p.i(f"v_lshl_add_u32 v{V_E[3 + 5]}, v{V_E[3 + 5]}, 1, v{V_E[3 + 6]}")         # k * 2 + h
        p.i(f"v_lshl_add_u32 v{V_E[3 + 5]}, v{V_E[3 + 7]}, 7, v{V_E[3 + 5]}")         # + c * 128
        p.i(f"v_min_u32 v{V_E[3 + 5]}, 383, v{V_E[3 + 5]}")
        p.i(f"v_lshlrev_b32 v{V_E[3 + 5]}, 2, v{V_E[3 + 5]}")
        p.i(f"global_load_dword v{V_E[1 + rep]}, v{V_E[3 + 5]}, {sp(S_SAVE)}")
    p.i(f"v_lshlrev_b32 v{V_E[3 + 2]}, 2, v{V_E[3 + 1]}")                            # t * 4
    p.i(f"v_add_u32 v{V_E[3 + 2]}, {HW_BASE}, v{V_E[3 + 2]}")                        # (a ds offset is 16 bits: the base rides in the address)
    p.i("s_waitcnt vmcnt(0) lgkmcnt(0)")
    p.i(f"ds_write_b32 v{V_E[3 + 2]}, v{V_E[0]}")
    p.i(f"ds_write_b32 v{V_E[3 + 2]}, v{V_E[1]} offset:1024")
    p.i(f"v_cmp_gt_u32 vcc, 128, v{V_E[3 + 1]}")
    p.i(f"s_and_saveexec_b64 {sp(S_SAVE)}, vcc")
    p.i(f"ds_write_b32 v{V_E[3 + 2]}, v{V_E[2]} offset:2048")
    p.i(f"s_mov_b64 exec, {sp(S_SAVE)}")
    # rows / tiles / tile groups
    p.i(f"s_min_i32 s{S_NROWS}, s{S_TMP}, s{S_NROWS}")
    p.i(f"s_add_i32 s{S_NTILES}, s{S_NROWS}, 31")
    p.i(f"s_ashr_i32 s{S_NTILES}, s{S_NTILES}, 5")
    p.i(f"s_add_i32 s{S_NGROUPS}, s{S_NTILES}, 3")
    p.i(f"s_ashr_i32 s{S_NGROUPS}, s{S_NGROUPS}, 2")
    # ring: chunks 0 and 1 in place, chunk 2 is fetched during chunk 0
    p.i(f"s_mov_b64 {sp(S_WCUR)}, {sp(S_WBASE)}")
    for c in range(1 if PIPEPUB else 2):
        for q in range(4):
            p.i(f"global_load_dwordx4 v[{ST + 4 * q}:{ST + 4 * q + 3}], v{V_WOFF}, {sp(S_WCUR)} offset:{1024 * q}")
        p.i(f"s_add_u32 s{S_WCUR}, s{S_WCUR}, {CHUNK_B}")
        p.i(f"s_addc_u32 s{S_WCUR + 1}, s{S_WCUR + 1}, 0")
        p.i("s_waitcnt vmcnt(0)")
        p.i(f"v_add_u32 v{V_PUB}, v{V_T0 + c}, v{V_WAVE4K}")
        for q in range(4):
            p.i(f"ds_write_b128 v{V_PUB}, v[{ST + 4 * q}:{ST + 4 * q + 3}] offset:{1024 * q}")
    p.i("s_waitcnt lgkmcnt(0)")
    p.i("s_barrier")
    p.i(f"s_mov_b64 {sp(S_PEXEC)}, 0")                 # no previous tile yet
    p.i(f"v_mov_b32 v{V_PROW}, 0")
    p.i(f"v_mov_b32 v{V_PSIG}, 0")
    p.i(f"s_cmp_ge_i32 s{S_TG}, s{S_NGROUPS}")
    long_branch_scc1(p, ".Lnf_a_done_%=")
    # first tile of this wave: X base, first X group in flight, first A operands
    emit_tile_select(p, S_TG, S_XNEXT)
    p.i(f"ds_read_b128 v[{P[0]}:{P[0] + 3}], v{V_T0}")
    p.i(f"ds_read_b128 v[{P[0] + 4}:{P[0] + 7}], v{V_T0} offset:1024")
    p.vmem(f"global_load_dwordx4 v[{XV[0]}:{XV[0] + 3}], v{V_LANE16}, {sp(S_XNEXT)} nt", "x0")
    if PIPEPUB:                 # chunk 1 rides in the staging quads (published during chunk 0), as at every later tile start
        for q in range(4):
            p.vmem(f"global_load_dwordx4 v[{ST + 4 * q}:{ST + 4 * q + 3}], v{V_WOFF}, {sp(S_WCUR)} offset:{1024 * q}", f"w{q}")
        p.i(f"s_add_u32 s{S_WCUR}, s{S_WCUR}, {CHUNK_B}")
        p.i(f"s_addc_u32 s{S_WCUR + 1}, s{S_WCUR + 1}, 0")
    p.i(".Lnf_a_tile_%=:")

    # ------------------------------------------------------------------ per-tile scalar setup
    # current tile = what emit_tile_select left in S_TILE / S_XNEXT; row and exec mask of its store
    p.i(f"s_mov_b64 {sp(S_XTILE)}, {sp(S_XNEXT)}")
    p.i(f"s_lshl_b32 s{S_TMP}, s{S_TG}, 2")
    p.i(f"s_add_u32 s{S_TMP}, s{S_TMP}, s{S_WAVE}")                       # tg * 4 + wave (the tile this wave OWNS, if < ntiles)
    p.i(f"s_lshl_b32 s{S_TMP + 1}, s{S_TILE}, 5")
    p.i(f"v_add_u32 v{V_ROW}, s{S_TMP + 1}, v{V_ROW}")                    # row = tile * 32 + j   (V_ROW held j)
    p.i(f"v_cmp_gt_i32 vcc, s{S_NROWS}, v{V_ROW}")                        # row < nrows
    p.i(f"s_mov_b32 s{S_SAVE}, -1")                                        # lanes 0..31 (h == 0) hold the tile's 32 rows
    p.i(f"s_mov_b32 s{S_SAVE + 1}, 0")
    p.i(f"s_and_b64 {sp(S_CEXEC)}, vcc, {sp(S_SAVE)}")
    p.i(f"s_cmp_lt_i32 s{S_TMP}, s{S_NTILES}")                            # owner
    p.i(f"s_cselect_b64 {sp(S_CEXEC)}, {sp(S_CEXEC)}, 0")

    # ------------------------------------------------------------------ fillers
    rgb_items = rgb_head_items()
    sig_items = sigma_head_items()
    fill_plan = {}                                   # slot idx -> list of (gap, [instr])
    skipb = knob("FILL_SKIP_BOUNDARY")
    ok = lambda s: not (skipb and s.idx % CH == CH - 1)        # noqa: E731
    if not knob("NOHEADS"):
        # (layer 1's slots take what a narrow feature row's layer 0 cannot hold: the head's registers are free until layer 8)
        place_fillers(slots, [s.idx for s in slots if ((s.kind == "x" and s.layer == 0) or (s.kind == "h" and s.layer == 1)) and ok(s)],
                      rgb_items, fill_plan, p)
        place_fillers(slots, [s.idx for s in slots if s.kind in ("vx", "vh") and ok(s)][:-2], sig_items, fill_plan, p)

    # ------------------------------------------------------------------ the slots
    for n, s in enumerate(slots):
        emit_slot(p, slots, n, fill_plan.get(n, []), nchunks)

    # ------------------------------------------------------------------ loop back
    p.i(f"s_add_u32 s{S_TG}, s{S_TG}, s{S_NBLK}")
    # carry this tile's store state over to the next tile's layer 0 (or to the epilogue)
    p.i(f"s_mov_b64 {sp(S_PEXEC)}, {sp(S_CEXEC)}")
    p.i(f"v_mov_b32 v{V_PROW}, v{V_ROW}")
    p.i(f"v_mov_b32 v{V_PSIG}, v{V_SIG}")
    p.i(f"v_and_b32 v{V_ROW}, 31, v{V_ROW}")                               # back to j
    p.i(f"s_cmp_lt_i32 s{S_TG}, s{S_NGROUPS}")
    long_branch_scc1(p, ".Lnf_a_tile_%=")
    # ------------------------------------------------------------------ epilogue: the last tile's rgb head
    p.i("s_nop 15")
    p.i("s_nop 7")
    for it in rgb_items:
        for ins in it[1]:
            p.i(ins if not callable(ins) else ins(p))
        if it[0] == "lds":
            p.i("s_waitcnt lgkmcnt(0)")
    p.i(".Lnf_a_done_%=:")
    p.i("s_waitcnt vmcnt(0) lgkmcnt(0)")
    return p.lines


def emit_tile_select(p, s_tg, s_xout):
    """S_TILE = min(tg * 4 + wave, ntiles - 1); s_xout = X + tile * 32 KB."""
    p.i(f"s_lshl_b32 s{S_TILE}, s{s_tg}, 2")
    p.i(f"s_add_u32 s{S_TILE}, s{S_TILE}, s{S_WAVE}")
    p.i(f"s_add_i32 s{S_TMP}, s{S_NTILES}, -1")
    p.i(f"s_min_i32 s{S_TILE}, s{S_TILE}, s{S_TMP}")
    p.i(f"s_mul_hi_u32 s{S_TMP + 1}, s{S_TILE}, {(QX + QD) * 1024}")      # tile * (QX + QD) groups * 1 KB   (high word)
    p.i(f"s_mul_i32 s{S_TMP}, s{S_TILE}, {(QX + QD) * 1024}")             # (low word)
    p.i(f"s_add_u32 s{s_xout}, s{S_XBASE}, s{S_TMP}")
    p.i(f"s_addc_u32 s{s_xout + 1}, s{S_XBASE + 1}, s{S_TMP + 1}")


# ------------------------------------------------------------------------------------------------------------------------
# heads as filler items: ("lds" | "valu", [instructions])
# ------------------------------------------------------------------------------------------------------------------------
def sigma_head_items():
    """sigma = sum_{b, r} relu(accB[b][r]) * wsig[(b, r), h]  (mul, add: the order and rounding of k_mlp_fwd_l), then
    + the other half's partial + bias.  accB lives in AGPRs: v_accvgpr_read first."""
    items = []
    items.append(("lds", [f"ds_read_b128 v[{HWQ[0]}:{HWQ[0] + 3}], v{V_HS} offset:0"]))
    items.append(("valu", [f"v_mov_b32 v{V_SIG}, 0"]))
    items.append(("next", []))
    for q in range(32):
        b, r0 = divmod(4 * q, 16)
        if q + 1 < 32:
            items.append(("lds", [f"ds_read_b128 v[{HWQ[(q + 1) & 1]}:{HWQ[(q + 1) & 1] + 3}], v{V_HS} offset:{16 * (q + 1)}"]))
        for e in range(4):
            k = 16 * b + r0 + e
            items.append(("valu", [f"v_accvgpr_read_b32 v{V_H[0]}, a{k}",
                                   f"v_max_f32 v{V_H[0]}, 0, v{V_H[0]}",
                                   f"v_mul_f32 v{V_H[1]}, v{V_H[0]}, v{HWQ[q & 1] + e}",
                                   f"v_add_f32 v{V_SIG}, v{V_SIG}, v{V_H[1]}"]))
        items.append(("next", []))
    items.append(("lds", [f"ds_bpermute_b32 v{V_H[0]}, v{V_BP}, v{V_SIG}"]))
    items.append(("next", []))
    items.append(("valu", [f"v_add_f32 v{V_SIG}, v{V_SIG}, v{V_H[0]}", f"v_add_f32 v{V_SIG}, s{S_BSIG}, v{V_SIG}"]))
    return items


def rgb_head_items():
    """rgb of the PREVIOUS tile from hd = a[128:191]: c = sum relu(hd) * wrgb (mul, add), + other half + bias, sigmoid as the
    compiler expands 1 / (1 + expf(-c)) (same instruction sequence: bit-identical to k_mlp_fwd_l), store under S_PEXEC."""
    items = []
    items.append(("lds", [f"ds_read_b128 v[{HWR[0] + 4 * c}:{HWR[0] + 4 * c + 3}], v{V_HR} offset:{512 * c}" for c in range(3)]))
    items.append(("valu", [f"v_mov_b32 v{V_C[c]}, 0" for c in range(3)]))
    # row_sample[prev row] (clamped row: idle waves read a valid index; the store is masked)
    items.append(("valu", [f"v_lshlrev_b32 v{V_IDX}, 2, v{V_PROW}"]))
    items.append(("vmem", [lambda p: p_vmem(p, f"global_load_dword v{V_IDX}, v{V_IDX}, {sp(S_RS)}", "idx")]))
    items.append(("next", []))
    for q in range(16):
        b, r0 = divmod(4 * q, 16)
        if q + 1 < 16:
            s_ = (q + 1) & 1
            items.append(("lds", [f"ds_read_b128 v[{HWR[s_] + 4 * c}:{HWR[s_] + 4 * c + 3}], v{V_HR} offset:{512 * c + 16 * (q + 1)}"
                                  for c in range(3)]))
        for e in range(4):
            k = 128 + 16 * b + r0 + e
            ins = [f"v_accvgpr_read_b32 v{V_H[0]}, a{k}", f"v_max_f32 v{V_H[0]}, 0, v{V_H[0]}"]
            for c in range(3):
                ins += [f"v_mul_f32 v{V_H[1]}, v{V_H[0]}, v{HWR[q & 1] + 4 * c + e}", f"v_add_f32 v{V_C[c]}, v{V_C[c]}, v{V_H[1]}"]
            items.append(("valu", ins[:4]))
            items.append(("valu", ins[4:]))
        items.append(("next", []))
    items.append(("lds", [f"ds_bpermute_b32 v{V_E[c]}, v{V_BP}, v{V_C[c]}" for c in range(3)]))
    items.append(("next", []))
    E = V_E
    for c in range(3):
        x, t, ph, r, ri, e = V_C[c], E[3], E[4], E[5], E[6], E[7]
        items.append(("valu", [f"v_add_f32 v{x}, v{x}, v{E[c]}", f"v_add_f32 v{x}, s{S_BRGB + c}, v{x}",
                               f"v_mul_f32 v{t}, 0xbfb8aa3b, v{x}", f"v_fma_f32 v{ph}, v{x}, s{S_NL2E}, -v{t}"]))
        items.append(("valu", [f"v_rndne_f32 v{r}, v{t}", f"v_fmac_f32 v{ph}, 0xb2a5705f, v{x}", f"v_sub_f32 v{t}, v{t}, v{r}",
                               f"v_add_f32 v{t}, v{t}, v{ph}"]))
        items.append(("valu", [f"v_cvt_i32_f32 v{ri}, v{r}", f"v_exp_f32 v{e}, v{t}", "s_nop 0", f"v_ldexp_f32 v{e}, v{e}, v{ri}"]))
        items.append(("valu", [f"v_cmp_nlt_f32 vcc, s{S_EHI}, v{x}", "s_nop 1", f"v_cndmask_b32 v{e}, 0, v{e}, vcc",
                               f"v_mov_b32 v{E[8]}, 0x7f800000"]))
        items.append(("valu", [f"v_cmp_ngt_f32 vcc, s{S_ELO}, v{x}", "s_nop 1", f"v_cndmask_b32 v{e}, v{E[8]}, v{e}, vcc",
                               f"v_add_f32 v{e}, 1.0, v{e}"]))
        d, ds, rc, e0, ns, q_, e1 = e, E[9], E[10], E[11], E[12], E[13], E[14]
        items.append(("valu", [f"v_div_scale_f32 v{ds}, {sp(S_TMP)}, v{d}, v{d}, 1.0", f"v_rcp_f32 v{rc}, v{ds}", "s_nop 0",
                               f"v_fma_f32 v{e0}, -v{ds}, v{rc}, 1.0"]))
        items.append(("valu", [f"v_div_scale_f32 v{ns}, vcc, 1.0, v{d}, 1.0", f"v_fmac_f32 v{rc}, v{e0}, v{rc}",
                               f"v_mul_f32 v{q_}, v{ns}, v{rc}", f"v_fma_f32 v{e1}, -v{ds}, v{q_}, v{ns}"]))
        items.append(("valu", [f"v_fmac_f32 v{q_}, v{e1}, v{rc}", f"v_fma_f32 v{e1}, -v{ds}, v{q_}, v{ns}", "s_nop 1",
                               f"v_div_fmas_f32 v{e1}, v{e1}, v{rc}, v{q_}"]))
        items.append(("valu", [f"v_div_fixup_f32 v{V_O + c}, v{e1}, v{d}, 1.0"]))
    items.append(("valu", [f"v_mov_b32 v{V_O + 3}, v{V_PSIG}"]))
    items.append(("wait", [lambda p: p.wait_vm("idx") or "s_nop 0"]))
    items.append(("valu", [f"v_ashrrev_i32 v{V_ADDR + 1}, 31, v{V_IDX}", f"v_mov_b32 v{V_ADDR}, v{V_IDX}",
                           f"v_lshl_add_u64 v[{V_ADDR}:{V_ADDR + 1}], v[{V_ADDR}:{V_ADDR + 1}], 4, {sp(S_OUT)}"]))
    items.append(("vmem", [f"s_mov_b64 {sp(S_SAVE)}, exec", f"s_mov_b64 exec, {sp(S_PEXEC)}",
                           lambda p: p_vmem(p, f"global_store_dwordx4 v[{V_ADDR}:{V_ADDR + 1}], v[{V_O}:{V_O + 3}], off", "st"),
                           "s_nop 1", f"s_mov_b64 exec, {sp(S_SAVE)}"]))
    return items


def p_vmem(p, text, tag):
    p.vm.append(tag)
    assert len(p.vm) < 60
    return text


def place_fillers(slots, idxs, items, plan, p):
    """Spread `items` over the slots `idxs`, in order.  An LDS item goes into gap 2 or 3 of a slot (its result is complete at the
    next slot's lgkmcnt(0)); "next" moves on to the next slot (the items behind it consume an LDS result); VALU items fill gaps
    4..7, at most two items (<= 8 instructions) per gap."""
    it = iter(idxs)
    cur = next(it)
    used = {}               # (slot, gap) -> count

    def take(kind):
        nonlocal cur
        while True:
            gaps = (2, 3) if kind == "lds" else (4, 5, 6, 7)
            cap = 1 if kind == "lds" else knob("VALU_CAP", 2)
            for g in gaps:
                if used.get((cur, g), 0) < cap:
                    used[(cur, g)] = used.get((cur, g), 0) + 1
                    return cur, g
            cur = next(it)

    for kind, ins in items:
        if kind == "next":
            cur = next(it)
            continue
        sl, g = take("lds" if kind == "lds" else "valu")
        plan.setdefault(sl, []).append((g, ins))


# ------------------------------------------------------------------------------------------------------------------------
# one slot
# ------------------------------------------------------------------------------------------------------------------------
def emit_slot(p, slots, n, fillers, nchunks):
    s = slots[n]
    NS = len(slots)
    par = n & 1
    pos = n % CH
    chunk = n // CH
    A = P[par]
    An = P[par ^ 1]
    nxt = slots[(n + 1) % NS]
    gap = [[] for _ in range(8)]
    pre = []

    # ---- operands of THIS slot are ready: LDS reads of the previous slot, X registers
    if not knob("NOLGKM"):
        pre.append("s_waitcnt lgkmcnt(0)")
    if s.kind == "x" and s.layer == 0 and s.k % 4 == 0 and not knob("XNOWAIT"):
        pre.append(lambda pp: pp.wait_vm("x0" if s.k == 0 else f"x{s.k // 4}"))
    if s.kind == "vx" and s.k == 0:
        pre.append(lambda pp: pp.wait_vm(f"d{QD - 1}"))

    # ---- MFMAs
    mf = []
    if s.kind in ("x", "h", "b"):
        dst = layer_dst(s.layer)
        first = (s.kind == "x" and s.k == 0) or (s.kind == "h" and s.k == 0 and s.layer != 4)
        if s.kind == "x":
            bop = f"v{XV[(s.k // 4) & 1] + (s.k & 3)}"
        elif s.kind == "h":
            bop = f"v{vb_reg(s.k)}"
        else:
            bop = f"v{V_ONE}"
        for b in range(8):
            mf.append(mfma(dst(b), A + b, bop, "0" if first else dst(b)))
    elif s.kind == "vx":
        xq = DG[s.k // 2] + 2 * (s.k & 1)
        for half in range(2):
            for b in range(4):
                mf.append(mfma(HD(b), A + 4 * half + b, f"v{xq + half}", "0" if (s.k == 0 and half == 0) else HD(b)))
    elif s.kind == "vh":
        for half in range(2):
            for b in range(4):
                mf.append(mfma(HD(b), A + 4 * half + b, f"v{2 * s.k + half}", HD(b)))      # B = accA raw (xyz_encoding_final has no activation)
    elif s.kind == "vb":
        for b in range(4):
            mf.append(mfma(HD(b), A + b, f"v{V_ONE}", HD(b)))
    nm = len(mf)          # (a "pad" slot has none)

    # ---- the NEXT slot's A operands (issued early: complete at the next slot's lgkmcnt(0))
    boundary = pos == CH - 1
    if PIPEPUB:
        o = (((chunk + 1) & 1) * CHUNK_B) if boundary else ((chunk & 1) * CHUNK_B + (pos + 1) * SLOT_B)
        rd = [f"ds_read_b128 v[{An}:{An + 3}], v{V_LANE16} offset:{o}", f"ds_read_b128 v[{An + 4}:{An + 7}], v{V_LANE16} offset:{o + 1024}"]
    elif boundary:
        rd = [f"ds_read_b128 v[{An}:{An + 3}], v{V_T1}", f"ds_read_b128 v[{An + 4}:{An + 7}], v{V_T1} offset:1024"]
    else:
        o = (pos + 1) * SLOT_B
        rd = [f"ds_read_b128 v[{An}:{An + 3}], v{V_T0} offset:{o}", f"ds_read_b128 v[{An + 4}:{An + 7}], v{V_T0} offset:{o + 1024}"]

    # ---- ring: refill in the first four slots of a chunk, rendezvous + publish on the last
    if pos < 4:
        def fetch(pp, q=pos):
            return p_vmem(pp, f"global_load_dwordx4 v[{ST + 4 * q}:{ST + 4 * q + 3}], v{V_WOFF}, {sp(S_WCUR)} offset:{1024 * q}", f"w{q}")
        if PIPEPUB and not knob("NOPUB"):
            # this wave's piece `pos` of the NEXT chunk: landed (requested one chunk ago), into the other half of the ring
            gap[knob("PUB_GAP", 2)].append(lambda pp, q=pos: pp.wait_vm(f"w{q}"))
            gap[knob("PUB_GAP", 2)].append(f"ds_write_b128 v{V_WOFF}, v[{ST + 4 * pos}:{ST + 4 * pos + 3}] "
                                           f"offset:{((chunk + 1) & 1) * CHUNK_B + 1024 * pos}")
        if not knob("NOFETCH"):
            gap[3].append(fetch)
        if pos == 3:
            if chunk == nchunks - 3:        # chunk + 2 was the stream's last: the next refill starts over (the next tile's chunk 0)
                gap[3].append(f"s_mov_b64 {sp(S_WCUR)}, {sp(S_WBASE)}")
            else:
                gap[3] += [f"s_add_u32 s{S_WCUR}, s{S_WCUR}, {CHUNK_B}", f"s_addc_u32 s{S_WCUR + 1}, s{S_WCUR + 1}, 0"]
    if knob("NOREADS"):
        rd = []
    if boundary and PIPEPUB:
        bg = knob("BAR_GAP", 0)
        if not knob("NOBAR"):
            gap[bg].append("s_barrier")           # chunk + 1 is complete and visible; everybody is done with this chunk's half
        gap[bg] += rd
    elif boundary:
        bg = knob("BAR_GAP", 0)
        gap[bg].append(lambda pp: pp.wait_vm("w3"))
        if not knob("NOBAR"):
            gap[bg].append("s_barrier")
        gap[bg] += rd
        gap[bg + 1].append(f"v_add_u32 v{V_PUB}, v{V_T2}, v{V_WAVE4K}")
        if not knob("NOPUB"):
            for q in range(4):
                gap[bg + 1 + q].append(f"ds_write_b128 v{V_PUB}, v[{ST + 4 * q}:{ST + 4 * q + 3}] offset:{1024 * q}")
        gap[6] += [f"v_mov_b32 v{V_TMP}, v{V_T0}", f"v_mov_b32 v{V_T0}, v{V_T1}", f"v_mov_b32 v{V_T1}, v{V_T2}",
                   f"v_mov_b32 v{V_T2}, v{V_TMP}"]
    elif rd:
        gap[knob("RD0_GAP", 0)].append(rd[0])
        gap[knob("RD1_GAP", 1)].append(rd[1])

    # ---- B operands of the hidden K-steps: relu(src register), RELU_BATCH K-steps per VALU burst (a burst costs the matrix pipe
    # about the same whether it holds one instruction or eight: measured), double-buffered groups
    RB = knob("RELU_BATCH", 4)
    grp = None
    if nxt.kind == "h" and nxt.k == 0:
        grp, g = (nxt.layer, 0), (4 if s.kind == "b" else 2)      # behind a bias slot: block 0 got its last term in this slot's first MFMA
    elif s.kind == "h" and s.k % RB == 0 and s.k + RB < 128:
        grp, g = (s.layer, s.k // RB + 1), knob("RELU_GAP", 3)
    if grp is not None and not knob("NORELU"):
        layer, gi = grp
        for k in range(gi * RB, (gi + 1) * RB):
            tgt = vb_reg(k)
            if layer % 2 == 1:                  # src = accA (VGPRs)
                gap[g].append(f"v_max_f32 v{tgt}, 0, v{k}")
            else:                               # src = accB (AGPRs)
                gap[g].append(f"v_accvgpr_read_b32 v{tgt}, a{k}")
        if layer % 2 == 0:
            for k in range(gi * RB, (gi + 1) * RB):
                gap[g].append(f"v_max_f32 v{vb_reg(k)}, 0, v{vb_reg(k)}")

    # ---- X traffic
    if s.kind == "x" and s.layer == 0:
        q, e = divmod(s.k, 4)
        if e == 0 and q + 1 < QX:
            def xl(pp, q=q):
                return p_vmem(pp, f"global_load_dwordx4 v[{XV[(q + 1) & 1]}:{XV[(q + 1) & 1] + 3}], v{V_LANE16}, {sp(S_X)} nt", f"x{q + 1}")
            gap[2] += [f"s_add_u32 s{S_X}, s{S_XTILE}, {1024 * (q + 1)}", f"s_addc_u32 s{S_X + 1}, s{S_XTILE + 1}, 0"]
            gap[3 if pos >= 4 else 5].append(xl)
        if e == 1:
            gap[2].append(f"ds_write_b128 v{V_STASH}, v[{XV[q & 1]}:{XV[q & 1] + 3}] offset:{1024 * q}")
    if s.kind == "x" and s.layer == 4:
        q, e = divmod(s.k, 4)
        if e == 0 and q + 1 < QX:
            gap[2].append(f"ds_read_b128 v[{XV[(q + 1) & 1]}:{XV[(q + 1) & 1] + 3}], v{V_STASH} offset:{1024 * (q + 1)}")
    if nxt.kind == "x" and nxt.layer == 4 and nxt.k == 0:
        gap[2].append(f"ds_read_b128 v[{XV[0]}:{XV[0] + 3}], v{V_STASH} offset:0")
    # view-branch X: the 7 direction groups behind the 25 position groups, ALL requested during layer 8's hidden part (one per
    # four slots) into DG — the two X quads + the rgb head's weight registers, idle between layer 0 and the next tile
    if s.kind == "h" and s.layer == 8 and 64 <= s.k < 64 + 4 * QD and s.k % 4 == 0:
        g_ = (s.k - 64) // 4

        def dl(pp, g_=g_):
            return p_vmem(pp, f"global_load_dwordx4 v[{DG[g_]}:{DG[g_] + 3}], v{V_LANE16}, {sp(S_X)} nt", f"d{g_}")
        gap[1] += [f"s_add_u32 s{S_X}, s{S_XTILE}, {1024 * (QX + g_)}", f"s_addc_u32 s{S_X + 1}, s{S_XTILE + 1}, 0"]
        gap[3 if pos >= 4 else 5].append(dl)
    # the next tile's first X group: requested in the middle of the view branch (the X registers are free from here on)
    if s.kind == "vh" and s.k == 8:
        gap[4] += [f"s_add_u32 s{S_TN}, s{S_TG}, s{S_NBLK}",
                   f"s_cmp_lt_i32 s{S_TN}, s{S_NGROUPS}",
                   f"s_cselect_b32 s{S_TN}, s{S_TN}, s{S_TG}"]       # past the end: this tile again (valid memory, never used)
    if s.kind == "vh" and s.k == 9:
        sel = Prog()
        emit_tile_select(sel, S_TN, S_XNEXT)
        gap[4] += sel.lines
    if s.kind == "vh" and s.k == 10:
        def nx(pp):
            return p_vmem(pp, f"global_load_dwordx4 v[{XV[0]}:{XV[0] + 3}], v{V_LANE16}, {sp(S_XNEXT)} nt", "x0")
        gap[5 if pos < 4 else 3].append(nx)

    # ---- fillers (heads)
    for g, ins in fillers:
        gap[min(g, nm - 1)] += ins

    if knob("BARE"):          # ceiling probe: the MFMAs alone (garbage results)
        pre, gap = [], [[] for _ in range(8)]
    # ---- emit
    for x in pre:
        t = x(p) if callable(x) else x
        if t:
            p.i(t)
    for m in range(nm):
        if knob("ALIGN") == 1 or (knob("ALIGN") == 2 and m == 0):
            p.i(".p2align 3")
        p.i(mf[m])
        gi = m if nm == 8 else m            # (4-MFMA slot: gaps 0..3; the rest of its fillers follow the last MFMA)
        for x in gap[gi]:
            t = x(p) if callable(x) else x
            if t:
                p.i(t)
    for gi in range(nm, 8):
        for x in gap[gi]:
            t = x(p) if callable(x) else x
            if t:
                p.i(t)


def main():
    lines = gen()
    out = sys.argv[1] if len(sys.argv) > 1 else None
    text = "".join('"%s\\n"\n' % ln for ln in lines)
    if out:
        with open(out, "w") as f:
            f.write("// generated by gen_mlp_a.py — do not edit\n")
            f.write("#define NF_A_LDS_BYTES_%d_%d %d\n" % (QX, QD, LDS_BYTES))
            f.write("#define NF_A_SLOTS_%d_%d %d\n" % (QX, QD, len(build_tile())))
            f.write(text)
    else:
        sys.stdout.write(text)
    sys.stderr.write("[gen_mlp_a] %d instructions, %d MFMAs, LDS %d bytes\n" %
                     (len(lines), sum("v_mfma" in ln for ln in lines), LDS_BYTES))


if __name__ == "__main__":
    main()
