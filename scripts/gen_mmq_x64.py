#!/usr/bin/env python3
"""Generator of the hand-scheduled K loop of the 64 x 64 wave-tile GEMM (csrc/hip/mmq_x64.hip): gfx950 assembly with PHYSICAL
registers, one inline-asm statement per weight format, written to csrc/hip/mmq_x64_loops.inc.

Why generated assembly: the loop is bound by the SIMD's own vector + matrix issue — two exact FMAs per (row, token, 32-group) triple
beside one int8 MFMA per 32 x 32 x 32 tile (reference role: vec_dot_*_q8_1_mma, HK/ggml/mmq.cuh:913-1737) — and hipcc neither keeps two
MFMA result sets in flight nor leaves prefetches where they are put.  Here every instruction is placed:
    MFMA(tile n)  ->  plain vector fillers in the MFMA's issue shadow (unpack, scale decode, addresses, loads)  ->  the FMAs of tile n-1
scripts/ubench_tile.hip measured that stream at 50-52 ns per tile per SIMD (two waves per SIMD) against 100+ in the round-3 kernel.

Wave tile: 64 weight rows x 64 tokens (2 x 2 MFMA tiles), lane = weight row, accumulator register = token:
    first stage   t   = fma(12582912 + C, dw, -12582912 dw)   = RN(C * dw) exactly (dw = d * sc has <= 19 significant bits)
    second stage  acc = fma(t, d8[token], acc)                  d8 as fp32 from the scratch, two tokens per v_pk_fma_f32
Weights: raw super-block bytes, row-major, by LDS-DMA into a wave-private two-stage ring (one stage = one 256-element K step of 64 rows);
activations + their scales straight from the x64 scratch layout (quantize.hip LAYOUT 5) into registers, one 32-group ahead.

usage: python scripts/gen_mmq_x64.py [--list]     (rewrites ggml-libtorch_amd/csrc/hip/mmq_x64_loops.inc)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "ggml-libtorch_amd", "csrc", "hip", "mmq_x64_loops.inc")

# ------------------------------------------------------------------------------------------------ register map (Q4_K)
ACC = 0            # v[0:63]    accumulators, tile ti = 2 tt + rt at 16 ti
CSET = (64, 80)    # two MFMA result sets
MAGICV = 96        # v[96:111]  0x4B400000 (input)
RAW = (112, 116)   # raw quant bytes of the current pair, per row tile
WOP = (120, 124)   # int8 MFMA operand of the current group, per row tile
ACT = ((128, 132), (136, 140))          # [group parity][tt]
D8 = (144, 160)                         # [tt], 16 registers each: the fp32 token scales of the current group, accumulator-register order
DWNM = (176, 184)  # per row tile: (dw, dw) (nm, nm) of the even group, (dw, dw) (nm, nm) of the odd group — duplicated pairs: see fma_block
HDR = (192, 200)   # per row tile: sc_lo, sc_hi, m_lo, m_hi, d, d, dmin, -
BMIN = (208, 212)  # per row tile: min-term operand (hi / lo split of -dmin * m), 4 registers
S8 = (216, 220)    # per token tile: s8 operand, 4 registers
V_LANE16, V_LDSD, V_LDSW0, V_HOFF, V_DMAOFF = 224, 225, 226, 227, 228      # inputs (V_LDSD: LDS address of this lane half's token scales)
V_LDSW, V_LDSWN, V_LDSHN = 229, 230, 231
T_DW, T_WHI, T_HW, T_HD = 232, 236, 240, 248                                 # temporaries by task (T_HW: header words, 8; also bmin temps)
N_VGPR = 250

# SGPRs (physical; the C++ side binds its values to them)
S_WRSRC, S_ARSRC = 36, 40          # s[36:39] weight tile descriptor, s[40:43] activation scratch descriptor
S_LDS, S_NSB, S_SBSTRIDE, S_WK, S_RB7 = 44, 45, 46, 47, 48     # S_RB7 = 7 * row_bytes: the rows one DMA instruction covers
S_F0, S_F1, S_D0, S_D1, S_S8 = 49, 50, 51, 52, 53               # byte offsets into the scratch: running fragment offsets, d8 tables and s8 operand of the current super-block
S_T0, S_T1, S_STAGE, S_NSTAGE, S_INC6, S_WKN = 54, 55, 56, 57, 58, 59
S_NEGM, S_1024, S_MASK0F, S_HI = 60, 62, 63, 66                  # s[60:61] = -12582912.0f x 2; s[66:67] = lanes 32-63
S_NEXT = 74                                                      # byte distance to the next super-block's records (0 on the last one of the slice)
S_BIG, S_256, S_EXEC = 68, 70, 72                                # s[68:69] cold-path mask, s[70:71] = 256.0f x 2, s[72:73] saved exec
S_WSTEP = 83                                                     # input: weight bytes from one K step of this wave to its next (K-slices are interleaved: slices x step bytes)
REC = 10240        # bytes of one (super-block, 32-token tile) record of the scratch
ROWS = 64
# experiment switches (scripts only; the shipped .inc is generated with the defaults)
X_INPLACE = int(os.environ.get("X64_INPLACE", "1"))    # 0: the first stage writes a separate register set instead of the MFMA's
X_NOPS = int(os.environ.get("X64_NOPS", "0"))          # s_nop wait states after every int8 MFMA
TSET = 240
X_NOMIN = int(os.environ.get("X64_NOMIN", "0"))        # 1: no min-term MFMA (wrong results; determinism experiments only)
X_PLAIN2 = int(os.environ.get("X64_PLAIN2", "0"))      # 1: second stage as plain v_fma_f32 on the raw d8 dwords (wrong results; determinism experiments only)
X_NOD8 = int(os.environ.get("X64_NOD8", "0"))          # 1: no token-scale loads inside the loop (wrong results; timing experiments only)
X_NOACT = int(os.environ.get("X64_NOACT", "0"))        # 1: no activation-fragment loads inside the loop (wrong results; timing experiments only)
X_NOFILL = int(os.environ.get("X64_NOFILL", "0"))      # timing experiments only (wrong results): 1 no dw_prep, 2 no unpack + raw reads, 8 no header decode inside the loop
X_NOWAIT = int(os.environ.get("X64_NOWAIT", "0"))      # timing experiments only (wrong results): 1 = no vmcnt waits inside the loop, 2 = no lgkmcnt waits, 3 = neither
X_DMAEARLY = int(os.environ.get("X64_DMAEARLY", "0"))  # 1: the next stage's ten DMA instructions in groups 0 .. 2 (one per slot) instead of 0 .. 4
X_NODMA = int(os.environ.get("X64_NODMA", "0"))        # 1: no LDS-DMA inside the loop, every super-block re-reads stage 0 (wrong results; determinism experiments only)


def vr(b, n=1):
    if n > 1:
        assert b % 2 == 0, b   # gfx950: 64-bit and wider VGPR tuples are even-aligned
    return f"v{b}" if n == 1 else f"v[{b}:{b + n - 1}]"


def sr(b, n=1):
    if n > 1:
        assert b % 2 == 0, b
    return f"s{b}" if n == 1 else f"s[{b}:{b + n - 1}]"


class Asm:
    """instruction list + in-order counters for vector memory (vmcnt) and LDS (lgkmcnt) operations"""

    def __init__(self):
        self.lines, self.vm, self.lg = [], [], []

    def i(self, s):
        self.lines.append(s)

    def vmem(self, s, tag):
        self.lines.append(s)
        self.vm.append(tag)

    def lds(self, s, tag):
        self.lines.append(s)
        self.lg.append(tag)

    def _wait(self, q, tag, name, limit):
        if tag not in q:
            return q
        idx = len(q) - 1 - q[::-1].index(tag)   # youngest operation with that tag
        n = len(q) - 1 - idx
        assert n <= limit, (name, n)
        self.lines.append(f"s_waitcnt {name}({n})")
        return q[idx + 1:]

    armed = False   # X_NOWAIT experiments: set once the loop body starts

    def wait_vm(self, tag):
        n0 = len(self.lines)
        self.vm = self._wait(self.vm, tag, "vmcnt", 63)
        if Asm.armed and (X_NOWAIT & 1):
            del self.lines[n0:]

    def wait_lg(self, tag):
        n0 = len(self.lines)
        self.lg = self._wait(self.lg, tag, "lgkmcnt", 15)
        if Asm.armed and (X_NOWAIT & 2):
            del self.lines[n0:]


class Q4K:
    name, type_id = "q4k", 12
    BS = 144                 # bytes of a row in one stage (one super-block)
    CPR, DIV = 9, 7282       # 16-byte chunks per row of a stage; c / CPR = (c * DIV) >> 16
    N_DMA = 10               # LDS-DMA instructions per stage: 7 rows (63 chunks) each, the last one row 63 alone
    QS_OFF = 16              # first quant byte inside the super-block
    STAGE = ROWS * 144
    ROWS = ROWS
    RPI = 7                  # rows one DMA instruction covers (63 // CPR)
    Q5 = False


class Q5K_R1:
    """Q5_K, one-row-tile waves only (32 rows x 64 tokens): 176-byte super-blocks {d, dmin, scales[12]; qh[32]; qs[128]} at a 176-byte pitch
    (11 chunks: odd, conflict-free) — two stages of 64 rows would not fit eight waves' LDS, two of 32 rows do (13 KB per wave).  Q4_K's
    header, scales and min term as they are; the operand is the nibble | bit g of qh[l] << 4: the lane's 16 qh bytes are read once per
    super-block (the next one's beside the next header) and cost two more vector instructions per operand register."""
    name, type_id = "q5k", 13
    BS = 176
    CPR, RPI = 11, 5
    ROWS = 32
    N_DMA = 7                # 6 x 5 rows + 2
    QS_OFF = 48
    STAGE = 32 * 176
    Q5 = True


class Q4K_R1(Q4K):
    """one row tile per wave (32 rows x 64 tokens): the second wave kind of the 96-row units (mmq_x64.hip, R3)"""
    ROWS = 32
    N_DMA = 5                # 4 x 7 rows + 4
    STAGE = 32 * 144


class Q80:
    """Q8_0: 34-byte blocks {fp16 d; int8 qs[32]}.  One ring stage = 128 elements = 136 bytes of a row, copied at a 144-byte pitch (the
    ninth 16-byte chunk runs 8 bytes into the next stage: never read), so the row-major LDS-DMA scheme and the conflict-free pitch of
    Q4_K carry over.  The int8 MFMA operand is the raw quant bytes: lane (row, h) needs bytes 34 g + 2 + 16 h .. + 15 of its row's stage,
    which two ALIGNED ds_read_b128 cover; v_alignbyte_b32 / v_mov_b32 move them into place (a misaligned ds_read_b128 costs 3.4 x)."""
    name, type_id = "q80", 8
    BS = 136
    RPI = 7
    CPR, DIV = 9, 7282
    N_DMA = 10
    STAGE = ROWS * 144
    ROWS = ROWS


class Q80_R1(Q80):
    ROWS = 32
    N_DMA = 5
    STAGE = 32 * 144


class Q40:
    """Q4_0: 18-byte blocks {fp16 d; 16 bytes: element j = low nibble of byte j, element 16 + j = high nibble}.  One ring stage = 256 elements
    = 8 blocks = 144 bytes of a row: Q4_K's two-stage ring and DMA scheme as they are.  A lane's K half IS a nibble plane (h = 0: the low
    nibbles of the block's 16 bytes, h = 1: the high ones), so both lane halves read the same bytes — 18 g + 2 into the row's stage, through
    two aligned chunk reads + v_alignbyte_b32 as Q8_0 — and the int8 operand is ((b ^ x) << s) & 0xF0 = 16 (n - 8), with the per-lane
    constants x = 0x08 / 0x80 and s = 4 / 0; the row scale is d / 16."""
    name, type_id = "q40", 2
    BS = 144
    RPI = 7
    CPR, DIV = 9, 7282
    N_DMA = 10
    STAGE = ROWS * 144
    ROWS = ROWS


class Q40_R1(Q40):
    ROWS = 32
    N_DMA = 5
    STAGE = 32 * 144


V_SH, V_XC = 212, 213        # Q4_0: per-lane shift (4 for the low-nibble half, 0 for the high one) and xor constant (0x08 / 0x80 per byte)
S_MASKF0, S_SIXTEENTH, S_NEGM16 = 64, 78, 80   # Q4_0: 0xf0f0f0f0; s[78:79] = 0.0625 x 2; s[80:81] = -12582912 / 16 x 2


F = Q4K
R1 = False                  # one-row-tile schedule (gen_r1): the tiles of a group are ti = 0, 2 and alternate the result sets
RAW8 = (192, 200)           # Q8_0: two aligned 16-byte chunks of the row's stage per row tile (the Q4_K header / min-term registers are free)
DREG = ((208, 209), (210, 211))   # Q8_0: the block's fp16 d, [row tile][group parity]
S_RUNA, S_RUNB, S_NEXTW = 75, 76, 77   # Q8_0: running source offsets of the two stage copies, byte step to the next 256 elements


T1 = False                  # one-tile schedule (gen_t1): one MFMA tile per group, result sets alternate by group


def cset(g, ti):
    if T1:
        return CSET[g & 1]
    return CSET[(2 * g + (ti >> 1)) & 1] if R1 else CSET[(4 * g + ti) & 1]


def mfma(a, g, ti):
    rt, tt = ti & 1, ti >> 1
    c = cset(g, ti)
    if ti == 0:
        a.wait_vm(f"act0_{g & 1}")
    if ti == 2:
        a.wait_vm(f"act1_{g & 1}")
    a.i(f"v_mfma_i32_32x32x32_i8 {vr(c, 16)}, {vr(ACT[g & 1][tt], 4)}, {vr(WOP[rt], 4)}, {vr(MAGICV, 16)}")
    if X_NOPS:
        a.i(f"s_nop {X_NOPS - 1}")


def fma_block(a, g, ti):
    """the two exact FMAs per triple of tile (g, ti); its MFMA was issued one tile ago.  Both stages packed (v_pk_fma_f32): the second
    stage read the token scales as packed fp16 through v_fma_mix_f32 at first — half the scale registers — but with two waves per
    SIMD that instruction returned wrong low-half results in lanes 48-63 from time to time (profiles/r04_x64_nondeterminism.txt);
    with plain or packed fp32 FMAs the kernel is bit-reproducible."""
    rt, tt = ti & 1, ti >> 1
    c = cset(g, ti)
    p = g & 1
    if ti in (0, 2):
        a.wait_lg(f"d8_{tt}")
    dw, nm = DWNM[rt] + 4 * p, DWNM[rt] + 4 * p + 2
    tdst = c if X_INPLACE else TSET
    for j in range(0, 16, 2):
        a.i(f"v_pk_fma_f32 {vr(tdst + j, 2)}, {vr(c + j, 2)}, {vr(dw, 2)}, {vr(nm, 2)}")
    acc = ACC + 16 * ti
    d8 = D8[tt]
    for j in range(0, 16, 2):
        a.i(f"v_pk_fma_f32 {vr(acc + j, 2)}, {vr(tdst + j, 2)}, {vr(d8 + j, 2)}, {vr(acc + j, 2)}")


def dw_prep(a, q, rt):
    """row scales of groups 2q, 2q+1: dw = d * sc, nm = -12582912 * dw, each as a DUPLICATED register pair, so that the first FMA stage
    needs no op_sel broadcast (fma_block)"""
    sc = HDR[rt] + (q >> 1)
    b0 = 2 * (q & 1)
    for k in range(4):
        a.i(f"v_cvt_f32_ubyte{b0 + (k >> 1)} {vr(T_DW + k)}, {vr(sc)}")
    for par in range(2):
        a.i(f"v_pk_mul_f32 {vr(DWNM[rt] + 4 * par, 2)}, {vr(T_DW + 2 * par, 2)}, {vr(HDR[rt] + 4, 2)}")
        a.i(f"v_pk_mul_f32 {vr(DWNM[rt] + 4 * par + 2, 2)}, {vr(DWNM[rt] + 4 * par, 2)}, {sr(S_NEGM, 2)}")


QH, QHN = 200, 204           # Q5_K (one-row-tile loop: the second row tile's header registers are free): the lane's 16 qh bytes, this / next super-block
S_MASK10 = 65


def q5_hi(a, rt, g):
    """Q5_K: operand of group g = (bit g of the qh bytes << 4) | nibble (in T_WHI)"""
    a.wait_lg("qh")
    sh = 4 - (g & 7)
    src = QH
    if sh:
        for k in range(4):
            a.i(f"v_lsh{'l' if sh > 0 else 'r'}rev_b32 {vr(T_DW + k)}, {abs(sh)}, {vr(QH + k)}")
        src = T_DW
    for k in range(4):
        a.i(f"v_and_or_b32 {vr(WOP[rt] + k)}, {vr(src + k)}, {sr(S_MASK10)}, {vr(T_WHI + k)}")


def w_hi(a, rt, g=None):
    for k in range(4):
        a.i(f"v_lshrrev_b32 {vr(T_WHI + k)}, 4, {vr(RAW[rt] + k)}")
    if F.Q5:
        for k in range(4):
            a.i(f"v_and_b32 {vr(T_WHI + k)}, {sr(S_MASK0F)}, {vr(T_WHI + k)}")
        q5_hi(a, rt, g)
        return
    for k in range(4):
        a.i(f"v_and_b32 {vr(WOP[rt] + k)}, {sr(S_MASK0F)}, {vr(T_WHI + k)}")


def w_lo(a, rt, g=None):
    a.wait_lg(f"raw{rt}")
    if F.Q5:
        for k in range(4):
            a.i(f"v_and_b32 {vr(T_WHI + k)}, {sr(S_MASK0F)}, {vr(RAW[rt] + k)}")
        q5_hi(a, rt, g)
        return
    for k in range(4):
        a.i(f"v_and_b32 {vr(WOP[rt] + k)}, {sr(S_MASK0F)}, {vr(RAW[rt] + k)}")


def raw_read(a, rt, q):
    """raw nibble bytes of pair q (q = 4: pair 0 of the NEXT stage) of this lane's row: 16 bytes at QS_OFF + 32 q + 16 h"""
    if q < 4:
        a.lds(f"ds_read_b128 {vr(RAW[rt], 4)}, {vr(V_LDSW)} offset:{rt * 32 * F.BS + 32 * q}", f"raw{rt}")
    else:
        a.lds(f"ds_read_b128 {vr(RAW[rt], 4)}, {vr(V_LDSWN)} offset:{rt * 32 * F.BS}", f"raw{rt}")


X_NOACT_ARMED = [False]
def act_loads(a, par):
    """activation fragments of the next group (the running offsets point at it) into ACT[par]"""
    if X_NOACT and X_NOACT_ARMED[0]:
        return
    a.vmem(f"buffer_load_dwordx4 {vr(ACT[par][0], 4)}, {vr(V_LANE16)}, {sr(S_ARSRC, 4)}, {sr(S_F0)} offen", f"act0_{par}")
    a.vmem(f"buffer_load_dwordx4 {vr(ACT[par][1], 4)}, {vr(V_LANE16)}, {sr(S_ARSRC, 4)}, {sr(S_F1)} offen", f"act1_{par}")


def d8_reads(a, tt, target):
    """the 16 fp32 token scales of group `target` for this lane half (64 contiguous bytes of the wave's LDS table) into D8[tt]: four
    broadcast ds_read_b128, issued right after the last FMA block that reads the current ones (two to three tiles of lead).
    (As global loads these cost 8 vector-memory instructions per group: with them the kernel was bound by the CU's one texture
    addresser, 46 of its cycles per tile where a tile has 22 — profiles/r04_x64_ablations.txt.)"""
    if target == 4:
        a.wait_vm(f"d8dma1_{tt}")        # groups 4-7 of this super-block: requested in its first group
    if target == 0:
        a.wait_vm(f"d8dma0_{tt}")        # groups 0-3 of the next super-block: requested in group 3 / 4
    for k in range(4):
        a.lds(f"ds_read_b128 {vr(D8[tt] + 4 * k, 4)}, {vr(V_LDSD)} offset:{1024 * tt + 128 * target + 16 * k}", f"d8_{tt}" if k == 3 else f"d8_{tt}x")


def d8_dma(a, tt, half, nxt):
    """LDS-DMA of four groups' token scales (512 bytes, 32 lanes) of token tile tt: half 1 = groups 4-7 of THIS super-block (their
    table slots held the previous one's until its group-7 reads), half 0 with nxt = groups 0-3 of the NEXT one"""
    s = S_D0 if tt == 0 else S_D1
    a.i(f"s_add_u32 {sr(S_T0)}, {sr(s)}, {512 * half}")
    if nxt:
        a.i(f"s_add_u32 {sr(S_T0)}, {sr(S_T0)}, {sr(S_NEXT)}")
    a.i(f"s_add_u32 m0, {sr(S_LDS)}, {2 * F.STAGE + 1024 * tt + 512 * half}")
    a.i(f"s_mov_b64 {sr(S_EXEC, 2)}, exec")
    a.i("s_mov_b64 exec, 0xffffffff")
    a.vmem(f"buffer_load_dwordx4 {vr(V_LANE16)}, {sr(S_ARSRC, 4)}, {sr(S_T0)} offen lds", f"d8dma{half}_{tt}")
    a.i(f"s_mov_b64 exec, {sr(S_EXEC, 2)}")


def advance_offsets(a, g):
    """after the fragment loads for group g + 1 were issued: step their running offsets to group g + 2 (g = 6: to group 0 of the next super-block)"""
    for s in (S_F0, S_F1):
        a.i(f"s_add_u32 {sr(s)}, {sr(s)}, {sr(S_INC6) if g == 6 else '0x400'}")


def dma_instr(a, j, dst_stage):
    """LDS-DMA instruction j of a stage: rows 7 j .. 7 j + 6 of the row-major [64 rows][BS] image (lane = chunk 63 j + lane; lane 63
    repeats lane 0 of instruction j + 1, so both write the same bytes).  Per-lane source offset = one constant register (row lane / CPR of
    the seven, chunk lane % CPR), the rest is scalar: soffset = K position + 7 j rows.  The last instruction holds row 63 only."""
    if j == 0:
        a.i(f"s_mov_b32 {sr(S_T1)}, {sr(S_WKN)}")
    else:
        a.i(f"s_add_u32 {sr(S_T1)}, {sr(S_T1)}, {sr(S_RB7)}")
    a.i(f"s_add_u32 m0, {sr(dst_stage)}, {16 * F.RPI * F.CPR * j}")
    if j == F.N_DMA - 1:
        a.i(f"s_mov_b64 {sr(S_EXEC, 2)}, exec")
        mask = (1 << (F.CPR * (F.ROWS - F.RPI * j))) - 1
        if mask < (1 << 32):
            a.i(f"s_mov_b64 exec, {hex(mask)}")
        else:
            a.i("s_mov_b32 exec_lo, -1")
            a.i(f"s_mov_b32 exec_hi, {hex(mask >> 32)}")
    else:
        a.i("s_nop 0")                                                   # SALU write of M0 -> LDS-DMA: one wait state
    a.vmem(f"buffer_load_dwordx4 {vr(V_DMAOFF)}, {sr(S_WRSRC, 4)}, {sr(S_T1)} offen lds", "dma")
    if j == F.N_DMA - 1:
        a.i(f"s_mov_b64 exec, {sr(S_EXEC, 2)}")


def hdr_read(a, rt):
    a.lds(f"ds_read_b128 {vr(T_HW + 4 * rt, 4)}, {vr(V_LDSHN)} offset:{rt * 32 * F.BS}", f"hdr{rt}")


def hdr_decode(a, rt):
    """{d | dmin << 16, scales[0..3], scales[4..7], scales[8..11]} -> sc_lo, sc_hi, m_lo, m_hi, d, dmin (get_scale_min_k4 for all 8 groups)"""
    w = T_HW + 4 * rt
    h = HDR[rt]
    t0, t1 = T_HD, T_HD + 1
    a.wait_lg(f"hdr{rt}")
    a.i(f"v_and_b32 {vr(h)}, 0x3f3f3f3f, {vr(w + 1)}")                         # sc 0..3
    a.i(f"v_and_b32 {vr(h + 2)}, 0x3f3f3f3f, {vr(w + 2)}")                     # m 0..3
    a.i(f"v_lshrrev_b32 {vr(t0)}, 2, {vr(w + 1)}")
    a.i(f"v_and_b32 {vr(t0)}, 0x30303030, {vr(t0)}")
    a.i(f"v_and_or_b32 {vr(h + 1)}, {vr(w + 3)}, {sr(S_MASK0F)}, {vr(t0)}")    # sc 4..7
    a.i(f"v_lshrrev_b32 {vr(t0)}, 2, {vr(w + 2)}")
    a.i(f"v_and_b32 {vr(t0)}, 0x30303030, {vr(t0)}")
    a.i(f"v_lshrrev_b32 {vr(t1)}, 4, {vr(w + 3)}")
    a.i(f"v_and_or_b32 {vr(h + 3)}, {vr(t1)}, {sr(S_MASK0F)}, {vr(t0)}")       # m 4..7
    a.i(f"v_cvt_f32_f16_sdwa {vr(h + 6)}, {vr(w)} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")   # dmin
    a.i(f"v_cvt_f32_f16_e32 {vr(h + 4)}, {vr(w)}")                              # d, twice: a pair for the packed multiplies
    a.i(f"v_cvt_f32_f16_e32 {vr(h + 5)}, {vr(w)}")


def bmin_prep(a, rt, scaled=False):
    """-dmin * m_g = hi + lo exactly (two fp16 values, hi by round-toward-zero) for the four groups 4 h .. 4 h + 3 of this lane's row,
    in the K order hi(g0) hi(g1) lo(g0) lo(g1) hi(g2) hi(g3) lo(g2) lo(g3).  Rows with |dmin| > 1024 (the products leave the fp16
    range) contribute zero here and go through the 2^-8-scaled cold pass (`scaled`), in which the other rows contribute zero."""
    h = HDR[rt]
    d = BMIN[rt]
    t = T_HW          # 10 temporaries: the header-word registers are free outside groups 6 / 7
    a.i(f"v_cndmask_b32_e64 {vr(t + 8)}, {vr(h + 2)}, {vr(h + 3)}, {sr(S_HI, 2)}")   # m bytes of groups 4 h .. 4 h + 3
    a.i(f"v_cmp_nle_f32_e64 vcc, |{vr(h + 6)}|, {sr(S_1024)}")                      # |dmin| > 1024 (or NaN)
    if not scaled:
        a.i(f"v_cndmask_b32_e64 {vr(t + 9)}, {vr(h + 6)}, 0, vcc")
    else:
        a.i(f"v_mul_f32_e32 {vr(t + 9)}, 0x3b800000, {vr(h + 6)}")                  # dmin * 2^-8
        a.i(f"v_cndmask_b32_e64 {vr(t + 9)}, 0, {vr(t + 9)}, vcc")
    for j in range(4):
        a.i(f"v_cvt_f32_ubyte{j} {vr(t + j)}, {vr(t + 8)}")
    for j in range(4):
        a.i(f"v_mul_f32_e64 {vr(t + j)}, {vr(t + 9)}, -{vr(t + j)}")                # p_j = -(dm * m_j), exact
    for j in (0, 2):
        a.i(f"v_cvt_pkrtz_f16_f32 {vr(d + j)}, {vr(t + j)}, {vr(t + j + 1)}")        # hi pair
        a.i(f"v_cvt_f32_f16_e32 {vr(t + 4)}, {vr(d + j)}")
        a.i(f"v_cvt_f32_f16_sdwa {vr(t + 5)}, {vr(d + j)} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
        a.i(f"v_sub_f32_e32 {vr(t + 4)}, {vr(t + j)}, {vr(t + 4)}")
        a.i(f"v_sub_f32_e32 {vr(t + 5)}, {vr(t + j + 1)}, {vr(t + 5)}")
        a.i(f"v_cvt_pkrtz_f16_f32 {vr(d + j + 1)}, {vr(t + 4)}, {vr(t + 5)}")        # lo pair


def s8_loads(a):
    a.vmem(f"buffer_load_dwordx4 {vr(S8[0], 4)}, {vr(V_LANE16)}, {sr(S_ARSRC, 4)}, {sr(S_S8)} offen", "s8_0")
    a.i(f"s_add_u32 {sr(S_T0)}, {sr(S_S8)}, {REC}")
    a.vmem(f"buffer_load_dwordx4 {vr(S8[1], 4)}, {vr(V_LANE16)}, {sr(S_ARSRC, 4)}, {sr(S_T0)} offen", "s8_1")


def min_mfma(a, ti):
    if X_NOMIN:
        return
    rt, tt = ti & 1, ti >> 1
    acc = ACC + 16 * ti
    a.wait_vm(f"s8_{tt}")
    a.i("s_nop 1")
    a.i(f"v_mfma_f32_32x32x16_f16 {vr(acc, 16)}, {vr(S8[tt], 4)}, {vr(BMIN[rt], 4)}, {vr(acc, 16)}")


def gen(label):
    a = Asm()
    # ---------------------------------------------------------------- prologue
    for s in (S_NEGM, S_NEGM + 1):
        a.i(f"s_mov_b32 {sr(s)}, 0xcb400000")                     # -12582912.0f
    for s in (S_256, S_256 + 1):
        a.i(f"s_mov_b32 {sr(s)}, 0x43800000")                     # 256.0f
    a.i(f"s_mov_b32 {sr(S_MASK0F)}, 0x0f0f0f0f")
    a.i(f"s_mov_b32 {sr(S_1024)}, 0x44800000")                    # 1024.0f
    a.i(f"v_cmp_ne_u32_e64 {sr(S_HI, 2)}, 16, {vr(V_HOFF)}")      # lanes 32-63 (hoff = 16 + 16 h)
    a.i(f"s_mov_b32 {sr(S_STAGE)}, {sr(S_LDS)}")                  # LDS byte address of the current stage / of the next one
    a.i(f"s_add_u32 {sr(S_NSTAGE)}, {sr(S_LDS)}, {F.STAGE}")
    a.i(f"s_mov_b32 {sr(S_WKN)}, {sr(S_WK)}")                     # stage 0 <- the first super-block of the slice
    for j in range(F.N_DMA):
        dma_instr(a, j, S_STAGE)
    # s8 of the first super-block, activations + token scales of group 0 (same issue order as at the end of the loop body)
    s8_loads(a)
    a.i(f"s_mov_b32 {sr(S_NEXT)}, 0")
    d8_dma(a, 0, 0, False)                                        # groups 0-3 of the first super-block (4-7: in the loop's first group, as in every iteration)
    d8_dma(a, 1, 0, False)
    act_loads(a, 0)
    for s in (S_F0, S_F1):
        a.i(f"s_add_u32 {sr(s)}, {sr(s)}, 0x400")                 # the fragment offsets now point at group 1
    a.i(f"v_mov_b32 {vr(V_LDSW)}, {vr(V_LDSW0)}")
    a.i(f"v_add_u32 {vr(V_LDSWN)}, {0 if X_NODMA else F.STAGE}, {vr(V_LDSW0)}")
    a.i(f"v_sub_u32 {vr(V_LDSHN)}, {vr(V_LDSW0)}, {vr(V_HOFF)}")  # header address of stage 0 (for the prologue only)
    # result set 1 = magic and zero scales for the first "previous tile" FMA block: it adds exactly zero
    for k in range(16):
        a.i(f"v_mov_b32 {vr(CSET[1] + k)}, {vr(MAGICV + k)}")
    for k in range(16):
        a.i(f"v_mov_b32 {vr(D8[1] + k)}, 0")
    for k in range(8):
        a.i(f"v_mov_b32 {vr(DWNM[1] + k)}, 0")
    a.wait_vm("dma")
    a.wait_vm("d8dma0_1")
    hdr_read(a, 0)
    hdr_read(a, 1)
    a.lds(f"ds_read_b128 {vr(RAW[0], 4)}, {vr(V_LDSW)} offset:0", "raw0")
    a.lds(f"ds_read_b128 {vr(RAW[1], 4)}, {vr(V_LDSW)} offset:{32 * F.BS}", "raw1")
    hdr_decode(a, 0)
    hdr_decode(a, 1)
    w_lo(a, 0)
    w_lo(a, 1)
    d8_reads(a, 0, 0)                                             # (token tile 1's: after the first slot's FMA block, as in every iteration)
    a.i(f"v_sub_u32 {vr(V_LDSHN)}, {vr(V_LDSWN)}, {vr(V_HOFF)}")
    # ---------------------------------------------------------------- loop over the super-blocks of the slice
    X_NOACT_ARMED[0] = True
    Asm.armed = True
    a.i(f"L_sb_{label}%=:")
    # (the s8 loads are the oldest operations in flight: the first vmcnt wait of the body — for act0, younger — covers them, so
    #  they need no entry of their own; the loop body's own s8 loads, issued in group 3, are covered the same way by group 4's waits)
    if a.vm[:2] == ["s8_0", "s8_1"]:
        a.vm = a.vm[2:]
    vm0, lg0 = list(a.vm), list(a.lg)
    a.i(f"s_cmp_eq_u32 {sr(S_NSB)}, 1")                                               # last super-block of the slice?
    a.i(f"s_cselect_b32 {sr(S_T0)}, 0, {sr(S_SBSTRIDE)}")                              # then "next" = this one again (harmless re-read)
    a.i(f"s_cselect_b32 {sr(S_T1)}, 0, {sr(S_WSTEP)}")                                        # (both selects before anything rewrites SCC)
    a.i(f"s_sub_u32 {sr(S_INC6)}, {sr(S_T0)}, {7 * 1024}")
    a.i(f"s_add_u32 {sr(S_S8)}, {sr(S_S8)}, {sr(S_T0)}")
    a.i(f"s_add_u32 {sr(S_WKN)}, {sr(S_WK)}, {sr(S_T1)}")
    a.i(f"s_mov_b32 {sr(S_NEXT)}, {sr(S_T0)}")
    for g in range(8):
        q = g >> 1
        for ti in range(4):
            rt = ti & 1
            mfma(a, g, ti)
            # ---- fillers in the MFMA's issue shadow
            if g % 2 == 0 and ti < 2 and not (X_NOFILL & 1):
                dw_prep(a, q, ti)
            if ti == 0:
                act_loads(a, (g + 1) & 1)
                advance_offsets(a, g)
            if g == 6 and ti == 2:
                a.wait_vm("dma")                 # the next stage has landed: its header and first pair are read below
            if ti >= 2 and not (X_NOFILL & 2):
                if g % 2 == 0:
                    w_hi(a, rt)
                    raw_read(a, rt, q + 1)
                else:
                    w_lo(a, rt)
            if X_DMAEARLY:
                if 4 * g + ti < F.N_DMA and not X_NODMA:
                    dma_instr(a, 4 * g + ti, S_NSTAGE)   # weight DMA of the next stage: ten instructions over groups 0 .. 2
            elif g <= 4 and ti < 2 and 2 * g + ti < F.N_DMA and not X_NODMA:
                dma_instr(a, 2 * g + ti, S_NSTAGE)   # weight DMA of the next stage: ten instructions over groups 0 .. 4
            if g == 0 and ti >= 2:
                d8_dma(a, ti - 2, 1, False)      # token scales of groups 4-7 of this super-block
            if (g, ti) == (3, 2):
                d8_dma(a, 0, 0, True)            # ... of groups 0-3 of the next one (this one's were consumed by FMA(3, 0) / FMA(3, 2))
            if (g, ti) == (4, 0):
                d8_dma(a, 1, 0, True)
            if g == 1 and ti < 2:
                bmin_prep(a, ti)                 # min-term operand of this super-block
            if g == 6 and ti == 3 and not (X_NOFILL & 8):
                hdr_read(a, 0)                   # header of the next super-block (sc / m of this one are dead by now)
                hdr_read(a, 1)
            if g == 7 and ti == 1 and not (X_NOFILL & 8):
                hdr_decode(a, 0)
            if g == 7 and ti == 2 and not (X_NOFILL & 8):
                hdr_decode(a, 1)
            # ---- FMAs of the previous tile (first iteration, tile (7, 3): adds zero)
            pg, pti = (g, ti - 1) if ti else ((g - 1) % 8, 3)
            fma_block(a, pg, pti)
            if pti == 1:
                d8_reads(a, 0, (pg + 1) % 8)     # token tile 0's scales of the next group (tiles 0 and 1 have read the current ones)
            if pti == 3:
                d8_reads(a, 1, (pg + 1) % 8)
            if pg == 2:
                min_mfma(a, pti)                 # right after the tile's group-2 FMA block, for every tile (the same summation order in all four:
                                                 # a row's result must not depend on which half of the unit it sits in), a group before its next one
            if g == 3 and ti == 0:
                # rows with |dmin| > 1024: cold pass
                a.i(f"v_cmp_nle_f32_e64 {sr(S_BIG, 2)}, |{vr(HDR[0] + 6)}|, {sr(S_1024)}")
                a.i(f"v_cmp_nle_f32_e64 vcc, |{vr(HDR[1] + 6)}|, {sr(S_1024)}")
                a.i(f"s_or_b64 {sr(S_BIG, 2)}, {sr(S_BIG, 2)}, vcc")
                a.i(f"s_cmp_lg_u64 {sr(S_BIG, 2)}, 0")
                a.i(f"s_cbranch_scc1 L_cold_{label}%=")
                a.i(f"L_warm_{label}%=:")
                s8_loads(a)                      # s8 of the next super-block (the S8 registers are free now)
    # stage swap + loop control
    a.i(f"s_add_u32 {sr(S_WK)}, {sr(S_WK)}, {sr(S_WSTEP)}")
    a.i(f"s_add_u32 {sr(S_D0)}, {sr(S_D0)}, {sr(S_NEXT)}")
    a.i(f"s_add_u32 {sr(S_D1)}, {sr(S_D1)}, {sr(S_NEXT)}")
    a.i(f"s_mov_b32 {sr(S_T0)}, {sr(S_STAGE)}")
    a.i(f"s_mov_b32 {sr(S_STAGE)}, {sr(S_NSTAGE)}")
    a.i(f"s_mov_b32 {sr(S_NSTAGE)}, {sr(S_T0)}")
    if not X_NODMA:
        a.i(f"v_mov_b32 {vr(T_DW)}, {vr(V_LDSW)}")
        a.i(f"v_mov_b32 {vr(V_LDSW)}, {vr(V_LDSWN)}")
        a.i(f"v_mov_b32 {vr(V_LDSWN)}, {vr(T_DW)}")
    a.i(f"v_sub_u32 {vr(V_LDSHN)}, {vr(V_LDSWN)}, {vr(V_HOFF)}")
    a.i(f"s_sub_u32 {sr(S_NSB)}, {sr(S_NSB)}, 1")
    a.i(f"s_cmp_lg_u32 {sr(S_NSB)}, 0")
    a.i(f"s_cbranch_scc1 L_sb_{label}%=")
    # the operations in flight at the end of the body must be the YOUNGEST ones of those in flight at its top, in the same order: every
    # wait count of the body (= operations younger than the awaited one) then holds on the second and later iterations too (what the
    # top-of-loop queue holds beyond that suffix is known complete by then: its waits are satisfied at once)
    assert a.vm == vm0[len(vm0) - len(a.vm):] and a.lg == lg0, (a.vm, vm0, a.lg, lg0)
    # ---------------------------------------------------------------- epilogue: the FMAs of the last tile, drain
    Asm.armed = False
    X_NOACT_ARMED[0] = False
    fma_block(a, 7, 3)
    a.i("s_waitcnt vmcnt(0) lgkmcnt(0)")
    a.i("s_nop 7")
    a.i("s_nop 7")                               # the min-term MFMAs have long retired; nothing is in flight when the asm ends
    a.i(f"s_branch L_end_{label}%=")
    # ---------------------------------------------------------------- cold pass: rows with |dmin| > 1024 at 2^-8 of their scale, x 256 afterwards
    a.i(f"L_cold_{label}%=:")
    bmin_prep(a, 0, scaled=True)
    bmin_prep(a, 1, scaled=True)
    z = CSET[1]                                  # free here: tile (2, 3)'s FMAs are done, MFMA(3, 1) comes after
    for ti in range(4):
        rt, tt = ti & 1, ti >> 1
        acc = ACC + 16 * ti
        a.i(f"v_mfma_f32_32x32x16_f16 {vr(z, 16)}, {vr(S8[tt], 4)}, {vr(BMIN[rt], 4)}, 0")
        a.i("s_nop 7")
        a.i("s_nop 7")
        for j in range(0, 16, 2):
            a.i(f"v_pk_fma_f32 {vr(acc + j, 2)}, {vr(z + j, 2)}, {sr(S_256, 2)}, {vr(acc + j, 2)}")
        a.i("s_nop 3")
    a.i(f"s_branch L_warm_{label}%=")
    a.i(f"L_end_{label}%=:")
    return a



def gen_r1(label, fmt=None):
    """Q4_K, ONE row tile per wave (32 rows x 64 tokens, tiles ti = 0, 2 = token tiles 0, 1; accumulators v[0:15], v[32:47]): the
    second wave kind of the 96-row units.  Same pipeline as gen() with two slots per group: slot (g, 0) carries the loads (activations,
    weight DMA), slot (g, 2) the unpack / scale work of the one row tile (its operand is free once MFMA(g, 2) has issued)."""
    global F, R1
    fmt = fmt or Q4K_R1
    F, R1 = fmt, True
    a = Asm()
    for s in (S_NEGM, S_NEGM + 1):
        a.i(f"s_mov_b32 {sr(s)}, 0xcb400000")
    for s in (S_256, S_256 + 1):
        a.i(f"s_mov_b32 {sr(s)}, 0x43800000")
    a.i(f"s_mov_b32 {sr(S_MASK0F)}, 0x0f0f0f0f")
    a.i(f"s_mov_b32 {sr(S_MASK10)}, 0x10101010")
    a.i(f"s_mov_b32 {sr(S_1024)}, 0x44800000")
    a.i(f"v_cmp_ne_u32_e64 {sr(S_HI, 2)}, {F.QS_OFF}, {vr(V_HOFF)}")      # lanes 32-63 (hoff = QS_OFF + 16 h)
    a.i(f"s_mov_b32 {sr(S_STAGE)}, {sr(S_LDS)}")
    a.i(f"s_add_u32 {sr(S_NSTAGE)}, {sr(S_LDS)}, {F.STAGE}")
    a.i(f"s_mov_b32 {sr(S_WKN)}, {sr(S_WK)}")
    for j in range(F.N_DMA):
        dma_instr(a, j, S_STAGE)
    s8_loads(a)
    a.i(f"s_mov_b32 {sr(S_NEXT)}, 0")
    d8_dma(a, 0, 0, False)
    d8_dma(a, 1, 0, False)
    act_loads(a, 0)
    for s in (S_F0, S_F1):
        a.i(f"s_add_u32 {sr(s)}, {sr(s)}, 0x400")
    d8_dma(a, 0, 1, False)                                        # groups 4-7 of the first super-block (later ones: in group 7 of the one before)
    d8_dma(a, 1, 1, False)
    a.i(f"v_mov_b32 {vr(V_LDSW)}, {vr(V_LDSW0)}")
    a.i(f"v_add_u32 {vr(V_LDSWN)}, {F.STAGE}, {vr(V_LDSW0)}")
    a.i(f"v_sub_u32 {vr(V_LDSHN)}, {vr(V_LDSW0)}, {vr(V_HOFF)}")
    # the first "previous tile" FMA block, (7, 2): result set 1 = magic, zero token scales, zero row scales of the odd group -> adds zero
    for k in range(16):
        a.i(f"v_mov_b32 {vr(CSET[1] + k)}, {vr(MAGICV + k)}")
    for k in range(16):
        a.i(f"v_mov_b32 {vr(D8[1] + k)}, 0")
    for k in range(4, 8):
        a.i(f"v_mov_b32 {vr(DWNM[0] + k)}, 0")
    a.wait_vm("dma")
    a.wait_vm("d8dma0_1")
    hdr_read(a, 0)
    if F.Q5:
        a.i(f"v_subrev_u32 {vr(T_HD)}, 32, {vr(V_LDSW)}")         # the lane's qh bytes: QS_OFF - 32 + 16 h into the row
        a.lds(f"ds_read_b128 {vr(QH, 4)}, {vr(T_HD)}", "qh")
    a.lds(f"ds_read_b128 {vr(RAW[0], 4)}, {vr(V_LDSW)} offset:0", "raw0")
    hdr_decode(a, 0)
    w_lo(a, 0, 0)
    d8_reads(a, 0, 0)
    a.i(f"v_sub_u32 {vr(V_LDSHN)}, {vr(V_LDSWN)}, {vr(V_HOFF)}")
    Asm.armed = True
    a.i(f"L_sb_{label}%=:")
    if a.vm[:2] == ["s8_0", "s8_1"]:
        a.vm = a.vm[2:]
    vm0, lg0 = list(a.vm), list(a.lg)
    a.i(f"s_cmp_eq_u32 {sr(S_NSB)}, 1")
    a.i(f"s_cselect_b32 {sr(S_T0)}, 0, {sr(S_SBSTRIDE)}")
    a.i(f"s_cselect_b32 {sr(S_T1)}, 0, {sr(S_WSTEP)}")
    a.i(f"s_sub_u32 {sr(S_INC6)}, {sr(S_T0)}, {7 * 1024}")
    a.i(f"s_add_u32 {sr(S_S8)}, {sr(S_S8)}, {sr(S_T0)}")
    a.i(f"s_add_u32 {sr(S_WKN)}, {sr(S_WK)}, {sr(S_T1)}")
    a.i(f"s_mov_b32 {sr(S_NEXT)}, {sr(S_T0)}")
    for g in range(8):
        q = g >> 1
        # ------------------------------------------------ slot (g, 0)
        mfma(a, g, 0)
        act_loads(a, (g + 1) & 1)
        advance_offsets(a, g)
        for j in (2 * g, 2 * g + 1):
            if j < F.N_DMA:
                dma_instr(a, j, S_NSTAGE)        # the next stage: five instructions over groups 0 .. 2
        if g == 1:
            bmin_prep(a, 0)
        if g == 3:
            d8_dma(a, 0, 0, True)                # token tile 0's scales of groups 0-3 of the next super-block (this one's group 3 was read in slot (2, 2))
        pg = (g - 1) % 8
        fma_block(a, pg, 2)
        d8_reads(a, 1, (pg + 1) % 8)
        if pg == 2:
            min_mfma(a, 2)
        if g == 3:
            a.i(f"v_cmp_nle_f32_e64 {sr(S_BIG, 2)}, |{vr(HDR[0] + 6)}|, {sr(S_1024)}")
            a.i(f"s_cmp_lg_u64 {sr(S_BIG, 2)}, 0")
            a.i(f"s_cbranch_scc1 L_cold_{label}%=")
            a.i(f"L_warm_{label}%=:")
            s8_loads(a)
        # ------------------------------------------------ slot (g, 2)
        mfma(a, g, 2)
        if g == 6:
            a.wait_vm("dma")
        if g % 2 == 0:
            w_hi(a, 0, g + 1)
            raw_read(a, 0, q + 1)
            dw_prep(a, q, 0)                     # row scales of groups g, g + 1 (the odd group's old ones were last read in slot (g, 0))
        else:
            if g == 7 and F.Q5:                  # the next super-block's qh bytes (read in slot (6, 2); group 7's operand is built)
                a.wait_lg("qh")
                for k in range(4):
                    a.i(f"v_mov_b32 {vr(QH + k)}, {vr(QHN + k)}")
            w_lo(a, 0, g + 1)
        if g == 3:
            d8_dma(a, 1, 0, True)
        if g == 6:
            hdr_read(a, 0)
            if F.Q5:
                a.i(f"v_subrev_u32 {vr(T_HD)}, 32, {vr(V_LDSWN)}")
                a.lds(f"ds_read_b128 {vr(QHN, 4)}, {vr(T_HD)}", "qh")
        if g == 7:
            hdr_decode(a, 0)
            d8_dma(a, 0, 1, True)                # groups 4-7 of the next super-block (this one's group 7 was read in slots (6, 2) / (7, 0))
            d8_dma(a, 1, 1, True)
        fma_block(a, g, 0)
        d8_reads(a, 0, (g + 1) % 8)
        if g == 2:
            min_mfma(a, 0)
    a.i(f"s_add_u32 {sr(S_WK)}, {sr(S_WK)}, {sr(S_WSTEP)}")
    a.i(f"s_add_u32 {sr(S_D0)}, {sr(S_D0)}, {sr(S_NEXT)}")
    a.i(f"s_add_u32 {sr(S_D1)}, {sr(S_D1)}, {sr(S_NEXT)}")
    a.i(f"s_mov_b32 {sr(S_T0)}, {sr(S_STAGE)}")
    a.i(f"s_mov_b32 {sr(S_STAGE)}, {sr(S_NSTAGE)}")
    a.i(f"s_mov_b32 {sr(S_NSTAGE)}, {sr(S_T0)}")
    a.i(f"v_mov_b32 {vr(T_DW)}, {vr(V_LDSW)}")
    a.i(f"v_mov_b32 {vr(V_LDSW)}, {vr(V_LDSWN)}")
    a.i(f"v_mov_b32 {vr(V_LDSWN)}, {vr(T_DW)}")
    a.i(f"v_sub_u32 {vr(V_LDSHN)}, {vr(V_LDSWN)}, {vr(V_HOFF)}")
    a.i(f"s_sub_u32 {sr(S_NSB)}, {sr(S_NSB)}, 1")
    a.i(f"s_cmp_lg_u32 {sr(S_NSB)}, 0")
    a.wait_lg("d8_1")                            # token tile 1's group-7 scales (slot (7, 0)): the top of the body reads them without a wait of its own
    a.i(f"s_cbranch_scc1 L_sb_{label}%=")
    assert a.vm == vm0[len(vm0) - len(a.vm):] and a.lg == lg0, (a.vm, vm0, a.lg, lg0)
    Asm.armed = False
    fma_block(a, 7, 2)
    a.i("s_waitcnt vmcnt(0) lgkmcnt(0)")
    a.i("s_nop 7")
    a.i("s_nop 7")
    a.i(f"s_branch L_end_{label}%=")
    a.i(f"L_cold_{label}%=:")
    bmin_prep(a, 0, scaled=True)
    a.i("s_nop 7")                               # the operand's last register was written by the instruction before: without wait states the
                                                 # first MFMA below read a stale one (token tile 0 of every one-tile row off by ~1 %); in gen()
                                                 # row tile 1's preparation sits between row tile 0's and the first MFMA
    z = CSET[1]                                  # free here: tile (2, 2)'s FMAs are done, MFMA(3, 2) comes after
    for ti in (0, 2):
        acc = ACC + 16 * ti
        a.i(f"v_mfma_f32_32x32x16_f16 {vr(z, 16)}, {vr(S8[ti >> 1], 4)}, {vr(BMIN[0], 4)}, 0")
        a.i("s_nop 7")
        a.i("s_nop 7")
        for j in range(0, 16, 2):
            a.i(f"v_pk_fma_f32 {vr(acc + j, 2)}, {vr(z + j, 2)}, {sr(S_256, 2)}, {vr(acc + j, 2)}")
        a.i("s_nop 3")
    a.i(f"s_branch L_warm_{label}%=")
    a.i(f"L_end_{label}%=:")
    F, R1 = Q4K, False
    return a


V_LDSDT = 214                # one-tile loops: LDS address of the CURRENT super-block's token-scale table (two tables, used alternately)
V_DSUM = 215                 # ... the sum of this lane's two table addresses (the other table = sum - this one)
S_DTAB = 82                  # ... LDS address of the table the next super-block's scales are copied into
S_DSUM = 84                  # ... the sum of the two tables' addresses


def t1_d8_dma(a, nxt):
    """one-tile loops: all eight groups' token scales of one super-block (1 KB: two 32-lane DMA instructions) into the table S_DTAB points
    at; the token-tile-1 table of the two-tile loops is the second table here, so a request has a whole iteration of lead"""
    for half in range(2):
        a.i(f"s_add_u32 {sr(S_T0)}, {sr(S_D0)}, {512 * half}")
        if nxt:
            a.i(f"s_add_u32 {sr(S_T0)}, {sr(S_T0)}, {sr(S_NEXT)}")
        a.i(f"s_add_u32 m0, {sr(S_DTAB)}, {512 * half}")
        a.i(f"s_mov_b64 {sr(S_EXEC, 2)}, exec")
        a.i("s_mov_b64 exec, 0xffffffff")
        a.vmem(f"buffer_load_dwordx4 {vr(V_LANE16)}, {sr(S_ARSRC, 4)}, {sr(S_T0)} offen lds", f"d8t{half}")
        a.i(f"s_mov_b64 exec, {sr(S_EXEC, 2)}")


def t1_d8_reads(a, target):
    if target == 0:
        a.wait_vm("d8t1")                # this super-block's table: requested at the top of the iteration before (or in the prologue)
    for k in range(4):
        a.lds(f"ds_read_b128 {vr(D8[0] + 4 * k, 4)}, {vr(V_LDSDT)} offset:{128 * target + 16 * k}", "d8_0" if k == 3 else "d8_0x")


def t1_common_prologue(a, mid=None):
    a.i(f"s_add_u32 {sr(S_DTAB)}, {sr(S_LDS)}, {2 * F.STAGE}")
    a.i(f"s_lshl_b32 {sr(S_DSUM)}, {sr(S_DTAB)}, 1")
    a.i(f"s_add_u32 {sr(S_DSUM)}, {sr(S_DSUM)}, 1024")
    a.i(f"v_mov_b32 {vr(V_LDSDT)}, {vr(V_LDSD)}")
    a.i(f"v_lshlrev_b32 {vr(V_DSUM)}, 1, {vr(V_LDSD)}")
    a.i(f"v_add_u32 {vr(V_DSUM)}, 1024, {vr(V_DSUM)}")
    a.i(f"s_mov_b32 {sr(S_NEXT)}, 0")
    t1_d8_dma(a, False)                                           # the first super-block's scales -> table 0
    a.i(f"s_add_u32 {sr(S_DTAB)}, {sr(S_DTAB)}, 1024")             # the next one's go to table 1
    if mid:
        mid()                                                     # (vector-memory operations that the loop body issues before its last fragment load)
    a.vmem(f"buffer_load_dwordx4 {vr(ACT[0][0], 4)}, {vr(V_LANE16)}, {sr(S_ARSRC, 4)}, {sr(S_F0)} offen", "act0_0")
    a.i(f"s_add_u32 {sr(S_F0)}, {sr(S_F0)}, 0x400")
    for k in range(16):
        a.i(f"v_mov_b32 {vr(CSET[1] + k)}, {vr(MAGICV + k)}")     # the first "previous tile" FMA block, group 7: adds zero
    for k in range(16):
        a.i(f"v_mov_b32 {vr(D8[0] + k)}, 0")
    for k in range(4, 8):
        a.i(f"v_mov_b32 {vr(DWNM[0] + k)}, 0")


def t1_act_load(a, g):
    """the fragment of group g + 1 (token tile 0 only), then the running offset to group g + 2"""
    par = (g + 1) & 1
    a.vmem(f"buffer_load_dwordx4 {vr(ACT[par][0], 4)}, {vr(V_LANE16)}, {sr(S_ARSRC, 4)}, {sr(S_F0)} offen", f"act0_{par}")
    a.i(f"s_add_u32 {sr(S_F0)}, {sr(S_F0)}, {sr(S_INC6) if g == 6 else '0x400'}")


def gen_t1(label, fmt=None):
    """Q4_K / Q5_K, ONE MFMA tile per wave and group (32 rows x 32 tokens; accumulators v[0:15]): batches of 17 - 32 tokens.  One slot per
    group: MFMA(g) -> loads (the next fragment, weight DMA in groups 0 .. 3) -> unpack of group g + 1 (the operand is free once the
    MFMA has issued) -> the FMAs of group g - 1 -> the row scales of groups g, g + 1 (even g: they may only change once the FMAs of
    group g - 1 have read the old ones) -> the token scales of group g.  The scales of a whole super-block are copied by LDS-DMA one
    iteration ahead into the second of two tables (the two-tile loops' token-tile-1 table)."""
    global F, R1, T1
    fmt = fmt or Q4K_R1
    F, R1, T1 = fmt, True, True
    a = Asm()
    for s in (S_NEGM, S_NEGM + 1):
        a.i(f"s_mov_b32 {sr(s)}, 0xcb400000")
    for s in (S_256, S_256 + 1):
        a.i(f"s_mov_b32 {sr(s)}, 0x43800000")
    a.i(f"s_mov_b32 {sr(S_MASK0F)}, 0x0f0f0f0f")
    a.i(f"s_mov_b32 {sr(S_MASK10)}, 0x10101010")
    a.i(f"s_mov_b32 {sr(S_1024)}, 0x44800000")
    a.i(f"v_cmp_ne_u32_e64 {sr(S_HI, 2)}, {F.QS_OFF}, {vr(V_HOFF)}")
    a.i(f"s_mov_b32 {sr(S_STAGE)}, {sr(S_LDS)}")
    a.i(f"s_add_u32 {sr(S_NSTAGE)}, {sr(S_LDS)}, {F.STAGE}")
    a.i(f"s_mov_b32 {sr(S_WKN)}, {sr(S_WK)}")
    for j in range(F.N_DMA):
        dma_instr(a, j, S_STAGE)
    a.vmem(f"buffer_load_dwordx4 {vr(S8[0], 4)}, {vr(V_LANE16)}, {sr(S_ARSRC, 4)}, {sr(S_S8)} offen", "s8_0")
    t1_common_prologue(a)
    a.i(f"v_mov_b32 {vr(V_LDSW)}, {vr(V_LDSW0)}")
    a.i(f"v_add_u32 {vr(V_LDSWN)}, {F.STAGE}, {vr(V_LDSW0)}")
    a.i(f"v_sub_u32 {vr(V_LDSHN)}, {vr(V_LDSW0)}, {vr(V_HOFF)}")
    a.wait_vm("dma")
    hdr_read(a, 0)
    if F.Q5:
        a.i(f"v_subrev_u32 {vr(T_HD)}, 32, {vr(V_LDSW)}")
        a.lds(f"ds_read_b128 {vr(QH, 4)}, {vr(T_HD)}", "qh")
    a.lds(f"ds_read_b128 {vr(RAW[0], 4)}, {vr(V_LDSW)} offset:0", "raw0")
    hdr_decode(a, 0)
    w_lo(a, 0, 0)
    a.i(f"v_sub_u32 {vr(V_LDSHN)}, {vr(V_LDSWN)}, {vr(V_HOFF)}")
    Asm.armed = True
    a.i(f"L_sb_{label}%=:")
    if a.vm[:1] == ["s8_0"]:
        a.vm = a.vm[1:]
    vm0, lg0 = list(a.vm), list(a.lg)
    a.i(f"s_cmp_eq_u32 {sr(S_NSB)}, 1")
    a.i(f"s_cselect_b32 {sr(S_T0)}, 0, {sr(S_SBSTRIDE)}")
    a.i(f"s_cselect_b32 {sr(S_T1)}, 0, {sr(S_WSTEP)}")
    a.i(f"s_sub_u32 {sr(S_INC6)}, {sr(S_T0)}, {7 * 1024}")
    a.i(f"s_add_u32 {sr(S_S8)}, {sr(S_S8)}, {sr(S_T0)}")
    a.i(f"s_add_u32 {sr(S_WKN)}, {sr(S_WK)}, {sr(S_T1)}")
    a.i(f"s_mov_b32 {sr(S_NEXT)}, {sr(S_T0)}")
    for g in range(8):
        q = g >> 1
        mfma(a, g, 0)
        if g == 1:
            t1_d8_dma(a, True)                   # the next super-block's token scales into the other table (the last one: its own again,
                                                 # harmless) — after group 0's read of THIS table, whose wait would otherwise cover it
        t1_act_load(a, g)
        for j in (2 * g, 2 * g + 1):
            if j < F.N_DMA:
                dma_instr(a, j, S_NSTAGE)
        if g == 1:
            bmin_prep(a, 0)
        if g == 6:
            a.wait_vm("dma")
        if g % 2 == 0:
            w_hi(a, 0, g + 1)
            raw_read(a, 0, q + 1)
        else:
            if g == 7 and F.Q5:
                a.wait_lg("qh")
                for k in range(4):
                    a.i(f"v_mov_b32 {vr(QH + k)}, {vr(QHN + k)}")
            w_lo(a, 0, g + 1)
        if g == 6:
            hdr_read(a, 0)
            if F.Q5:
                a.i(f"v_subrev_u32 {vr(T_HD)}, 32, {vr(V_LDSWN)}")
                a.lds(f"ds_read_b128 {vr(QHN, 4)}, {vr(T_HD)}", "qh")
        pg = (g - 1) % 8
        fma_block(a, pg, 0)
        if g % 2 == 0:
            dw_prep(a, q, 0)                     # (after the FMAs of group g - 1, which read the odd group's old scales)
        t1_d8_reads(a, g)
        if pg == 2:
            min_mfma(a, 0)
            a.i(f"v_cmp_nle_f32_e64 {sr(S_BIG, 2)}, |{vr(HDR[0] + 6)}|, {sr(S_1024)}")
            a.i(f"s_cmp_lg_u64 {sr(S_BIG, 2)}, 0")
            a.i(f"s_cbranch_scc1 L_cold_{label}%=")
            a.i(f"L_warm_{label}%=:")
            a.vmem(f"buffer_load_dwordx4 {vr(S8[0], 4)}, {vr(V_LANE16)}, {sr(S_ARSRC, 4)}, {sr(S_S8)} offen", "s8_0")
        if g == 7:
            hdr_decode(a, 0)
    a.i(f"s_add_u32 {sr(S_WK)}, {sr(S_WK)}, {sr(S_WSTEP)}")
    a.i(f"s_add_u32 {sr(S_D0)}, {sr(S_D0)}, {sr(S_NEXT)}")
    a.i(f"s_sub_u32 {sr(S_DTAB)}, {sr(S_DSUM)}, {sr(S_DTAB)}")     # the other table (an xor of 0x400 is NOT it: the tables' addresses are not 2 KB-aligned)
    a.i(f"v_sub_u32 {vr(V_LDSDT)}, {vr(V_DSUM)}, {vr(V_LDSDT)}")
    a.i(f"s_mov_b32 {sr(S_T0)}, {sr(S_STAGE)}")
    a.i(f"s_mov_b32 {sr(S_STAGE)}, {sr(S_NSTAGE)}")
    a.i(f"s_mov_b32 {sr(S_NSTAGE)}, {sr(S_T0)}")
    a.i(f"v_mov_b32 {vr(T_DW)}, {vr(V_LDSW)}")
    a.i(f"v_mov_b32 {vr(V_LDSW)}, {vr(V_LDSWN)}")
    a.i(f"v_mov_b32 {vr(V_LDSWN)}, {vr(T_DW)}")
    a.i(f"v_sub_u32 {vr(V_LDSHN)}, {vr(V_LDSWN)}, {vr(V_HOFF)}")
    a.i(f"s_sub_u32 {sr(S_NSB)}, {sr(S_NSB)}, 1")
    a.i(f"s_cmp_lg_u32 {sr(S_NSB)}, 0")
    a.wait_lg("d8_0")                            # group 7's token scales: the top of the body reads them without a wait of its own
    a.i(f"s_cbranch_scc1 L_sb_{label}%=")
    assert a.vm == vm0[len(vm0) - len(a.vm):] and a.lg == lg0[len(lg0) - len(a.lg):], (a.vm, vm0, a.lg, lg0)
    Asm.armed = False
    fma_block(a, 7, 0)
    a.i("s_waitcnt vmcnt(0) lgkmcnt(0)")
    a.i("s_nop 7")
    a.i("s_nop 7")
    a.i(f"s_branch L_end_{label}%=")
    a.i(f"L_cold_{label}%=:")
    bmin_prep(a, 0, scaled=True)
    a.i("s_nop 7")
    z = CSET[0]                                  # free here: group 2's FMAs are done (result set 0), MFMA(3) went to set 1, MFMA(4) comes after
    a.i(f"v_mfma_f32_32x32x16_f16 {vr(z, 16)}, {vr(S8[0], 4)}, {vr(BMIN[0], 4)}, 0")
    a.i("s_nop 7")
    a.i("s_nop 7")
    for j in range(0, 16, 2):
        a.i(f"v_pk_fma_f32 {vr(ACC + j, 2)}, {vr(z + j, 2)}, {sr(S_256, 2)}, {vr(ACC + j, 2)}")
    a.i("s_nop 3")
    a.i(f"s_branch L_warm_{label}%=")
    a.i(f"L_end_{label}%=:")
    F, R1, T1 = Q4K, False, False
    return a


def q80_reads(a, rt, g):
    """the two aligned chunks that hold group g's quant bytes of this lane's row and K half, and the block's d (g = 8, 9: groups 0, 1 of
    the next 256 elements; stage buffer = (g >> 2) & 1)"""
    buf, gl = (g >> 2) & 1, g & 3
    base = buf * F.STAGE + rt * 32 * 144
    a.lds(f"ds_read_b128 {vr(RAW8[rt], 4)}, {vr(V_LDSW)} offset:{base + 32 * gl}", f"raw{rt}x")
    a.lds(f"ds_read_b128 {vr(RAW8[rt] + 4, 4)}, {vr(V_LDSW)} offset:{base + 32 * gl + 16}", f"raw{rt}")
    a.lds(f"ds_read_u16 {vr(DREG[rt][g & 1])}, {vr(V_LDSHN)} offset:{base + 34 * gl}", f"d{rt}_{g & 1}")


def q80_w_prep(a, rt, g):
    """int8 operand of group g from the chunk pair: the quant bytes start 2 (g & 3) + 2 bytes into it"""
    a.wait_lg(f"raw{rt}")
    sh = 2 * (g & 3) + 2
    d0, rem = sh // 4, sh % 4
    for i in range(4):
        if rem:
            a.i(f"v_alignbyte_b32 {vr(WOP[rt] + i)}, {vr(RAW8[rt] + i + d0 + 1)}, {vr(RAW8[rt] + i + d0)}, {rem}")
        else:
            a.i(f"v_mov_b32 {vr(WOP[rt] + i)}, {vr(RAW8[rt] + i + d0)}")


def q80_dw_prep(a, rt, g):
    """row scale of group g: dw = d (fp16 -> fp32), nm = -12582912 * dw, as duplicated pairs"""
    p = g & 1
    a.wait_lg(f"d{rt}_{p}")
    a.i(f"v_cvt_f32_f16_e32 {vr(DWNM[rt] + 4 * p)}, {vr(DREG[rt][p])}")
    a.i(f"v_cvt_f32_f16_e32 {vr(DWNM[rt] + 4 * p + 1)}, {vr(DREG[rt][p])}")
    a.i(f"v_pk_mul_f32 {vr(DWNM[rt] + 4 * p + 2, 2)}, {vr(DWNM[rt] + 4 * p, 2)}, {sr(S_NEGM, 2)}")


def q80_dma(a, j, buf, s_run, tag):
    """LDS-DMA instruction j (rows 7 j .. 7 j + 6) of the copy into stage buffer `buf`; s_run = its running source offset"""
    if j:
        a.i(f"s_add_u32 {sr(s_run)}, {sr(s_run)}, {sr(S_RB7)}")
    a.i(f"s_add_u32 m0, {sr(S_LDS)}, {buf * F.STAGE + 16 * 7 * F.CPR * j}")
    if j == F.N_DMA - 1:
        a.i(f"s_mov_b64 {sr(S_EXEC, 2)}, exec")
        mask = (1 << (F.CPR * (F.ROWS - 7 * j))) - 1
        if mask < (1 << 32):
            a.i(f"s_mov_b64 exec, {hex(mask)}")
        else:
            a.i("s_mov_b32 exec_lo, -1")
            a.i(f"s_mov_b32 exec_hi, {hex(mask >> 32)}")
    else:
        a.i("s_nop 0")
    a.vmem(f"buffer_load_dwordx4 {vr(V_DMAOFF)}, {sr(S_WRSRC, 4)}, {sr(s_run)} offen lds", tag)
    if j == F.N_DMA - 1:
        a.i(f"s_mov_b64 exec, {sr(S_EXEC, 2)}")


def gen_q80(label):
    """Q8_0: 256 elements per loop iteration = two 128-element stages in FIXED buffers (A: groups 0-3, B: groups 4-7); A of the next
    iteration is requested in groups 3-4 (after this one's last read of A), B of the next one in group 7 + the next iteration's first
    slots.  No header, no min term: the block's d comes with the quant bytes."""
    global F
    F = Q80
    a = Asm()
    for s in (S_NEGM, S_NEGM + 1):
        a.i(f"s_mov_b32 {sr(s)}, 0xcb400000")
    a.i(f"v_sub_u32 {vr(V_LDSHN)}, {vr(V_LDSW0)}, {vr(V_HOFF)}")  # row base of this lane (hoff = 16 h): where the block headers are read
    a.i(f"v_mov_b32 {vr(V_LDSW)}, {vr(V_LDSW0)}")
    a.i(f"s_mov_b32 {sr(S_RUNA)}, {sr(S_WK)}")
    for j in range(F.N_DMA):
        q80_dma(a, j, 0, S_RUNA, "dmaA")
    a.i(f"s_add_u32 {sr(S_RUNB)}, {sr(S_WK)}, {F.BS}")
    a.i(f"s_mov_b32 {sr(S_NEXT)}, 0")
    d8_dma(a, 0, 0, False)
    d8_dma(a, 1, 0, False)
    for k in range(16):
        a.i(f"v_mov_b32 {vr(CSET[1] + k)}, {vr(MAGICV + k)}")
    for k in range(16):
        a.i(f"v_mov_b32 {vr(D8[1] + k)}, 0")
    for k in range(8):
        a.i(f"v_mov_b32 {vr(DWNM[1] + k)}, 0")
    act_loads(a, 0)
    for s in (S_F0, S_F1):
        a.i(f"s_add_u32 {sr(s)}, {sr(s)}, 0x400")
    for j in range(8):
        q80_dma(a, j, 1, S_RUNB, "dmaB")                          # (the last two: in the loop's first slots, as in every iteration)
    a.wait_vm("dmaA")
    a.wait_vm("d8dma0_1")
    q80_reads(a, 0, 0)
    q80_reads(a, 1, 0)
    q80_w_prep(a, 0, 0)
    q80_reads(a, 0, 1)
    d8_reads(a, 0, 0)
    q80_w_prep(a, 1, 0)
    q80_reads(a, 1, 1)
    # (d of group 0 is in DREG[.][0] — waited for by the first dw_prep; reads in flight: group 1's, token tile 0's scales)
    a.i(f"L_sb_{label}%=:")
    vm0, lg0 = list(a.vm), list(a.lg)
    a.i(f"s_cmp_eq_u32 {sr(S_NSB)}, 1")
    a.i(f"s_cselect_b32 {sr(S_T0)}, 0, {sr(S_SBSTRIDE)}")
    a.i(f"s_cselect_b32 {sr(S_NEXTW)}, 0, {sr(S_WSTEP)}")
    a.i(f"s_sub_u32 {sr(S_INC6)}, {sr(S_T0)}, {7 * 1024}")
    a.i(f"s_mov_b32 {sr(S_NEXT)}, {sr(S_T0)}")
    for g in range(8):
        for ti in range(4):
            rt = ti & 1
            mfma(a, g, ti)
            if ti < 2:
                q80_dw_prep(a, ti, g)
            if ti == 0:
                act_loads(a, (g + 1) & 1)
                advance_offsets(a, g)
            if ti >= 2:
                if (g, ti) == (2, 2):
                    a.wait_vm("dmaB")            # groups 4-7 of this iteration have landed: group 4's chunks are read below
                if (g, ti) == (6, 2):
                    a.wait_vm("dmaA")            # ... groups 0-3 of the next one
                q80_w_prep(a, rt, g + 1)
                q80_reads(a, rt, g + 2)
            # stage copies: B of this iteration's last two instructions, A of the next one in groups 3 / 4, B of the next one in group 7
            if g == 0 and ti < 2:
                q80_dma(a, 8 + ti, 1, S_RUNB, "dmaB")
            if g == 3 or (g == 4 and ti < 2):
                for j in ((2 * ti, 2 * ti + 1) if g == 3 else (8 + ti,)):
                    if j == 0:
                        a.i(f"s_add_u32 {sr(S_RUNA)}, {sr(S_WK)}, {sr(S_NEXTW)}")
                    q80_dma(a, j, 0, S_RUNA, "dmaA")
            if g == 7:
                for j in (2 * ti, 2 * ti + 1):
                    if j == 0:
                        a.i(f"s_add_u32 {sr(S_RUNB)}, {sr(S_WK)}, {sr(S_NEXTW)}")
                        a.i(f"s_add_u32 {sr(S_RUNB)}, {sr(S_RUNB)}, {F.BS}")
                    q80_dma(a, j, 1, S_RUNB, "dmaB")
            if g == 0 and ti >= 2:
                d8_dma(a, ti - 2, 1, False)
            if (g, ti) == (3, 2):
                d8_dma(a, 0, 0, True)
            if (g, ti) == (4, 0):
                d8_dma(a, 1, 0, True)
            pg, pti = (g, ti - 1) if ti else ((g - 1) % 8, 3)
            fma_block(a, pg, pti)
            if pti == 1:
                d8_reads(a, 0, (pg + 1) % 8)
            if pti == 3:
                d8_reads(a, 1, (pg + 1) % 8)
    a.i(f"s_add_u32 {sr(S_WK)}, {sr(S_WK)}, {sr(S_WSTEP)}")
    a.i(f"s_add_u32 {sr(S_D0)}, {sr(S_D0)}, {sr(S_NEXT)}")
    a.i(f"s_add_u32 {sr(S_D1)}, {sr(S_D1)}, {sr(S_NEXT)}")
    a.i(f"s_sub_u32 {sr(S_NSB)}, {sr(S_NSB)}, 1")
    a.i(f"s_cmp_lg_u32 {sr(S_NSB)}, 0")
    a.i(f"s_cbranch_scc1 L_sb_{label}%=")
    assert a.vm == vm0[len(vm0) - len(a.vm):] and a.lg == lg0[len(lg0) - len(a.lg):], (a.vm, vm0, a.lg, lg0)
    fma_block(a, 7, 3)
    a.i("s_waitcnt vmcnt(0) lgkmcnt(0)")
    F = Q4K
    return a


def gen_q80_r1(label):
    """Q8_0, one row tile per wave (32 rows x 64 tokens): gen_q80's pipeline with two slots per group, as gen_r1 is to gen.  Both stage
    copies of the next 256 elements are issued inside the iteration (A in groups 2-3, B in groups 6-7: five instructions each)."""
    global F, R1
    F, R1 = Q80_R1, True
    a = Asm()
    for s in (S_NEGM, S_NEGM + 1):
        a.i(f"s_mov_b32 {sr(s)}, 0xcb400000")
    a.i(f"v_sub_u32 {vr(V_LDSHN)}, {vr(V_LDSW0)}, {vr(V_HOFF)}")
    a.i(f"v_mov_b32 {vr(V_LDSW)}, {vr(V_LDSW0)}")
    a.i(f"s_mov_b32 {sr(S_RUNA)}, {sr(S_WK)}")
    for j in range(F.N_DMA):
        q80_dma(a, j, 0, S_RUNA, "dmaA")
    a.i(f"s_add_u32 {sr(S_RUNB)}, {sr(S_WK)}, {F.BS}")
    a.i(f"s_mov_b32 {sr(S_NEXT)}, 0")
    d8_dma(a, 0, 0, False)
    d8_dma(a, 1, 0, False)
    for k in range(16):
        a.i(f"v_mov_b32 {vr(CSET[1] + k)}, {vr(MAGICV + k)}")
    for k in range(16):
        a.i(f"v_mov_b32 {vr(D8[1] + k)}, 0")
    for k in range(4, 8):
        a.i(f"v_mov_b32 {vr(DWNM[0] + k)}, 0")
    for j in range(3):                           # (issue order as at the end of the loop body)
        q80_dma(a, j, 1, S_RUNB, "dmaB")
    act_loads(a, 0)
    for s in (S_F0, S_F1):
        a.i(f"s_add_u32 {sr(s)}, {sr(s)}, 0x400")
    for j in range(3, F.N_DMA):
        q80_dma(a, j, 1, S_RUNB, "dmaB")
    d8_dma(a, 0, 1, False)
    d8_dma(a, 1, 1, False)
    a.wait_vm("dmaA")
    a.wait_vm("d8dma0_1")
    q80_reads(a, 0, 0)
    q80_w_prep(a, 0, 0)
    q80_reads(a, 0, 1)
    d8_reads(a, 0, 0)
    Asm.armed = True
    a.i(f"L_sb_{label}%=:")
    vm0, lg0 = list(a.vm), list(a.lg)
    a.i(f"s_cmp_eq_u32 {sr(S_NSB)}, 1")
    a.i(f"s_cselect_b32 {sr(S_T0)}, 0, {sr(S_SBSTRIDE)}")
    a.i(f"s_cselect_b32 {sr(S_NEXTW)}, 0, {sr(S_WSTEP)}")
    a.i(f"s_sub_u32 {sr(S_INC6)}, {sr(S_T0)}, {7 * 1024}")
    a.i(f"s_mov_b32 {sr(S_NEXT)}, {sr(S_T0)}")
    for g in range(8):
        # ------------------------------------------------ slot (g, 0)
        mfma(a, g, 0)
        q80_dw_prep(a, 0, g)
        act_loads(a, (g + 1) & 1)
        advance_offsets(a, g)
        if g in (2, 3):                          # stage A of the next 256 elements (this iteration's last read of A: slot (1, 2))
            for j in ((0, 1, 2) if g == 2 else (3, 4)):
                if j == 0:
                    a.i(f"s_add_u32 {sr(S_RUNA)}, {sr(S_WK)}, {sr(S_NEXTW)}")
                q80_dma(a, j, 0, S_RUNA, "dmaA")
        if g in (6, 7):                          # stage B of the next 256 elements (last read of B: slot (5, 2))
            for j in ((0, 1, 2) if g == 6 else (3, 4)):
                if j == 0:
                    a.i(f"s_add_u32 {sr(S_RUNB)}, {sr(S_WK)}, {sr(S_NEXTW)}")
                    a.i(f"s_add_u32 {sr(S_RUNB)}, {sr(S_RUNB)}, {F.BS}")
                q80_dma(a, j, 1, S_RUNB, "dmaB")
        if g == 3:
            d8_dma(a, 0, 0, True)
        pg = (g - 1) % 8
        fma_block(a, pg, 2)
        d8_reads(a, 1, (pg + 1) % 8)
        # ------------------------------------------------ slot (g, 2)
        mfma(a, g, 2)
        if g == 2:
            a.wait_vm("dmaB")                    # groups 4-7 of this iteration have landed: group 4's chunks are read below
        if g == 6:
            a.wait_vm("dmaA")                    # ... groups 0-3 of the next one
        q80_w_prep(a, 0, g + 1)
        q80_reads(a, 0, g + 2)
        if g == 3:
            d8_dma(a, 1, 0, True)
        if g == 7:
            d8_dma(a, 0, 1, True)
            d8_dma(a, 1, 1, True)
        fma_block(a, g, 0)
        d8_reads(a, 0, (g + 1) % 8)
    a.i(f"s_add_u32 {sr(S_WK)}, {sr(S_WK)}, {sr(S_WSTEP)}")
    a.i(f"s_add_u32 {sr(S_D0)}, {sr(S_D0)}, {sr(S_NEXT)}")
    a.i(f"s_add_u32 {sr(S_D1)}, {sr(S_D1)}, {sr(S_NEXT)}")
    a.i(f"s_sub_u32 {sr(S_NSB)}, {sr(S_NSB)}, 1")
    a.i(f"s_cmp_lg_u32 {sr(S_NSB)}, 0")
    a.wait_lg("d8_1")
    a.i(f"s_cbranch_scc1 L_sb_{label}%=")
    assert a.vm == vm0[len(vm0) - len(a.vm):] and a.lg == lg0[len(lg0) - len(a.lg):], (a.vm, vm0, a.lg, lg0)
    Asm.armed = False
    fma_block(a, 7, 2)
    a.i("s_waitcnt vmcnt(0) lgkmcnt(0)")
    F, R1 = Q4K, False
    return a


def q40_reads(a, rt, g):
    """the two aligned chunks that hold block g's 16 quant bytes of this lane's row, and the block's d (g = 8, 9: blocks 0, 1 of the
    next stage)"""
    base_reg = V_LDSW if g < 8 else V_LDSWN
    gl = g & 7
    c = (18 * gl + 2) // 16
    base = rt * 32 * 144
    a.lds(f"ds_read_b128 {vr(RAW8[rt], 4)}, {vr(base_reg)} offset:{base + 16 * c}", f"raw{rt}x")
    a.lds(f"ds_read_b128 {vr(RAW8[rt] + 4, 4)}, {vr(base_reg)} offset:{base + 16 * c + 16}", f"raw{rt}")
    a.lds(f"ds_read_u16 {vr(DREG[rt][g & 1])}, {vr(base_reg)} offset:{base + 18 * gl}", f"d{rt}_{g & 1}")


def q40_w_prep(a, rt, g):
    """int8 operand of block g: the 16 quant bytes start (18 g + 2) % 16 bytes into the chunk pair; this lane half's nibbles -> 16 (n - 8)"""
    a.wait_lg(f"raw{rt}")
    sh = (18 * (g & 7) + 2) % 16
    d0, rem = sh // 4, sh % 4
    for i in range(4):
        if rem:
            a.i(f"v_alignbyte_b32 {vr(T_WHI + i)}, {vr(RAW8[rt] + i + d0 + 1)}, {vr(RAW8[rt] + i + d0)}, {rem}")
        else:
            a.i(f"v_xor_b32 {vr(T_WHI + i)}, {vr(V_XC)}, {vr(RAW8[rt] + i + d0)}")
    if rem:
        for i in range(4):
            a.i(f"v_xor_b32 {vr(T_WHI + i)}, {vr(V_XC)}, {vr(T_WHI + i)}")
    for i in range(4):
        a.i(f"v_lshlrev_b32 {vr(T_WHI + i)}, {vr(V_SH)}, {vr(T_WHI + i)}")
    for i in range(4):
        a.i(f"v_and_b32 {vr(WOP[rt] + i)}, {sr(S_MASKF0)}, {vr(T_WHI + i)}")


def q40_dw_prep(a, rt, g):
    """row scale of block g: dw = d / 16 (the operand is 16 (n - 8)), nm = -12582912 * dw, as duplicated pairs"""
    p = g & 1
    a.wait_lg(f"d{rt}_{p}")
    a.i(f"v_cvt_f32_f16_e32 {vr(T_DW)}, {vr(DREG[rt][p])}")
    a.i(f"v_cvt_f32_f16_e32 {vr(T_DW + 1)}, {vr(DREG[rt][p])}")
    a.i(f"v_pk_mul_f32 {vr(DWNM[rt] + 4 * p, 2)}, {vr(T_DW, 2)}, {sr(S_SIXTEENTH, 2)}")
    a.i(f"v_pk_mul_f32 {vr(DWNM[rt] + 4 * p + 2, 2)}, {vr(T_DW, 2)}, {sr(S_NEGM16, 2)}")


def q40_prologue_consts(a):
    for s in (S_SIXTEENTH, S_SIXTEENTH + 1):
        a.i(f"s_mov_b32 {sr(s)}, 0x3d800000")                    # 0.0625f
    for s in (S_NEGM16, S_NEGM16 + 1):
        a.i(f"s_mov_b32 {sr(s)}, 0xc9400000")                    # -12582912 / 16 = -786432.0f
    a.i(f"s_mov_b32 {sr(S_MASKF0)}, 0xf0f0f0f0")
    a.i(f"v_lshrrev_b32 {vr(V_XC)}, 2, {vr(V_HOFF)}")             # hoff = 16 h -> 0 / 4
    a.i(f"v_sub_u32 {vr(V_SH)}, 4, {vr(V_XC)}")                   # shift: 4 (low-nibble half) / 0
    a.i(f"v_mov_b32 {vr(T_DW)}, 0x08080808")
    a.i(f"v_lshlrev_b32 {vr(V_XC)}, {vr(V_XC)}, {vr(T_DW)}")      # xor constant: 0x08 / 0x80 per byte
    a.i(f"v_mov_b32 {vr(V_LDSW)}, {vr(V_LDSW0)}")                 # (ldsw0 = the row's stage base: both lane halves read the same bytes)
    a.i(f"v_add_u32 {vr(V_LDSWN)}, {F.STAGE}, {vr(V_LDSW0)}")
    a.i(f"s_mov_b32 {sr(S_STAGE)}, {sr(S_LDS)}")
    a.i(f"s_add_u32 {sr(S_NSTAGE)}, {sr(S_LDS)}, {F.STAGE}")
    a.i(f"s_mov_b32 {sr(S_WKN)}, {sr(S_WK)}")


def q40_loop_end(a):
    a.i(f"s_add_u32 {sr(S_WK)}, {sr(S_WK)}, {sr(S_WSTEP)}")
    a.i(f"s_add_u32 {sr(S_D0)}, {sr(S_D0)}, {sr(S_NEXT)}")
    a.i(f"s_add_u32 {sr(S_D1)}, {sr(S_D1)}, {sr(S_NEXT)}")
    a.i(f"s_mov_b32 {sr(S_T0)}, {sr(S_STAGE)}")
    a.i(f"s_mov_b32 {sr(S_STAGE)}, {sr(S_NSTAGE)}")
    a.i(f"s_mov_b32 {sr(S_NSTAGE)}, {sr(S_T0)}")
    a.i(f"v_mov_b32 {vr(T_DW)}, {vr(V_LDSW)}")
    a.i(f"v_mov_b32 {vr(V_LDSW)}, {vr(V_LDSWN)}")
    a.i(f"v_mov_b32 {vr(V_LDSWN)}, {vr(T_DW)}")
    a.i(f"s_sub_u32 {sr(S_NSB)}, {sr(S_NSB)}, 1")
    a.i(f"s_cmp_lg_u32 {sr(S_NSB)}, 0")


def q40_loop_top(a):
    a.i(f"s_cmp_eq_u32 {sr(S_NSB)}, 1")
    a.i(f"s_cselect_b32 {sr(S_T0)}, 0, {sr(S_SBSTRIDE)}")
    a.i(f"s_cselect_b32 {sr(S_T1)}, 0, {sr(S_WSTEP)}")
    a.i(f"s_sub_u32 {sr(S_INC6)}, {sr(S_T0)}, {7 * 1024}")
    a.i(f"s_add_u32 {sr(S_WKN)}, {sr(S_WK)}, {sr(S_T1)}")
    a.i(f"s_mov_b32 {sr(S_NEXT)}, {sr(S_T0)}")


def gen_q40(label):
    """Q4_0: Q8_0's per-group pipeline (chunk-pair reads two groups ahead, operand one group ahead, the block's d beside them) on Q4_K's
    two-stage ring (one stage = 256 elements, the next stage's ten DMA instructions in groups 0 .. 4)."""
    global F
    F = Q40
    a = Asm()
    for s in (S_NEGM, S_NEGM + 1):
        a.i(f"s_mov_b32 {sr(s)}, 0xcb400000")
    q40_prologue_consts(a)
    for j in range(F.N_DMA):
        dma_instr(a, j, S_STAGE)
    a.i(f"s_mov_b32 {sr(S_NEXT)}, 0")
    d8_dma(a, 0, 0, False)
    d8_dma(a, 1, 0, False)
    for k in range(16):
        a.i(f"v_mov_b32 {vr(CSET[1] + k)}, {vr(MAGICV + k)}")
    for k in range(16):
        a.i(f"v_mov_b32 {vr(D8[1] + k)}, 0")
    for k in range(8):
        a.i(f"v_mov_b32 {vr(DWNM[1] + k)}, 0")
    act_loads(a, 0)
    for s in (S_F0, S_F1):
        a.i(f"s_add_u32 {sr(s)}, {sr(s)}, 0x400")
    a.wait_vm("dma")
    a.wait_vm("d8dma0_1")
    q40_reads(a, 0, 0)
    q40_reads(a, 1, 0)
    q40_w_prep(a, 0, 0)
    q40_reads(a, 0, 1)
    d8_reads(a, 0, 0)
    q40_w_prep(a, 1, 0)
    q40_reads(a, 1, 1)
    Asm.armed = True
    a.i(f"L_sb_{label}%=:")
    vm0, lg0 = list(a.vm), list(a.lg)
    q40_loop_top(a)
    for g in range(8):
        for ti in range(4):
            rt = ti & 1
            mfma(a, g, ti)
            if ti < 2:
                q40_dw_prep(a, ti, g)
            if ti == 0:
                act_loads(a, (g + 1) & 1)
                advance_offsets(a, g)
            if ti >= 2:
                if (g, ti) == (6, 2):
                    a.wait_vm("dma")             # the next stage has landed: its blocks 0, 1 are read in groups 6, 7
                q40_w_prep(a, rt, g + 1)
                q40_reads(a, rt, g + 2)
            if g <= 4 and ti < 2 and 2 * g + ti < F.N_DMA:
                dma_instr(a, 2 * g + ti, S_NSTAGE)
            if g == 0 and ti >= 2:
                d8_dma(a, ti - 2, 1, False)
            if (g, ti) == (3, 2):
                d8_dma(a, 0, 0, True)
            if (g, ti) == (4, 0):
                d8_dma(a, 1, 0, True)
            pg, pti = (g, ti - 1) if ti else ((g - 1) % 8, 3)
            fma_block(a, pg, pti)
            if pti == 1:
                d8_reads(a, 0, (pg + 1) % 8)
            if pti == 3:
                d8_reads(a, 1, (pg + 1) % 8)
    q40_loop_end(a)
    a.i(f"s_cbranch_scc1 L_sb_{label}%=")
    assert a.vm == vm0[len(vm0) - len(a.vm):] and a.lg == lg0[len(lg0) - len(a.lg):], (a.vm, vm0, a.lg, lg0)
    Asm.armed = False
    fma_block(a, 7, 3)
    a.i("s_waitcnt vmcnt(0) lgkmcnt(0)")
    F = Q4K
    return a


def gen_q40_r1(label):
    """Q4_0, one row tile per wave: gen_q40 with two slots per group (as gen_q80_r1 is to gen_q80); the next stage's five DMA instructions
    in groups 0 .. 2."""
    global F, R1
    F, R1 = Q40_R1, True
    a = Asm()
    for s in (S_NEGM, S_NEGM + 1):
        a.i(f"s_mov_b32 {sr(s)}, 0xcb400000")
    q40_prologue_consts(a)
    for j in range(F.N_DMA):
        dma_instr(a, j, S_STAGE)
    a.i(f"s_mov_b32 {sr(S_NEXT)}, 0")
    d8_dma(a, 0, 0, False)
    d8_dma(a, 1, 0, False)
    for k in range(16):
        a.i(f"v_mov_b32 {vr(CSET[1] + k)}, {vr(MAGICV + k)}")
    for k in range(16):
        a.i(f"v_mov_b32 {vr(D8[1] + k)}, 0")
    for k in range(4, 8):
        a.i(f"v_mov_b32 {vr(DWNM[0] + k)}, 0")
    act_loads(a, 0)
    for s in (S_F0, S_F1):
        a.i(f"s_add_u32 {sr(s)}, {sr(s)}, 0x400")
    d8_dma(a, 0, 1, False)
    d8_dma(a, 1, 1, False)
    a.wait_vm("dma")
    a.wait_vm("d8dma0_1")
    q40_reads(a, 0, 0)
    q40_w_prep(a, 0, 0)
    q40_reads(a, 0, 1)
    d8_reads(a, 0, 0)
    Asm.armed = True
    a.i(f"L_sb_{label}%=:")
    vm0, lg0 = list(a.vm), list(a.lg)
    q40_loop_top(a)
    for g in range(8):
        mfma(a, g, 0)
        q40_dw_prep(a, 0, g)
        act_loads(a, (g + 1) & 1)
        advance_offsets(a, g)
        for j in (2 * g, 2 * g + 1):
            if j < F.N_DMA:
                dma_instr(a, j, S_NSTAGE)
        if g == 3:
            d8_dma(a, 0, 0, True)
        pg = (g - 1) % 8
        fma_block(a, pg, 2)
        d8_reads(a, 1, (pg + 1) % 8)
        mfma(a, g, 2)
        if g == 6:
            a.wait_vm("dma")
        q40_w_prep(a, 0, g + 1)
        q40_reads(a, 0, g + 2)
        if g == 3:
            d8_dma(a, 1, 0, True)
        if g == 7:
            d8_dma(a, 0, 1, True)
            d8_dma(a, 1, 1, True)
        fma_block(a, g, 0)
        d8_reads(a, 0, (g + 1) % 8)
    q40_loop_end(a)
    a.wait_lg("d8_1")
    a.i(f"s_cbranch_scc1 L_sb_{label}%=")
    assert a.vm == vm0[len(vm0) - len(a.vm):] and a.lg == lg0[len(lg0) - len(a.lg):], (a.vm, vm0, a.lg, lg0)
    Asm.armed = False
    fma_block(a, 7, 2)
    a.i("s_waitcnt vmcnt(0) lgkmcnt(0)")
    F, R1 = Q4K, False
    return a


def gen_q80_t1(label):
    """Q8_0, one MFMA tile per wave and group (32 rows x 32 tokens): gen_q80_r1 with one slot per group, as gen_t1 is to gen_r1."""
    global F, R1, T1
    F, R1, T1 = Q80_R1, True, True
    a = Asm()
    for s in (S_NEGM, S_NEGM + 1):
        a.i(f"s_mov_b32 {sr(s)}, 0xcb400000")
    a.i(f"v_sub_u32 {vr(V_LDSHN)}, {vr(V_LDSW0)}, {vr(V_HOFF)}")
    a.i(f"v_mov_b32 {vr(V_LDSW)}, {vr(V_LDSW0)}")
    a.i(f"s_mov_b32 {sr(S_RUNA)}, {sr(S_WK)}")
    for j in range(F.N_DMA):
        q80_dma(a, j, 0, S_RUNA, "dmaA")
    a.i(f"s_add_u32 {sr(S_RUNB)}, {sr(S_WK)}, {F.BS}")
    t1_common_prologue(a, lambda: [q80_dma(a, j, 1, S_RUNB, "dmaB") for j in range(3)])
    for j in range(3, F.N_DMA):
        q80_dma(a, j, 1, S_RUNB, "dmaB")
    a.wait_vm("dmaA")
    q80_reads(a, 0, 0)
    q80_w_prep(a, 0, 0)
    q80_reads(a, 0, 1)
    Asm.armed = True
    a.i(f"L_sb_{label}%=:")
    vm0, lg0 = list(a.vm), list(a.lg)
    a.i(f"s_cmp_eq_u32 {sr(S_NSB)}, 1")
    a.i(f"s_cselect_b32 {sr(S_T0)}, 0, {sr(S_SBSTRIDE)}")
    a.i(f"s_cselect_b32 {sr(S_NEXTW)}, 0, {sr(S_WSTEP)}")
    a.i(f"s_sub_u32 {sr(S_INC6)}, {sr(S_T0)}, {7 * 1024}")
    a.i(f"s_mov_b32 {sr(S_NEXT)}, {sr(S_T0)}")
    for g in range(8):
        mfma(a, g, 0)
        q80_dw_prep(a, 0, g)
        if g == 1:
            t1_d8_dma(a, True)
        t1_act_load(a, g)
        if g in (2, 3):                          # stage A of the next 256 elements (this iteration's last read of A: group 1)
            for j in ((0, 1, 2) if g == 2 else (3, 4)):
                if j == 0:
                    a.i(f"s_add_u32 {sr(S_RUNA)}, {sr(S_WK)}, {sr(S_NEXTW)}")
                q80_dma(a, j, 0, S_RUNA, "dmaA")
        if g in (6, 7):                          # stage B of the next 256 elements (last read of B: group 5)
            for j in ((0, 1, 2) if g == 6 else (3, 4)):
                if j == 0:
                    a.i(f"s_add_u32 {sr(S_RUNB)}, {sr(S_WK)}, {sr(S_NEXTW)}")
                    a.i(f"s_add_u32 {sr(S_RUNB)}, {sr(S_RUNB)}, {F.BS}")
                q80_dma(a, j, 1, S_RUNB, "dmaB")
        if g == 2:
            a.wait_vm("dmaB")
        if g == 6:
            a.wait_vm("dmaA")
        q80_w_prep(a, 0, g + 1)
        q80_reads(a, 0, g + 2)
        fma_block(a, (g - 1) % 8, 0)
        t1_d8_reads(a, g)
    a.i(f"s_add_u32 {sr(S_WK)}, {sr(S_WK)}, {sr(S_WSTEP)}")
    a.i(f"s_add_u32 {sr(S_D0)}, {sr(S_D0)}, {sr(S_NEXT)}")
    a.i(f"s_sub_u32 {sr(S_DTAB)}, {sr(S_DSUM)}, {sr(S_DTAB)}")
    a.i(f"v_sub_u32 {vr(V_LDSDT)}, {vr(V_DSUM)}, {vr(V_LDSDT)}")
    a.i(f"s_sub_u32 {sr(S_NSB)}, {sr(S_NSB)}, 1")
    a.i(f"s_cmp_lg_u32 {sr(S_NSB)}, 0")
    a.wait_lg("d8_0")
    a.i(f"s_cbranch_scc1 L_sb_{label}%=")
    assert a.vm == vm0[len(vm0) - len(a.vm):] and a.lg == lg0[len(lg0) - len(a.lg):], (a.vm, vm0, a.lg, lg0)
    Asm.armed = False
    fma_block(a, 7, 0)
    a.i("s_waitcnt vmcnt(0) lgkmcnt(0)")
    F, R1, T1 = Q4K, False, False
    return a


def gen_q40_t1(label):
    """Q4_0, one MFMA tile per wave and group: gen_q40_r1 with one slot per group."""
    global F, R1, T1
    F, R1, T1 = Q40_R1, True, True
    a = Asm()
    for s in (S_NEGM, S_NEGM + 1):
        a.i(f"s_mov_b32 {sr(s)}, 0xcb400000")
    q40_prologue_consts(a)
    for j in range(F.N_DMA):
        dma_instr(a, j, S_STAGE)
    t1_common_prologue(a)
    a.wait_vm("dma")
    q40_reads(a, 0, 0)
    q40_w_prep(a, 0, 0)
    q40_reads(a, 0, 1)
    Asm.armed = True
    a.i(f"L_sb_{label}%=:")
    vm0, lg0 = list(a.vm), list(a.lg)
    q40_loop_top(a)
    for g in range(8):
        mfma(a, g, 0)
        q40_dw_prep(a, 0, g)
        if g == 1:
            t1_d8_dma(a, True)
        t1_act_load(a, g)
        for j in (2 * g, 2 * g + 1):
            if j < F.N_DMA:
                dma_instr(a, j, S_NSTAGE)
        if g == 6:
            a.wait_vm("dma")
        q40_w_prep(a, 0, g + 1)
        q40_reads(a, 0, g + 2)
        fma_block(a, (g - 1) % 8, 0)
        t1_d8_reads(a, g)
    q40_loop_end(a)
    a.i(f"s_sub_u32 {sr(S_DTAB)}, {sr(S_DSUM)}, {sr(S_DTAB)}")
    a.i(f"v_sub_u32 {vr(V_LDSDT)}, {vr(V_DSUM)}, {vr(V_LDSDT)}")
    a.i(f"s_cmp_lg_u32 {sr(S_NSB)}, 0")
    a.wait_lg("d8_0")
    a.i(f"s_cbranch_scc1 L_sb_{label}%=")
    assert a.vm == vm0[len(vm0) - len(a.vm):] and a.lg == lg0[len(lg0) - len(a.lg):], (a.vm, vm0, a.lg, lg0)
    Asm.armed = False
    fma_block(a, 7, 0)
    a.i("s_waitcnt vmcnt(0) lgkmcnt(0)")
    F, R1, T1 = Q4K, False, False
    return a


def emit(a, fn_name):
    asm = "\n".join(f'      "{l}\\n"' for l in a.lines)
    outs_v = set(range(64, N_VGPR if X_INPLACE else 256)) - set(range(MAGICV, MAGICV + 16)) - {V_LANE16, V_LDSD, V_LDSW0, V_HOFF, V_DMAOFF}
    clob_v = ", ".join(f'"v{i}"' for i in sorted(outs_v))
    s_mod = {S_T0, S_T1, S_STAGE, S_NSTAGE, S_INC6, S_WKN, S_NEGM, S_NEGM + 1, S_1024, S_MASK0F, S_HI, S_HI + 1, S_BIG, S_BIG + 1, S_256, S_256 + 1,
             S_EXEC, S_EXEC + 1, S_NEXT, S_RUNA, S_RUNB, S_NEXTW, S_MASKF0, S_MASK10, S_DTAB, S_DSUM, S_SIXTEENTH, S_SIXTEENTH + 1, S_NEGM16, S_NEGM16 + 1}
    clob_s = ", ".join(f'"s{i}"' for i in sorted(s_mod))
    return f'''// GENERATED by scripts/gen_mmq_x64.py — do not edit.  {len(a.lines)} instructions.
static __device__ __forceinline__ void {fn_name}(v32f& acc0, v32f& acc1, const v16i& magic, unsigned lane16, unsigned ldsd, unsigned ldsw0,
                                               unsigned hoff, unsigned dmaoff, __amdgpu_buffer_rsrc_t wrsrc, __amdgpu_buffer_rsrc_t arsrc,
                                               unsigned lds, unsigned nsb, unsigned sbstride, unsigned wk, unsigned rb7, unsigned f0,
                                               unsigned f1, unsigned d0, unsigned d1, unsigned s8, unsigned wstep) {{
  asm volatile(
{asm}
      : "+{{v[0:31]}}"(acc0), "+{{v[32:63]}}"(acc1), "+{{s{S_NSB}}}"(nsb), "+{{s{S_WK}}}"(wk), "+{{s{S_F0}}}"(f0), "+{{s{S_F1}}}"(f1),
        "+{{s{S_D0}}}"(d0), "+{{s{S_D1}}}"(d1), "+{{s{S_S8}}}"(s8)
      : "{{v[{MAGICV}:{MAGICV + 15}]}}"(magic), "{{v{V_LANE16}}}"(lane16), "{{v{V_LDSD}}}"(ldsd), "{{v{V_LDSW0}}}"(ldsw0), "{{v{V_HOFF}}}"(hoff),
        "{{v{V_DMAOFF}}}"(dmaoff), "{{s[{S_WRSRC}:{S_WRSRC + 3}]}}"(wrsrc), "{{s[{S_ARSRC}:{S_ARSRC + 3}]}}"(arsrc), "{{s{S_LDS}}}"(lds),
        "{{s{S_SBSTRIDE}}}"(sbstride), "{{s{S_RB7}}}"(rb7), "{{s{S_WSTEP}}}"(wstep)
      : "memory", "scc", "vcc", "m0", "exec", {clob_s}, {clob_v});
}}
'''


if __name__ == "__main__":
    a = gen("q4k")
    b = gen_q80("q80")
    c = gen_r1("r1_q4k_")
    g5 = gen_r1("r1_q5k_", Q5K_R1)
    t4 = gen_t1("t1_q4k_")
    t5 = gen_t1("t1_q5k_", Q5K_R1)
    t8 = gen_q80_t1("t1_q80_")
    t0 = gen_q40_t1("t1_q40_")
    d = gen_q80_r1("r1_q80_")
    e4 = gen_q40("q40_")
    f4 = gen_q40_r1("r1_q40_")
    if "--list" in sys.argv:
        print("\n".join((b if "q80" in sys.argv else c if "r1" in sys.argv else a).lines))
    with open(os.environ.get("X64_OUT", OUT), "w") as f:
        f.write(emit(a, "x64_loop_q4k"))
        f.write(emit(b, "x64_loop_q80"))
        f.write(emit(c, "x64_loop_q4k_r1"))
        f.write(emit(d, "x64_loop_q80_r1"))
        f.write(emit(e4, "x64_loop_q40"))
        f.write(emit(f4, "x64_loop_q40_r1"))
        f.write(emit(g5, "x64_loop_q5k_r1"))
        f.write(emit(t4, "x64_loop_q4k_t1"))
        f.write(emit(t5, "x64_loop_q5k_t1"))
        f.write(emit(t8, "x64_loop_q80_t1"))
        f.write(emit(t0, "x64_loop_q40_t1"))
    print(len(a.lines), "+", len(b.lines), "instructions ->", OUT, file=sys.stderr)
