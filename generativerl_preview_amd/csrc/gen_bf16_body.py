#!/usr/bin/env python3
"""Generator of the hand-scheduled gfx950 assembly body of the bf16 persistent integrator (engine_bf16.hip.inc).

One solver step of k_sample_bf16 evaluates the whole TemporalSpatialResidualNet for a wave's 32 rows: ~3250
v_mfma_f32_32x32x16_bf16 whose weight (A) fragments stream through the LDS ring and whose activations never leave
registers.  With ONE wave per SIMD every instruction of that wave takes a 4-cycle issue slot (8 for v_exp / v_rcp
and for the MFMA's own issue: MI355X_MICROARCH.md, "vector-instruction ISSUE cost"), so the step cannot run faster
than max(32 x MFMAs, 4 x issue slots).  The compiler-scheduled C++ body needs 8.3 instructions per MFMA (41.9
issue cycles; profiles/r02_isa_counts_before.json): accumulators live in AGPRs and every element is copied out with
v_accvgpr_read, every ordering pin costs an s_nop, every ds_read gets its own s_waitcnt.  This generator emits the
same arithmetic with ~6.4 instructions per MFMA:

  * register plan (512 per lane): VGPR  v[0:63]    two accumulator buffers (tile n computes into one while tile
                                                   n-1's epilogue reads the other: MFMA results are VGPR-form, no
                                                   v_accvgpr_read for the SiLU epilogues)
                                        v[64:191]  h1 of the current residual block (cvt_pk writes it directly)
                                        v[192:215] weight-fragment queue (ds_read_b128 runs PF fragments ahead)
                                        v[216:223] epilogue temporaries
                                        v[224:255] left to the compiler (operands of the asm statement)
                                  AGPR  a[0:31] SiLU(t-embedding) fragments, a[32:47] last_block accumulator,
                                        a[48:175] d0 / later u, d2, up0-out, a[176:239] d1  (MFMA B operands)
  * epilogues (exp2 -> 1+e -> rcp -> fma, residual, bf16 pack) are software-pipelined element by element and spread
    evenly over the MFMA gaps of the NEXT tile; the independent chain of a tile (time-MLP / modify_x) is issued
    first, so that the last epilogue of a layer overlaps the first tile of the next one;
  * one s_waitcnt per TWO weight fragments, bias tiles are ds_read straight into the accumulator buffer a few gaps
    ahead, the LDS-DMA refill (4 x buffer_load ... lds) follows the per-chunk barrier as before.

Everything is a pure function of the network shape: `python gen_bf16_body.py <out.inc>` writes one C++ header with
an inline-asm function per instantiated shape.  Numerics are those of the C++ body (same operation order per
element, same KD-scaled activation domain)."""
import os
import sys

# ---- knobs (environment): the defaults are the shipped schedule; GEN_X are timing experiments with WRONG results.
# A stray GEN_* variable in the build shell must not ship a different kernel: any of them set without
# GEN_EXPERIMENT=1 is refused, and the knob values are written into the generated header (B16_GEN_DEFAULT_KNOBS),
# where engine_bf16_inst.hip static_asserts the shipped schedule unless it is itself built with -DB16_EXPERIMENT.
_KNOB_VARS = sorted(k for k in os.environ if k.startswith("GEN_") and k != "GEN_EXPERIMENT")
if _KNOB_VARS and os.environ.get("GEN_EXPERIMENT") != "1":
    sys.exit("gen_bf16_body.py: %s set in the environment; experiment builds need GEN_EXPERIMENT=1 "
             "(they never go into libgrlhip.so)" % ", ".join(_KNOB_VARS))
DBG_LO = int(os.environ.get("GEN_LO", "2"))
DBG_NOP = int(os.environ.get("GEN_NOP", "0"))
X = set(filter(None, os.environ.get("GEN_X", "").split(",")))   # novalu, nodma, nobarrier, nofrag, nowait, nowaitv
SPREAD = int(os.environ.get("GEN_SPREAD", "-1"))  # > 0: one LDS-DMA piece every SPREAD MFMAs; -1: pieces go into the cheapest gaps (least VALU/LDS work) of the chunk; 0: all after the barrier
WAITB = int(os.environ.get("GEN_WAITB", "4"))    # one s_waitcnt per WAITB weight fragments
BSPREAD = int(os.environ.get("GEN_BSPREAD", "2"))  # > 0: bias ds_reads spread, BSPREAD per gap
SKEW = int(os.environ.get("GEN_SKEW", "0"))      # > 0: after every ring hand-off wave w idles w * SKEW wait states
DPOS = int(os.environ.get("GEN_DPOS", "0"))      # where in an MFMA gap an LDS-DMA piece goes: 0 after the gap's fragment read, 1 at the end of the gap (after its VALU work), 2 right after the MFMA
DMAPOL = os.environ.get("GEN_DMAPOL", "")         # cache-policy bits on the LDS-DMA loads of the weight stream: "", "nt", "sc1", "sc0 sc1" ...
FOLD = int(os.environ.get("GEN_FOLD", "1"))      # 1: the time-MLP bias of a residual block is folded into its dense2 bias by the packer (b2' = b2 + W2 bt): the T chain starts from 0, no bias read
ROT = int(os.environ.get("GEN_ROT", "0"))        # 1: one body per wave; wave w issues its LDS-DMA pieces in the gaps = w (mod 4) of each chunk, so that no two waves of the CU issue a piece in the same MFMA gap
ACC = (0, 32)       # two accumulator buffers of 32 registers
HB = 64             # 32 fragments x 4
QBASE = 192
QSLOTS = int(os.environ.get("GEN_PF", "8"))   # weight-fragment queue depth: 4 AGPR slots (a[240:255]) + the rest in VGPRs
QV = max(QSLOTS - 4, 0)                        # VGPR slots
TMP = QBASE + 4 * QV                           # 8 epilogue temporaries
NV = TMP + 8                                   # VGPRs the body owns: v[0:NV-1]; the rest is the compiler's
A_STE, A_ACCL, A_R1, A_R2 = 0, 32, 48, 176


class Cfg:
    def __init__(self, name, XD, CD, CH, TD, TCH, H0, H1, H2, nslot=None, cf=None, guide=None):
        nslot = nslot or int(os.environ.get("GEN_NSLOT", "4"))
        cf = cf or int(os.environ.get("GEN_CF", "16"))
        self.name = name
        self.XD, self.CD, self.CH, self.TD, self.TCH, self.H0, self.H1, self.H2 = XD, CD, CH, TD, TCH, H0, H1, H2
        self.KS_X = (XD + 15) // 16
        self.KS_T0, self.KS_T, self.NT_T = (TD + CH) // 16, TCH // 16, TCH // 32
        self.K0, self.K1, self.K2 = H0 // 16, H1 // 16, H2 // 16
        self.N0, self.N1, self.N2 = H0 // 32, H1 // 32, H2 // 32
        assert self.KS_X == 1 and H0 == 2 * H1 and H1 == 2 * H2 and H0 <= 512
        # bias table offsets (floats), identical to b16::Cfg in engine_bf16.hip.inc
        self.B_SORT0, self.B_SORT2 = 0, TCH
        self.B_FIRST = 2 * TCH
        self.B_LAST = self.B_FIRST + 4 * H0
        self.B_DOWN0 = self.B_LAST + 32
        self.B_DOWN1 = self.B_DOWN0 + 4 * H1
        self.B_MID = self.B_DOWN1 + 4 * H2
        self.B_UP0 = self.B_MID + 3 * H2
        self.B_UP1 = self.B_UP0 + 3 * H1
        self.nslot, self.cf = nslot, cf
        self.BIAS_FLOATS = self.B_UP1 + 3 * H0
        # energy guidance (b16::GuideCfg): ConcatenateMLP [t_emb(GTD) | x | c] -> G1 -> G2 -> 1, forward + d/dx by hand
        self.guide = guide
        self.KS_C = (CD + 15) // 16
        if guide:
            self.GTD, self.G1, self.G2 = guide
            self.GKT, self.GK1, self.GK2, self.GN1, self.GN2 = self.GTD // 16, self.G1 // 16, self.G2 // 16, self.G1 // 32, self.G2 // 32
            self.GKS1 = self.GKT + self.KS_X + self.KS_C
            self.GB_1, self.GB_2, self.GB_W3 = self.BIAS_FLOATS, self.BIAS_FLOATS + self.G1, self.BIAS_FLOATS + self.G1 + self.G2
            assert self.G1 == self.G2 and self.GK1 * 4 * 2 <= 128
        # park slots (1 KiB per fragment and lane set): [t_emb | c_emb] (KS_T0), x, then for guided bodies the energy
        # net's t_emb fragments and the raw condition fragments
        self.PARK_X = self.KS_T0
        self.PARK_G = self.KS_T0 + 1
        self.PARK_FRAGS = self.KS_T0 + 1 + ((self.GKT + self.KS_C) if guide else 0)


def vq(b):
    return "v[%d:%d]" % (b, b + 3)


def aq(b):
    return "a[%d:%d]" % (b, b + 3)


def qslot(n):
    """Register quad of weight-queue slot n % QSLOTS."""
    k = n % QSLOTS
    assert 4 <= QSLOTS <= 10
    return vq(QBASE + 4 * k) if k < QV else aq(240 + 4 * (k - QV))


class Op:
    """One epilogue instruction: text, the ops it depends on (index in its list, minimum distance)."""
    def __init__(self, text, deps=(), trans=False):
        self.text, self.deps, self.trans = text, list(deps), trans


def order_ops(ops):
    """Greedy in-order list scheduling: keep the given priority order, delay an op until each dependence is
    at least `mind` instructions old (pads with s_nop only if nothing else can go)."""
    out, pos, rem = [], {}, list(range(len(ops)))
    while rem:
        for i in rem:
            if all(d in pos and len(out) - pos[d] >= md for d, md in ops[i].deps):
                pos[i] = len(out)
                out.append(ops[i])
                rem.remove(i)
                break
        else:
            out.append(Op("s_nop 0"))
    return out


def silu_epilogue(pa, res_kind, res, out_kind, out, nt_tmp=4):
    """Epilogue of one 32x32 tile whose pre-activation u = KD * z sits in VGPRs pa[0:16).
    value[e] = u * rcp(1 + exp2(u)) (+ residual):
      res_kind 'none'            -> value in pa[e]
               'acc'  (res = base of the second accumulator, VGPR)  -> v_fmac into res[e]
               'frag' (res = two AGPR fragment bases holding the bf16 identity residual) -> value in pa[e]
    out_kind 'v' (out = VGPR base of 8 packed registers) | 'a' (AGPR base of fragment 0, fragment 1)."""
    ops = []
    T = [TMP + i for i in range(nt_tmp)]
    PT = [TMP + 4 + i for i in range(4)]   # two (packed, low) pairs for the identity residual
    idx = {}

    def add(key, text, deps=(), trans=False):
        idx[key] = len(ops)
        ops.append(Op(text, [(idx[k], md) for k, md in deps], trans))

    def stage(e, s):
        t = T[e % nt_tmp]
        if s == 0:
            deps = []
            if e >= nt_tmp:
                deps.append((("D", e - nt_tmp), 1))   # temporary reuse
            add(("A", e), "v_exp_f32_e32 v%d, v%d" % (t, pa + e), deps, True)
        elif s == 1:
            add(("B", e), "v_add_f32_e32 v%d, 1.0, v%d" % (t, t), [(("A", e), 3)])
        elif s == 2:
            add(("C", e), "v_rcp_f32_e32 v%d, v%d" % (t, t), [(("B", e), 2)], True)
        else:
            if res_kind == "none":
                add(("D", e), "v_mul_f32_e32 v%d, v%d, v%d" % (pa + e, pa + e, t), [(("C", e), 3)])
            elif res_kind == "acc":
                add(("D", e), "v_fmac_f32_e32 v%d, v%d, v%d" % (res + e, pa + e, t), [(("C", e), 3)])
            else:
                j = e // 2
                r = PT[2 * (j % 2) + 1] if e % 2 == 0 else PT[2 * (j % 2)]
                add(("D", e), "v_fma_f32 v%d, v%d, v%d, v%d" % (pa + e, pa + e, t, r),
                    [(("C", e), 3), (("R1", j) if e % 2 == 0 else ("R2", j), 2)])

    def residual(j):   # packed bf16 pair j (elements 2j, 2j+1) of the identity residual -> two fp32 temporaries
        frag, reg = res[j // 4], j % 4
        p, lo = PT[2 * (j % 2)], PT[2 * (j % 2) + 1]
        deps = [(("D", 2 * (j - 2)), 1), (("D", 2 * (j - 2) + 1), 1)] if j >= 2 else []
        add(("R0", j), "v_accvgpr_read_b32 v%d, a%d" % (p, frag + reg), deps)
        add(("R1", j), "v_lshlrev_b32_e32 v%d, 16, v%d" % (lo, p), [(("R0", j), 2)])
        add(("R2", j), "v_and_b32_e32 v%d, 0xffff0000, v%d" % (p, p), [(("R1", j), 1)])

    def pack(j):
        val = res if res_kind == "acc" else pa
        d0, d1 = ("D", 2 * j), ("D", 2 * j + 1)
        if out_kind == "v":
            add(("P", j), "v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (out + j, val + 2 * j, val + 2 * j + 1), [(d0, 2), (d1, 2)])
        else:
            add(("P", j), "v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (val + 2 * j, val + 2 * j, val + 2 * j + 1), [(d0, 2), (d1, 2)])
            add(("W", j), "v_accvgpr_write_b32 a%d, v%d" % (out[j // 4] + j % 4, val + 2 * j), [(("P", j), 2)])

    # priority order: element pipeline skewed by one stage per slot (A(e) in slot e + sh, D(e) in slot e + sh + 3);
    # the residual pair j is unpacked at the end of slot 2j + 2 (after the last user of its temporaries, two slots
    # before its first use); pack(j) one slot after D(2j + 1)
    sh = 1 if res_kind == "frag" else 0
    for k in range(16 + 3 + 4 + sh):
        for s in range(4):
            e = k - s - sh
            if 0 <= e < 16:
                stage(e, s)
        if res_kind == "frag" and k >= 2 and k % 2 == 0 and (k - 2) // 2 < 8:
            residual((k - 2) // 2)
        jp = k - 5 - sh
        if jp >= 0 and jp % 2 == 1 and jp // 2 < 8:
            pack(jp // 2)
    return order_ops(ops)


KD_INV_BITS = 0xbf317218   # -ln 2 = 1 / KD
VK = TMP + 7               # guidance epilogues keep 1 / KD here (the main net's 'frag' epilogues use it as a temporary)


def _pipeline(stages, n_el, post=None, every=1):
    """Software-pipelined element loop: stage s of element e is issued in slot e + s; post(j) (pair j = elements 2j,
    2j + 1) one slot after the last stage of element 2j + 1."""
    ns = len(stages)
    for k in range(n_el + ns + 2):
        for s_i in range(ns):
            e = k - s_i
            if 0 <= e < n_el:
                stages[s_i](e)
        if post is not None:
            jp = k - ns - 1
            if jp >= 0 and jp % 2 == 1 and jp // 2 < n_el // 2:
                post(jp // 2)


def guide_epilogue_1(pa, hv, out_h, out_d):
    """Energy net, layer 1: u = KD z1 in pa[0:16).  sg = 1 / (1 + 2^u); h' = u sg -> hv[e] (second accumulator);
    silu'(z1) = sg (1 + (u - h') / KD) -> pa[e]; both packed to bf16: h' -> VGPRs out_h[0:8), silu' -> out_d[0:8)."""
    ops, idx = [], {}
    T = [TMP + i for i in range(7)]   # sg lives from stage A to stage G: 7 elements in flight

    def add(key, text, deps=(), trans=False):
        idx[key] = len(ops)
        ops.append(Op(text, [(idx[k], md) for k, md in deps], trans))
    add("K", "v_mov_b32_e32 v%d, 0x%x" % (VK, KD_INV_BITS))
    st = [
        lambda e: add(("A", e), "v_exp_f32_e32 v%d, v%d" % (T[e % 7], pa + e), [(("G", e - 7), 1)] if e >= 7 else [], True),
        lambda e: add(("B", e), "v_add_f32_e32 v%d, 1.0, v%d" % (T[e % 7], T[e % 7]), [(("A", e), 3)]),
        lambda e: add(("C", e), "v_rcp_f32_e32 v%d, v%d" % (T[e % 7], T[e % 7]), [(("B", e), 2)], True),
        lambda e: add(("D", e), "v_mul_f32_e32 v%d, v%d, v%d" % (hv + e, pa + e, T[e % 7]), [(("C", e), 3)]),
        lambda e: add(("E", e), "v_sub_f32_e32 v%d, v%d, v%d" % (pa + e, pa + e, hv + e), [(("D", e), 2)]),
        lambda e: add(("F", e), "v_fmaak_f32 v%d, v%d, v%d, 0x3f800000" % (pa + e, pa + e, VK), [(("E", e), 2), ("K", 2)]),
        lambda e: add(("G", e), "v_mul_f32_e32 v%d, v%d, v%d" % (pa + e, pa + e, T[e % 7]), [(("F", e), 2)]),
    ]

    def pack(j):
        add(("PH", j), "v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (out_h + j, hv + 2 * j, hv + 2 * j + 1), [(("D", 2 * j), 2), (("D", 2 * j + 1), 2)])
        add(("PD", j), "v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (out_d + j, pa + 2 * j, pa + 2 * j + 1), [(("G", 2 * j), 2), (("G", 2 * j + 1), 2)])
    _pipeline(st, 16, pack)
    return order_ops(ops)


def guide_epilogue_2(pa, w3, out):
    """Energy net, layer 2 and the output weight: u = KD z2 in pa[0:16), w3 tile in w3[0:16).
    g2 = w3 silu'(z2) = w3 (sg + (u / KD) sg (1 - sg)) -> packed bf16 into AGPR fragments out = (frag0, frag1)."""
    ops, idx = [], {}
    T = [TMP + i for i in range(7)]   # sg lives from stage A to stage G: 7 elements in flight

    def add(key, text, deps=(), trans=False):
        idx[key] = len(ops)
        ops.append(Op(text, [(idx[k], md) for k, md in deps], trans))
    add("K", "v_mov_b32_e32 v%d, 0x%x" % (VK, KD_INV_BITS))
    st = [
        lambda e: add(("A", e), "v_exp_f32_e32 v%d, v%d" % (T[e % 7], pa + e), [(("G", e - 7), 1)] if e >= 7 else [], True),
        lambda e: add(("B", e), "v_add_f32_e32 v%d, 1.0, v%d" % (T[e % 7], T[e % 7]), [(("A", e), 3)]),
        lambda e: add(("C", e), "v_rcp_f32_e32 v%d, v%d" % (T[e % 7], T[e % 7]), [(("B", e), 2)], True),
        lambda e: add(("D", e), "v_mul_f32_e32 v%d, v%d, v%d" % (pa + e, pa + e, VK), [("K", 2)]),
        lambda e: add(("E", e), "v_mul_f32_e32 v%d, v%d, v%d" % (pa + e, pa + e, T[e % 7]), [(("D", e), 2), (("C", e), 3)]),
        lambda e: add(("F", e), "v_fma_f32 v%d, -v%d, v%d, v%d" % (pa + e, pa + e, T[e % 7], pa + e), [(("E", e), 2)]),
        lambda e: add(("G", e), "v_add_f32_e32 v%d, v%d, v%d" % (pa + e, pa + e, T[e % 7]), [(("F", e), 2)]),
        lambda e: add(("H", e), "v_mul_f32_e32 v%d, v%d, v%d" % (pa + e, pa + e, w3 + e), [(("G", e), 2)]),
    ]

    def pack(j):
        add(("P", j), "v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (pa + 2 * j, pa + 2 * j, pa + 2 * j + 1), [(("H", 2 * j), 2), (("H", 2 * j + 1), 2)])
        add(("W", j), "v_accvgpr_write_b32 a%d, v%d" % (out[j // 4] + j % 4, pa + 2 * j), [(("P", j), 2)])
    _pipeline(st, 16, pack)
    return order_ops(ops)


def guide_epilogue_3(pa, d1, out):
    """g1 = (W2^T g2) * silu'(z1): pa[0:16) times the packed bf16 pairs in VGPRs d1[0:8) -> AGPR fragments out."""
    ops, idx = [], {}
    T = [TMP + i for i in range(4)]

    def add(key, text, deps=(), trans=False):
        idx[key] = len(ops)
        ops.append(Op(text, [(idx[k], md) for k, md in deps], trans))
    for j in range(8):
        lo, hi = T[2 * (j % 2)], T[2 * (j % 2) + 1]
        prev = [(("M1", j - 2), 1), (("M0", j - 2), 1)] if j >= 2 else []
        add(("L", j), "v_lshlrev_b32_e32 v%d, 16, v%d" % (lo, d1 + j), prev)
        add(("U", j), "v_and_b32_e32 v%d, 0xffff0000, v%d" % (hi, d1 + j), prev)
        add(("M0", j), "v_mul_f32_e32 v%d, v%d, v%d" % (pa + 2 * j, pa + 2 * j, lo), [(("L", j), 2)])
        add(("M1", j), "v_mul_f32_e32 v%d, v%d, v%d" % (pa + 2 * j + 1, pa + 2 * j + 1, hi), [(("U", j), 2)])
        add(("P", j), "v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (pa + 2 * j, pa + 2 * j, pa + 2 * j + 1), [(("M0", j), 2), (("M1", j), 2)])
        add(("W", j), "v_accvgpr_write_b32 a%d, v%d" % (out[j // 4] + j % 4, pa + 2 * j), [(("P", j), 2)])
    return order_ops(ops)


class Body:
    def __init__(self, cfg, rot=None):
        self.c = cfg
        self.rot = rot      # wave id this body is for (ROT), or None: one body for all waves
        self.mf = []        # MFMA list: dict(D, B, chain-first flag, bias seqs to wait for, tile)
        self.tiles = []     # dict(first, last, buf, ops (epilogue), outputs (register names), kind)
        self.out = []
        self.frag_reader = {}

    # ------------------------------------------------------------------ program construction
    def chain(self, acc, bfrags, tile=None, bias=None, zero=False, wait=None):
        """A run of MFMAs accumulating into `acc` (string of 16 registers), one per B fragment.  zero: the first one
        starts from 0 instead of a preloaded bias tile; wait: name of an LDS read group its first MFMA depends on."""
        for i, b in enumerate(bfrags):
            self.mf.append(dict(D=acc, B=b, tile=tile, bias=bias if i == 0 else None, zero=zero and i == 0,
                                wait=wait if i == 0 else None))

    def new_tile(self, kind):
        t = dict(idx=len(self.tiles), buf=ACC[len(self.tiles) % 2], first=len(self.mf), kind=kind, ops=[], outs=[],
                 bias=[])
        self.tiles.append(t)
        return t

    def end_tile(self, t, ops, outs):
        t["last"] = len(self.mf) - 1
        t["ops"], t["outs"] = ops, outs

    def acc_v(self, base):
        return "v[%d:%d]" % (base, base + 15)

    def block_h(self, b1, bt, in_frags, ste, nt):
        """h1[n] = silu(W1 in + b1) + (Wt ste + bt); stream order per tile: time-MLP fragments, then dense1."""
        for n in range(nt):
            t = self.new_tile("h")
            A, Tt = t["buf"], t["buf"] + 16
            if FOLD:
                t["bias"] = [(A, b1 + 32 * n)]
                self.chain(self.acc_v(Tt), ste, t, zero=True)
            else:
                t["bias"] = [(Tt, bt + 32 * n), (A, b1 + 32 * n)]
                self.chain(self.acc_v(Tt), ste, t, bias=Tt)
            self.chain(self.acc_v(A), in_frags, t, bias=A)
            out = HB + 8 * n
            self.end_tile(t, silu_epilogue(A, "acc", Tt, "v", out), [vq(out), vq(out + 4)])

    def block_o(self, b2, bm, h1, in_frags, nt, modify, out_frags):
        """out[n] = silu(W2 h1 + b2) + (Wm in + bm | in[n]); stream order per tile: modify_x, then dense2."""
        for n in range(nt):
            t = self.new_tile("o")
            Bq, M = t["buf"], t["buf"] + 16
            o = (out_frags[2 * n], out_frags[2 * n + 1])
            if modify:
                t["bias"] = [(M, bm + 32 * n), (Bq, b2 + 32 * n)]
                self.chain(self.acc_v(M), in_frags, t, bias=M)
                self.chain(self.acc_v(Bq), h1, t, bias=Bq)
                ops = silu_epilogue(Bq, "acc", M, "a", o)
            else:
                t["bias"] = [(Bq, b2 + 32 * n)]
                self.chain(self.acc_v(Bq), h1, t, bias=Bq)
                ops = silu_epilogue(Bq, "frag", (in_frags_base(in_frags[2 * n]), in_frags_base(in_frags[2 * n + 1])), "a", o)
            self.end_tile(t, ops, [aq(o[0]), aq(o[1])])

    def build(self):
        c = self.c
        hb = [vq(HB + 4 * i) for i in range(32)]
        ste = [aq(A_STE + 4 * i) for i in range(c.KS_T)]
        r1 = [A_R1 + 4 * i for i in range(32)]
        r2 = [A_R2 + 4 * i for i in range(16)]
        # [t_emb | c_emb] and x fragments: written by the caller to this lane's park slots in LDS, read at the top of
        # the body into the d1 registers (free until down0 writes d1, long after the first block)
        assert c.KS_T0 + 1 <= c.K1
        tc = [aq(A_R2 + 4 * i) for i in range(c.KS_T0)]
        xb = [aq(A_R2 + 4 * c.KS_T0)]
        # ---- sort_t.0 -> z (hb[0:KS_T)), sort_t.2 -> ste
        for n in range(c.NT_T):
            t = self.new_tile("s")
            A = t["buf"]
            t["bias"] = [(A, c.B_SORT0 + 32 * n)]
            self.chain(self.acc_v(A), tc, t, bias=A)
            self.end_tile(t, silu_epilogue(A, "none", None, "v", HB + 8 * n), [hb[2 * n], hb[2 * n + 1]])
        for n in range(c.NT_T):
            t = self.new_tile("s")
            A = t["buf"]
            t["bias"] = [(A, c.B_SORT2 + 32 * n)]
            self.chain(self.acc_v(A), hb[:c.KS_T], t, bias=A)
            o = (A_STE + 8 * n, A_STE + 8 * n + 4)
            self.end_tile(t, silu_epilogue(A, "none", None, "a", o), [aq(o[0]), aq(o[1])])
        accL = "a[%d:%d]" % (A_ACCL, A_ACCL + 15)
        H0, H1, H2 = c.H0, c.H1, c.H2
        d0 = r1[:c.K0]
        d1 = r2[:c.K1]
        d2 = r1[c.K1:c.K1 + c.K2]                 # a[A_R1 + 64 ...)
        u = r1[c.K1 + c.K2:c.K1 + 2 * c.K2]
        up0 = r1[:c.K1]
        A = lambda l: [aq(b) for b in l]  # noqa: E731
        # ---- first block, d0 half of last_block
        self.block_h(c.B_FIRST, c.B_FIRST + H0, xb, ste, c.N0)
        self.block_o(c.B_FIRST + 2 * H0, c.B_FIRST + 3 * H0, hb[:c.K0], xb, c.N0, True, d0)
        self.chain(accL, A(d0))
        # ---- down blocks
        self.block_h(c.B_DOWN0, c.B_DOWN0 + H1, A(d0), ste, c.N1)
        self.block_o(c.B_DOWN0 + 2 * H1, c.B_DOWN0 + 3 * H1, hb[:c.K1], A(d0), c.N1, True, d1)
        self.block_h(c.B_DOWN1, c.B_DOWN1 + H2, A(d1), ste, c.N2)
        self.block_o(c.B_DOWN1 + 2 * H2, c.B_DOWN1 + 3 * H2, hb[:c.K2], A(d1), c.N2, True, d2)
        # ---- middle block (identity residual)
        self.block_h(c.B_MID, c.B_MID + H2, A(d2), ste, c.N2)
        self.block_o(c.B_MID + 2 * H2, 0, hb[:c.K2], A(d2), c.N2, False, u)
        # ---- up blocks over the concatenations (identity residual, in place)
        cat0 = u + d2
        self.block_h(c.B_UP0, c.B_UP0 + H1, A(cat0), ste, c.N1)
        self.block_o(c.B_UP0 + 2 * H1, 0, hb[:c.K1], A(cat0), c.N1, False, up0)
        cat1 = up0 + d1
        self.block_h(c.B_UP1, c.B_UP1 + H0, A(cat1), ste, c.N0)
        self.block_o(c.B_UP1 + 2 * H0, 0, hb[:c.K0], A(cat1), c.N0, False, cat1)
        self.chain(accL, A(cat1))
        self.guide_first = len(self.mf)
        if c.guide:
            self.build_guide()
        self.nfrags = len(self.mf)

    def build_guide(self):
        """dE/dx of the energy net at (t, x, c) (engine_bf16.hip.inc, guide_grad): every register of the main net is
        free by now except the last_block accumulator.
          in1  = [t_emb_g | x | c] fragments   a[0:4*KS1)      (read from the park slots; ste is dead)
          h1'  = KD silu(z1)                   v[HB ...)        16 fragments
          d1   = silu'(z1) (bf16)              v[HB + 64 ...)   16 fragments
          g2   = w3 silu'(z2)                  a[A_R1 ...)      16 fragments
          g1   = (W2^T g2) d1                  a[A_R1 + 64 ...) 16 fragments
          dE/dx accumulator                    a[A_R2 : A_R2 + 16)"""
        c = self.c
        in1 = [aq(A_STE + 4 * i) for i in range(c.GKS1)]
        H1V, D1V = HB, HB + 4 * c.GK1
        h1 = [vq(H1V + 4 * i) for i in range(c.GK1)]
        G2A, G1A = A_R1, A_R1 + 4 * c.GK2
        g2 = [aq(G2A + 4 * i) for i in range(c.GK2)]
        g1 = [aq(G1A + 4 * i) for i in range(c.GK1)]
        for n in range(c.GN1):
            t = self.new_tile("g1")
            A, Hv = t["buf"], t["buf"] + 16
            t["bias"] = [(A, c.GB_1 + 32 * n)]
            self.chain(self.acc_v(A), in1, t, bias=A, wait="in1" if n == 0 else None)
            self.end_tile(t, guide_epilogue_1(A, Hv, H1V + 8 * n, D1V + 8 * n),
                          [h1[2 * n], h1[2 * n + 1], vq(D1V + 8 * n), vq(D1V + 8 * n + 4)])
        for n in range(c.GN2):
            t = self.new_tile("g2")
            A, W3 = t["buf"], t["buf"] + 16
            t["bias"] = [(W3, c.GB_W3 + 32 * n), (A, c.GB_2 + 32 * n)]
            self.chain(self.acc_v(A), h1, t, bias=A)
            o = (G2A + 8 * n, G2A + 8 * n + 4)
            self.end_tile(t, guide_epilogue_2(A, W3, o), [aq(o[0]), aq(o[1])])
        for n in range(c.GN1):
            t = self.new_tile("g3")
            A = t["buf"]
            self.chain(self.acc_v(A), g2, t, zero=True)
            o = (G1A + 8 * n, G1A + 8 * n + 4)
            self.end_tile(t, guide_epilogue_3(A, D1V + 8 * n, o), [aq(o[0]), aq(o[1])])
        self.chain("a[%d:%d]" % (A_R2, A_R2 + 15), g1, zero=True)

    # ------------------------------------------------------------------ emission
    def emit(self, s):
        self.out.append(s)

    def lds_issue(self):
        self.lds_seq += 1
        return self.lds_seq

    def lds_wait(self, seq):
        if seq > self.lds_done:
            n = self.lds_seq - seq
            assert n >= 0
            # the counter has 4 bits: the hardware never lets more than 15 LGKM operations be outstanding, so a
            # target with >= 15 younger operations is complete by construction
            if "nowait" not in X and n < 15:
                self.emit("s_waitcnt lgkmcnt(%d)" % n)
            self.lds_done = seq

    def advance(self, chunk):
        """Ring hand-off at the first read of chunk `chunk`: own pieces landed, barrier, refill the slot freed two
        chunks ago with chunk + NSLOT - 2 (engine_bf16.hip.inc, Ring)."""
        c = self.c
        loads = c.cf // 4
        src = (chunk + c.nslot - 2) % self.chunks
        slot = (chunk + c.nslot - 2) % c.nslot
        if "nowaitv" not in X:
            self.emit("s_waitcnt vmcnt(%d)" % ((c.nslot - 3) * loads))
        if "nobarrier" not in X:
            self.emit("s_barrier")
        if SKEW:
            # stagger the waves: they share ONE vector-memory path per CU and every LDS-DMA piece holds its wave for
            # the ~16 cycles the path needs for 1 KiB; issued by four waves at the same moment the pieces queue up.
            self.lbl = getattr(self, "lbl", 0) + 1
            for w in (1, 2, 3):
                self.emit("s_cmp_lt_u32 %%[wid], %d" % w)
                self.emit("s_cbranch_scc1 .Lskew%d_%%=" % self.lbl)
                self.emit("s_nop %d" % (SKEW - 1))
            self.emit(".Lskew%d_%%=:" % self.lbl)
        if "nodma" in X:
            return
        self.emit("s_mov_b32 %%[soff], 0x%x" % (src * c.cf * 1024))
        self.emit("s_add_u32 m0, %%[ldsw], 0x%x" % (slot * c.cf * 1024))
        if SPREAD:
            # m0 / soff stay valid until the next hand-off; the pieces are drained by dma_tick()
            assert not getattr(self, "dma_pending", []), "previous chunk's pieces not issued yet"
            self.dma_pending = [p for p in range(loads)]
            self.dma_count = 0
            if self.rot is not None:
                # one piece per MFMA gap and CU: this wave's are the gaps = rot (mod 4) counted from the hand-off
                g0 = self.cur_gap + 1
                self.dma_gaps = set(g0 + self.rot + (c.cf // loads) * i for i in range(loads))
            elif SPREAD < 0:
                # slack-aware: the `loads` cheapest of the next gaps (by the issue cost of what is already placed
                # there), leaving the last two gaps before the next hand-off alone
                g0 = self.cur_gap
                cand = [g for g in range(g0, min(g0 + c.cf - 2, self.nfrags))]
                cand.sort(key=lambda g: (self.gapcost.get(g, 0), g))
                self.dma_gaps = set(cand[:loads])
            return
        self.emit("s_nop 0")
        for p in range(loads):
            self.piece(p)

    def piece(self, p):
        """One 1 KiB LDS-DMA piece of the current chunk.  The immediate offset has 12 bits: every fourth piece steps
        the source offset and the LDS destination (m0) by 4 KiB."""
        if p and p % 4 == 0:
            self.emit("s_add_u32 %[soff], %[soff], 0x1000")
            self.emit("s_add_u32 m0, m0, 0x1000")
            self.emit("s_nop 0")
        off = " offset:%d" % ((p % 4) * 1024) if p % 4 else ""
        self.emit("buffer_load_dwordx4 %%[voff], %%[rsrc], %%[soff] offen%s%s lds" % (off, (" " + DMAPOL) if DMAPOL else ""))

    def dma_tick(self, flush=False):
        if not SPREAD:
            return
        pend = getattr(self, "dma_pending", [])
        self.dma_count = getattr(self, "dma_count", 0) + 1
        if SPREAD < 0:
            while pend and (flush or self.cur_gap in getattr(self, "dma_gaps", ())):
                self.piece(pend.pop(0))
                if not flush:
                    break
            return
        while pend and (flush or self.dma_count % SPREAD == 0):
            self.piece(pend.pop(0))
            if not flush:
                break

    def frag_read(self, n):
        """ds_read of weight fragment n into its queue slot (+ the ring hand-off when it opens a chunk)."""
        c = self.c
        if n % c.cf == 0:
            self.advance(n // c.cf)
        pos = n % (c.cf * c.nslot)
        if "nofrag" in X:
            self.frag_seq[n] = self.lds_seq
            return
        assert pos < 64, "the ring must fit the 16-bit immediate offset of ds_read"
        self.emit("ds_read_b128 %s, %%[rd] offset:%d" % (qslot(n), pos * 1024))
        self.frag_seq[n] = self.lds_issue()

    def generate(self):
        c = self.c
        self.build()
        nf = self.nfrags
        used = (nf + c.cf - 1) // c.cf
        self.chunks = (used + c.nslot - 1) // c.nslot * c.nslot
        self.lds_seq = self.lds_done = 0
        self.frag_seq = {}
        PF = QSLOTS
        def consumer(name, after):   # first MFMA reading an output fragment AFTER it is produced
            for g in range(after + 1, nf):
                if self.mf[g]["B"] == name:
                    return g
            return nf + 8
        # Epilogue window of tile t: gaps [lo, hi).  lo: two MFMAs after the tile's last one (its results are final
        # and past the MFMA -> VALU read hazard).  hi: before the first MFMA that consumes an output fragment, and
        # before the bias tiles of tile t+2 are read into the accumulator buffer this epilogue still reads.
        nt = len(self.tiles)
        for t in self.tiles:
            ln = t["last"] - t["first"] + 1
            t["lead"] = 5 if ln >= 16 else 3
        self.tiles[0]["bias_gap"] = -1
        if nt > 1:
            self.tiles[1]["bias_gap"] = max(self.tiles[1]["first"] - self.tiles[1]["lead"], -1)
        valu_at, bias_at, forced = {}, {}, {}
        for t in self.tiles:
            lo = t["last"] + DBG_LO
            cidx = min(consumer(o, t["last"]) for o in t["outs"])
            serial = cidx - 1 <= lo      # the next MFMAs consume this tile's output at once: no gaps to hide in
            hi = cidx if serial else cidx - 1
            if t["idx"] + 1 < nt and not serial:
                # windows must not overlap: consecutive epilogues share the temporaries
                hi = min(hi, max(self.tiles[t["idx"] + 1]["last"] + DBG_LO, lo + 1))
            if t["idx"] + 2 < nt:
                t2 = self.tiles[t["idx"] + 2]
                cap = t2["first"] - 1
                assert cap >= hi if serial else cap > lo, (t["idx"], lo, hi, cap)
                if not serial:
                    hi = min(hi, cap, max(t2["first"] - t2["lead"], lo + 1))
                t2["bias_gap"] = max(t2["first"] - t2["lead"], hi)
                assert hi <= t2["bias_gap"] <= t2["first"] - 1, (t["idx"], hi, t2["bias_gap"], t2["first"])
            ops = t["ops"]
            if serial:
                # results of the tile's last MFMA must be final before the first VALU read (MFMA -> VALU hazard is
                # not interlocked): idle for two MFMA lengths, run the whole epilogue, then the consumer
                forced[cidx] = [Op("s_nop 15"), Op("s_nop 15")] + ops + [Op("s_nop 1")]
            else:
                if DBG_NOP:
                    ops = [Op("s_nop 15")] * DBG_NOP + ops
                for i, op in enumerate(ops):
                    valu_at.setdefault(lo + i * (hi - lo) // len(ops), []).append(op)
            t["window"] = (lo, hi)
        for t in self.tiles:
            bias_at.setdefault(t["bias_gap"], []).extend(t["bias"])
        # issue cost (cycles) of what each gap already holds: VALU, bias reads; used to place the DMA pieces
        self.gapcost = {}
        for g, ops in valu_at.items():
            self.gapcost[g] = sum(8 if op.trans else 4 for op in ops)
        for g, bl in bias_at.items():
            self.gapcost[g] = self.gapcost.get(g, 0) + 16 * len(bl)
        self.cur_gap = -1
        # ---- prologue: parked inputs, last_block bias into its accumulator, first biases, queue priming
        for i in range(c.KS_T0 + 1):
            self.emit("ds_read_b128 %s, %%[park] offset:%d" % (aq(A_R2 + 4 * i), i * 1024))
            self.lds_issue()
        for q in range(4):
            self.emit("ds_read_b128 %s, %%[bias] offset:%d" % (aq(A_ACCL + 4 * q), (c.B_LAST + 8 * q) * 4))
            accl_seq = self.lds_issue()
        bias_seq = {}

        bias_queue = []

        def bias_reads(gap):
            for dst, off in bias_at.get(gap, []):
                for q in range(4):
                    bias_queue.append((gap, dst, q, off))
            n = len(bias_queue) if (not BSPREAD or gap < 0) else min(BSPREAD, len(bias_queue))
            for _ in range(n):
                bgap, dst, q, off = bias_queue.pop(0)
                self.emit("ds_read_b128 %s, %%[bias] offset:%d" % (vq(dst + 4 * q), (off + 8 * q) * 4))
                bias_seq[(bgap, dst)] = self.lds_issue()
        bias_reads(-1)
        for n in range(min(PF, nf)):
            self.frag_read(n)
        group_seq = {}
        in1_gap = -1
        if c.guide:
            # the energy net's inputs replace the (dead) SiLU(t-embedding) fragments a dozen gaps before their first use
            in1_gap = self.guide_first - 12
            ste_names = set(aq(A_STE + 4 * i) for i in range(max(c.KS_T, c.GKS1)))
            assert all(m["B"] not in ste_names for m in self.mf[in1_gap:self.guide_first]), "ste still in use"
        for g, m in enumerate(self.mf):
            self.cur_gap = g
            for op in forced.get(g, []):
                self.emit(op.text)
            # operands ready?
            need = self.frag_seq[g]
            if g % WAITB == 0:                                   # one wait per WAITB fragments
                for k in range(1, WAITB):
                    if (g + k) in self.frag_seq:
                        need = max(need, self.frag_seq[g + k])
            if m["bias"] is not None:
                key = (m["tile"]["bias_gap"], m["bias"])
                while any(b[0] == key[0] and b[1] == key[1] for b in bias_queue):   # still queued: issue them now
                    bgap, dst, q, off = bias_queue.pop(0)
                    self.emit("ds_read_b128 %s, %%[bias] offset:%d" % (vq(dst + 4 * q), (off + 8 * q) * 4))
                    bias_seq[(bgap, dst)] = self.lds_issue()
                need = max(need, bias_seq[key])
            if m["D"].startswith("a[") and not getattr(self, "_accl_ready", False):
                need = max(need, accl_seq)
                self._accl_ready = True
            if m.get("wait"):
                need = max(need, group_seq[m["wait"]])
            self.lds_wait(need)
            self.emit("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (m["D"], qslot(g), m["B"], "0" if m.get("zero") else m["D"]))
            if g == in1_gap:
                park = [c.PARK_G + i for i in range(c.GKT)] + [c.PARK_X] + [c.PARK_G + c.GKT + i for i in range(c.KS_C)]
                for i, slot in enumerate(park):
                    self.emit("ds_read_b128 %s, %%[park] offset:%d" % (aq(A_STE + 4 * i), slot * 1024))
                    group_seq["in1"] = self.lds_issue()
            if DPOS == 2 and not (g + PF < nf and (g + PF) % c.cf == 0):
                self.dma_tick()
            if g + PF < nf:
                if (g + PF) % c.cf == 0:
                    self.dma_tick(flush=True)     # pieces of the previous chunk must be issued before the next hand-off
                self.frag_read(g + PF)
            if DPOS == 0 or (DPOS == 2 and g + PF < nf and (g + PF) % c.cf == 0):
                self.dma_tick()
            bias_reads(g)
            for op in valu_at.get(g, []):
                if "novalu" not in X:
                    self.emit(op.text)
            if DPOS == 1:
                self.dma_tick()
        # epilogue ops scheduled past the last MFMA (none expected) and the padding chunks of the ring
        for g in sorted(k for k in valu_at if k >= nf):
            for op in valu_at[g]:
                self.emit(op.text)
        for ch in range(used, self.chunks):
            self.dma_tick(flush=True)
            self.advance(ch)
        self.dma_tick(flush=True)
        self.emit("s_waitcnt lgkmcnt(0)")
        self.emit("s_nop 15")
        self.emit("s_nop 7")
        for i in range(8):
            self.emit("v_accvgpr_read_b32 %%[o%d], a%d" % (i, A_ACCL + i))
        if c.guide:
            for i in range(8):
                self.emit("v_accvgpr_read_b32 %%[o%d], a%d" % (8 + i, A_R2 + i))
        return self.out


def in_frags_base(name):
    assert name.startswith("a["), name
    return int(name[2:name.index(":")])


def function_text(cfg):
    bodies = [Body(cfg, r) for r in range(4)] if ROT else [Body(cfg)]
    all_lines = [b.generate() for b in bodies]
    b, lines = bodies[0], all_lines[0]
    cnt = {}
    for l in lines:
        k = l.split()[0]
        cnt[k] = cnt.get(k, 0) + 1
    ntc = cfg.KS_T0
    o = []
    o.append("// ---- %s: %d fragments per step, %d instructions (%s)" % (
        cfg.name, b.nfrags, len(lines), ", ".join("%s %d" % kv for kv in sorted(cnt.items(), key=lambda kv: -kv[1])[:12])))
    o.append("struct AsmBody_%s {" % cfg.name)
    o.append("  static constexpr bool ON = true;")
    o.append("  static constexpr int FRAGS = %d, NSLOT = %d, CF = %d, KS_T0 = %d, PARK_FRAGS = %d, PARK_X = %d, PARK_G = %d, "
             "OUTPUTS = %d, INSTRUCTIONS = %d;" % (b.nfrags, cfg.nslot, cfg.cf, ntc, cfg.PARK_FRAGS, cfg.PARK_X, cfg.PARK_G,
                                                  16 if cfg.guide else 8, len(lines)))
    o.append("  // park: LDS address of this lane's PARK_FRAGS input fragments, 1 KiB apart: [t_emb | c_emb], x at PARK_X, and for")
    o.append("  // guided bodies from PARK_G the energy net's t_emb fragments, then the raw condition fragments.")
    o.append("  // o[0:8) = KD * net; guided bodies: o[8:16) = dE/dx of the energy net")
    o.append("  static __device__ __forceinline__ void run(uint32_t park, uint32_t voff, uint32_t rd, uint32_t bias, "
             "i32x4 rsrc, uint32_t ldsw, uint32_t wid, float* o) {")
    o.append("    uint32_t soff;")
    outs = ", ".join('[o%d] "=v"(o[%d])' % (i, i) for i in range(16 if cfg.guide else 8)) + ', [soff] "=&s"(soff)'
    ins = '[park] "v"(park), [voff] "v"(voff), [rd] "v"(rd), [bias] "v"(bias), [rsrc] "s"(rsrc), [ldsw] "s"(ldsw), [wid] "s"(wid)'
    clob = ['"memory"', '"scc"']
    clob += ['"v%d"' % i for i in range(NV)]
    clob += ['"a%d"' % i for i in range(256)]
    for r, blines in enumerate(all_lines):
        if len(all_lines) > 1:   # one body per wave of the workgroup (wid is wave-uniform)
            o.append("    %sif (wid == %d) {" % ("" if r == 0 else "else ", r) if r < len(all_lines) - 1 else "    else {")
        o.append("    asm volatile(")
        for l in blines:
            o.append('      "%s\\n\\t"' % l)
        o.append("      : %s" % outs)
        o.append("      : %s" % ins)
        o.append("      : %s);" % ", ".join(clob))
        if len(all_lines) > 1:
            o.append("    }")
    o.append("  }")
    o.append("};")
    return "\n".join(o), cnt, len(lines), b


CONFIGS = [
    # shape classes of engine_bf16_variants.hpp: the body depends on the hidden shape only (state <= 16 features in
    # one k-step; the condition embedding is computed outside the body)
    Cfg("cond", 16, 32, 32, 32, 128, 512, 256, 128),
    Cfg("uncond", 16, 0, 0, 32, 32, 512, 256, 128),
    Cfg("cond_guided", 16, 32, 32, 32, 128, 512, 256, 128, guide=(32, 256, 256)),
]

if __name__ == "__main__":
    out = sys.argv[1]
    knobs = dict(LO=DBG_LO, NOP=DBG_NOP, X=",".join(sorted(X)), SPREAD=SPREAD, WAITB=WAITB, BSPREAD=BSPREAD, SKEW=SKEW,
                 PF=QSLOTS, NSLOT=CONFIGS[0].nslot, CF=CONFIGS[0].cf, ROT=ROT, DPOS=DPOS, DMAPOL=DMAPOL, FOLD=FOLD)
    shipped = dict(LO=2, NOP=0, X="", SPREAD=-1, WAITB=4, BSPREAD=2, SKEW=0, PF=8, NSLOT=4, CF=16, ROT=0, DPOS=0, DMAPOL="",
                   FOLD=1)
    parts = ["// GENERATED by gen_bf16_body.py -- do not edit.  See that file for the design.", "#pragma once",
             "// generator knobs: " + " ".join("%s=%s" % kv for kv in sorted(knobs.items())),
             "#define B16_GEN_DEFAULT_KNOBS %d" % int(knobs == shipped),
             "#define B16_GEN_FOLD_TIME_BIAS %d" % FOLD,
             "namespace grlhip {", "namespace b16 {", "typedef int i32x4 __attribute__((ext_vector_type(4)));"]
    for cfg in CONFIGS:
        text, cnt, n, b = function_text(cfg)
        parts.append(text)
        mf = cnt.get("v_mfma_f32_32x32x16_bf16", 1)
        trans = cnt.get("v_exp_f32_e32", 0) + cnt.get("v_rcp_f32_e32", 0)
        sys.stderr.write("%s: %d MFMA, %d instructions (%.2f per MFMA), issue bound %.1f cycles per MFMA\n" % (
            cfg.name, mf, n, n / mf, 4.0 * (n + trans + mf) / mf))
    parts += ["}  // namespace b16", "}  // namespace grlhip", ""]
    with open(out, "w") as f:
        f.write("\n".join(parts))
