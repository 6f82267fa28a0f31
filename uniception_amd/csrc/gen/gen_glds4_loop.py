#!/usr/bin/env python3
"""Generator of the hand-scheduled K-loop of the four-wave dense bf16 GEMM (gemm_bf16_glds4_kernel, gemm_glds_kernel.h).

    python uniception_amd/csrc/gen/gen_glds4_loop.py > uniception_amd/csrc/gemm_glds4_loop.inc        (committed; build.py does not run this)

256 x 256 workgroup tile, FOUR waves of 128 x 128 (one per SIMD), 256 fp32 accumulators per lane pinned in a0..a255 — the layout of
the vendor's hand-written GEMM kernels.  hipcc cannot produce it (round 1 / round 2: 604 v_accvgpr moves and 1 KB of scratch in the
loops from the intrinsic form), so the loop is emitted as ONE inline-asm statement whose registers are fixed here:

    a[4 * (8 i + j) + r]   accumulator r of MFMA fragment (i, j): i = 16-row block of the wave's 128 rows, j = 16-column block
    v[FR0 ...]             two fragment sets (current / next 32-deep K half): 8 A + 8 W fragments of 4 registers each
    s[SB ...]              8 wave-uniform 64-bit source bases of the wave's DMA pieces, loop state

K is consumed in HALF-STAGES of 32 (64-byte LDS rows, 512 rows = 32 KiB per half-stage) through a ring of FOUR buffers; one phase =
one half-stage = 64 v_mfma_f32_16x16x32_bf16 per wave:

    phase p   s_waitcnt vmcnt(16) lgkmcnt(0); s_barrier      half-stage p + 1 has landed for everyone (p + 2 and p + 3 stay in flight:
                                                             the loads are never drained inside the loop), buffer p % 4 is free
              64 MFMAs on fragment set p & 1                 | 16 ds_read_b128: fragments of half-stage p + 1 -> set (p + 1) & 1
                                                             | 8 LDS-DMA pieces of 1 KiB: half-stage p + 4 -> buffer p % 4

so a DMA piece has three phases (~1.5 us) to land before anyone waits for it.  What the two-stage form of this kernel taught (round 3,
tools/_libs experiments at 8192^3: 1254 TFLOP/s as first written, 1584 without its barrier, 1846 without its DMA): one wave per SIMD
has nobody to cover a wait, so every cycle a wave spends in `s_waitcnt vmcnt(0)` + `s_barrier` behind loads issued half a step
earlier is an idle matrix pipe; issue order and per-wave staggering of the pieces change little.

The macros UC_GLDS4_LOOP_SWAP / _NOSWAP expand to the asm text; operands (see gemm_glds_kernel.h): %0 / %1 LDS read addresses of the
wave's A / W fragment rows in buffer 0, %2 per-lane source byte offset, %3 / %4 source base (low, high dword), %5 first row of the
wave's 128-row DMA slab, %6 last valid 16-row group start, %7 row pitch in bytes, %8 LDS byte address of the wave's DMA slab in
buffer 0, %9 number of 64-deep K-steps (>= 1).
"""
import os
import sys

FR0 = 128              # first fragment VGPR: set s, operand o (0 = A, 1 = W), block b -> v[FR0 + 64 s + 32 o + 4 b .. + 3]
SB = 36                # s[SB + 2 q : SB + 2 q + 1] = source base of piece q (8 pieces)
S_LDS, S_CNT, S_T0, S_T1, S_M0 = 52, 53, 54, 55, 56
V_OFF = 120                        # running per-lane source offset
V_A, V_A2, V_W, V_W2 = 121, 122, 123, 124   # LDS read addresses: buffers 0-1 / 2-3 (+ 65536), A and W
BUF = 32768
# experiment switches (environment; the committed .inc is generated with none of them set)
NODMA = os.environ.get("G4_NODMA") == "1"          # no DMA pieces inside the loop (WRONG results)
NOBAR = os.environ.get("G4_NOBAR") == "1"          # no per-phase barrier (wrong)
DMA_AT = [int(x) for x in os.environ.get("G4_DMA_AT", "6,14,22,30,38,46,54,62").split(",")]    # piece q behind MFMA DMA_AT[q]
READ_STEP = int(os.environ.get("G4_READ_STEP", "2"))     # a fragment read behind every READ_STEP-th MFMA, from the first


def frag(s, o, b):
    r = FR0 + 64 * s + 32 * o + 4 * b
    return f"v[{r}:{r + 3}]"


def acc(i, j):
    r = 4 * (8 * i + j)
    return f"a[{r}:{r + 3}]"


class Body:
    def __init__(self):
        self.lines = []

    def emit(self, s):
        self.lines.append(s)

    def text(self):
        return "\n".join(f'    "{ln}\\n\\t"' for ln in self.lines)


def mfma(b, swap, s, i, j):
    # SWAP: first operand = W rows -> C^T fragments (a lane owns 4 consecutive columns of one row); else first operand = A rows
    a, w = frag(s, 0, i), frag(s, 1, j)
    if swap:
        b.emit(f"v_mfma_f32_16x16x32_bf16 {acc(i, j)}, {w}, {a}, {acc(i, j)}")
    else:
        b.emit(f"v_mfma_f32_16x16x32_bf16 {acc(i, j)}, {a}, {w}, {acc(i, j)}")


def ds_read(b, s, o, blk, buf):
    reg = (V_A, V_A2)[buf >> 1] if o == 0 else (V_W, V_W2)[buf >> 1]
    b.emit(f"ds_read_b128 {frag(s, o, blk)}, v{reg} offset:{(buf & 1) * BUF + blk * 1024}")


def dma_piece(b, q, buf):
    # M0 = LDS byte address of the piece (wave-uniform); hazard M0 write -> LDS-DMA: s_nop 0
    b.emit(f"s_add_u32 m0, s{S_LDS}, {buf * BUF + q * 1024}")
    b.emit("s_nop 0")
    b.emit(f"global_load_lds_dwordx4 v{V_OFF}, s[{SB + 2 * q}:{SB + 2 * q + 1}]")


def issue_half_stage(b, buf):
    for q in range(8):
        dma_piece(b, q, buf)
    b.emit(f"v_add_u32 v{V_OFF}, 64, v{V_OFF}")


def read_frags(b, s, buf):
    for j in range(8):
        ds_read(b, s, 1, j, buf)
    for i in range(8):
        ds_read(b, s, 0, i, buf)


def phase(b, swap, buf, vm, reads, dma, last=False):
    """Phase p with p % 4 == buf: synchronise, then 64 MFMAs on set buf & 1 with the fillers in between."""
    if last:
        b.emit("s_waitcnt lgkmcnt(0)")
    else:
        b.emit(f"s_waitcnt vmcnt({vm}) lgkmcnt(0)")
        if not NOBAR:
            b.emit("s_barrier")
    cur, nxt = buf & 1, (buf + 1) & 1
    nbuf = (buf + 1) & 3
    rl = ([(nxt, 1, j, nbuf) for j in range(8)] + [(nxt, 0, i, nbuf) for i in range(8)]) if reads else []
    slots = {}
    for k, x in enumerate(rl):
        slots.setdefault(READ_STEP * k + 1, []).append(("r", x))
    if dma and not NODMA:
        for q in range(8):
            slots.setdefault(DMA_AT[q], []).append(("d", q))
    n = 0
    for i in range(8):
        for j in range(8):
            mfma(b, swap, cur, i, j)
            n += 1
            for kind, x in slots.get(n, []):
                if kind == "r":
                    ds_read(b, *x)
                else:
                    dma_piece(b, x, buf)
    if dma:
        b.emit(f"v_add_u32 v{V_OFF}, 64, v{V_OFF}")


def tail(b, swap, start):
    "the last four phases: no more DMA; buffers start, start + 1, ... (mod 4)"
    phase(b, swap, start, 16, True, False)
    phase(b, swap, (start + 1) & 3, 8, True, False)
    phase(b, swap, (start + 2) & 3, 0, True, False)
    phase(b, swap, (start + 3) & 3, 0, False, False, last=True)


def loop(swap):
    b = Body()
    e = b.emit
    tag = "s" if swap else "u"
    e(f"s_mov_b32 s{S_M0}, m0")                                # M0 is compiler-reserved: saved here, restored at the end
    e(f"v_mov_b32 v{V_OFF}, %2")
    e(f"v_mov_b32 v{V_A}, %0")
    e(f"v_add_u32 v{V_A2}, 0x10000, v{V_A}")
    e(f"v_mov_b32 v{V_W}, %1")
    e(f"v_add_u32 v{V_W2}, 0x10000, v{V_W}")
    e(f"s_mov_b32 s{S_LDS}, %8")
    for q in range(8):
        e(f"s_add_u32 s{S_T0}, %5, {16 * q}")
        e(f"s_min_u32 s{S_T0}, s{S_T0}, %6")                   # row groups past the matrix end: its last 16 rows (never stored)
        e(f"s_mul_hi_u32 s{S_T1}, s{S_T0}, %7")
        e(f"s_mul_i32 s{S_T0}, s{S_T0}, %7")
        e(f"s_add_u32 s{SB + 2 * q}, %3, s{S_T0}")
        e(f"s_addc_u32 s{SB + 2 * q + 1}, %4, s{S_T1}")
    issue_half_stage(b, 0)
    issue_half_stage(b, 1)
    e(f"s_cmp_lt_u32 %9, 2")
    e(f"s_cbranch_scc1 .Lg4{tag}_small_%=")
    issue_half_stage(b, 2)
    issue_half_stage(b, 3)
    for r in range(256):
        e(f"v_accvgpr_write_b32 a{r}, 0")
    e("s_waitcnt vmcnt(24)")
    e("s_barrier")
    read_frags(b, 0, 0)
    e(f"s_lshl_b32 s{S_CNT}, %9, 1")
    e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 4")                      # phases that still issue a half-stage: 2 nk - 4
    e(f"s_cmp_lt_u32 s{S_CNT}, 4")
    e(f"s_cbranch_scc1 .Lg4{tag}_rest_%=")
    e(f".Lg4{tag}_loop_%=:")
    for buf in range(4):
        phase(b, swap, buf, 16, True, True)
    e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 4")
    e(f"s_cmp_ge_u32 s{S_CNT}, 4")
    e(f"s_cbranch_scc1 .Lg4{tag}_loop_%=")
    e(f".Lg4{tag}_rest_%=:")
    e(f"s_cmp_eq_u32 s{S_CNT}, 0")
    e(f"s_cbranch_scc1 .Lg4{tag}_tail0_%=")
    phase(b, swap, 0, 16, True, True)
    phase(b, swap, 1, 16, True, True)
    tail(b, swap, 2)
    e(f"s_branch .Lg4{tag}_end_%=")
    e(f".Lg4{tag}_tail0_%=:")
    tail(b, swap, 0)
    e(f"s_branch .Lg4{tag}_end_%=")
    e(f".Lg4{tag}_small_%=:")                                  # K = 64: two phases
    for r in range(256):
        e(f"v_accvgpr_write_b32 a{r}, 0")
    e("s_waitcnt vmcnt(8)")
    e("s_barrier")
    read_frags(b, 0, 0)
    phase(b, swap, 0, 0, True, False)
    phase(b, swap, 1, 0, False, False, last=True)
    e(f".Lg4{tag}_end_%=:")
    e("s_nop 15")                                               # MFMA results -> v_accvgpr_read of the read-out macros
    e("s_nop 15")
    e(f"s_mov_b32 m0, s{S_M0}")
    return b.text()


def clobbers():
    c = [f"a{r}" for r in range(256)] + [f"v{r}" for r in range(V_OFF, 256)]
    c += [f"s{r}" for r in range(SB, S_M0 + 1)] + ["scc", "memory"]
    return ", ".join(f'"{x}"' for x in c)


def main():
    print("// GENERATED by csrc/gen/gen_glds4_loop.py — do not edit; regenerate and commit.")
    print("// K-loop of gemm_bf16_glds4_kernel: see the generator's docstring for the register map and the schedule.")
    print("#define UC_GLDS4_LOOP_SWAP \\")
    print(loop(True).replace("\n", " \\\n"))
    print("")
    print("#define UC_GLDS4_LOOP_NOSWAP \\")
    print(loop(False).replace("\n", " \\\n"))
    print("")
    print("#define UC_GLDS4_CLOBBERS " + clobbers())
    print("")
    # accumulator read-out: acc[i][jj][r] of column half hc <- a[4 * (8 i + 4 hc + jj) + r]
    for hc in range(2):
        print(f"#define UC_GLDS4_READ_HALF{hc}(ACC) \\")
        rows = []
        for i in range(8):
            for jj in range(4):
                for r in range(4):
                    rows.append(f'    asm volatile("v_accvgpr_read_b32 %0, a{4 * (8 * i + 4 * hc + jj) + r}" : "=v"(ACC[{i}][{jj}][{r}]));')
        print(" \\\n".join(rows))
        print("")


if __name__ == "__main__":
    main()
