#!/usr/bin/env python3
"""Generator of the hand-scheduled K-loop of the four-wave dense bf16 GEMM (gemm_bf16_glds4_kernel, gemm_glds4.hip).

    python uniception_amd/csrc/gen/gen_glds4_loop.py > uniception_amd/csrc/gemm_glds4_loop.inc        (committed; build.py does not run this)

256 x 256 x 64 workgroup tile, FOUR waves of 128 x 128 (one per SIMD), 256 fp32 accumulators per lane pinned in a0..a255 — the layout
of the vendor's hand-written GEMM kernels.  hipcc cannot produce it (round 1 / round 2: 604 v_accvgpr moves and 1 KB of scratch in
the loops from the intrinsic form), so the loop is emitted as ONE inline-asm statement whose registers are fixed here:

    a[4 * (8 i + j) + r]   accumulator r of MFMA fragment (i, j): i = 16-row block of the wave's 128 rows, j = 16-column block
    v[FR0 ...]             two fragment sets (current / next 32-deep K half): 8 A + 8 W fragments of 4 registers each
    s[SB ...]              16 wave-uniform 64-bit source bases of the wave's DMA pieces, loop state

Per K-step (64 deep): 128 v_mfma_f32_16x16x32_bf16, 32 ds_read_b128 (half of the 16-wave kernel's LDS read traffic per MFMA),
16 LDS-DMA pieces of 1 KiB, TWO s_barriers (see `step`): fragment reads of K half 1 behind MFMAs 1-16, barrier (the stage's buffer is
free), the pieces of K-step kt + 2 behind every fifth MFMA 20..95, s_waitcnt vmcnt(16) + barrier (stage kt + 1 complete), fragment
reads of its K half 0 behind MFMAs 99-114.  The first form (G4_SCHED=1: one mid-step barrier behind vmcnt(0), all pieces in the second
half of the step) is kept for the record: 10 % slower at 8192^3.

The macros UC_GLDS4_LOOP_SWAP / _NOSWAP expand to the asm text; operands (see gemm_glds_kernel.h): %0..%3 LDS read addresses (A half
0, A half 1, W half 0, W half 1 of stage 0), %4 / %5 per-lane source byte offsets of even / odd pieces, %6 / %7 source base (low, high
dword), %8 first row of the wave's 128-row DMA slab, %9 last valid 8-row group start, %10 row pitch in bytes, %11 LDS byte address of
the wave's DMA slab in stage 0, %12 number of K-steps (>= 1).
"""
import os
import sys

FR0 = 128              # first fragment VGPR: set s, operand o (0 = A, 1 = W), block b -> v[FR0 + 64 s + 32 o + 4 b .. + 3]
SB = 36                # s[SB + 2 q : SB + 2 q + 1] = source base of piece q (16 pieces)
S_DMA, S_DELTA, S_CNT, S_T0, S_T1, S_M0 = 68, 69, 70, 71, 72, 73
V_OFF0, V_OFF1 = 120, 121      # running per-lane source offsets (even / odd pieces)
V_A0, V_A1, V_W0, V_W1 = 122, 123, 124, 125   # running LDS read addresses of the current stage


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


def ds_read(b, s, o, blk, addr_reg):
    b.emit(f"ds_read_b128 {frag(s, o, blk)}, v{addr_reg} offset:{blk * 2048}")


def dma_piece(b, q):
    # M0 = LDS byte address of the piece (wave-uniform); hazard M0 write -> LDS-DMA: s_nop 0
    b.emit(f"s_add_u32 m0, s{S_DMA}, {q * 1024}")
    b.emit("s_nop 0")
    b.emit(f"global_load_lds_dwordx4 v{V_OFF1 if q & 1 else V_OFF0}, s[{SB + 2 * q}:{SB + 2 * q + 1}]")


# experiment switches (environment; the committed .inc is generated with none of them set)
NODMA = os.environ.get("G4_NODMA") == "1"          # no DMA pieces inside the loop (WRONG results: what the loop costs without them)
NOBAR = os.environ.get("G4_NOBAR") == "1"          # no barriers (wrong)
SCHED = int(os.environ.get("G4_SCHED", "2"))       # 1 = the first form (one mid-step barrier, vmcnt(0)); 2 = two barriers, counted vmcnt
B1 = int(os.environ.get("G4_B1", "18"))            # MFMA behind which the first barrier of a step sits
DSTEP = int(os.environ.get("G4_DSTEP", "5"))       # one DMA piece behind every DSTEP-th MFMA
R1 = int(os.environ.get("G4_R1", "1"))             # a K-half-1 fragment read behind every R1-th MFMA from the first
R0 = int(os.environ.get("G4_R0", "1"))             # a K-half-0 fragment read (next stage) behind every R0-th MFMA after the second barrier


def phase(b, swap, cur_set, reads, dmas):
    """64 MFMAs on fragment set `cur_set`.  Fillers: the `reads` = (set, operand, block, address register) of the NEXT phase's
    fragments behind the odd MFMAs of the first half (W first: the next phase's first MFMA group needs all eight W fragments; all of
    them have returned long before the phase ends), one of the `dmas` (piece numbers, three instructions each) behind every fourth
    MFMA of the whole phase — evenly spread pieces measured best (clustered at either end: -3 %)."""
    slots = {}
    for k, x in enumerate(reads):
        slots[2 * k + 1] = ("r", x)
    for k, q in enumerate([] if NODMA else dmas):
        slots[4 * k + 2] = ("d", q)
    n = 0
    for i in range(8):
        for j in range(8):
            mfma(b, swap, cur_set, i, j)
            n += 1
            if n in slots:
                kind, x = slots[n]
                if kind == "r":
                    ds_read(b, *x)
                else:
                    dma_piece(b, x)


def step_one_barrier(b, swap, with_reads, with_dma, last):
    # phase A: set 0, read K half 1 of the current stage into set 1
    rd = [(1, 1, j, V_W1) for j in range(8)] + [(1, 0, i, V_A1) for i in range(8)]
    b.emit("s_waitcnt lgkmcnt(0)")
    phase(b, swap, 0, rd, [])
    # mid-step synchronisation
    if last:
        b.emit("s_waitcnt lgkmcnt(0)")
    else:
        b.emit("s_waitcnt vmcnt(0) lgkmcnt(0)")
        if not NOBAR:
            b.emit("s_barrier")
        for r in (V_A0, V_A1, V_W0, V_W1):          # read addresses move to the other stage
            b.emit(f"v_add_u32 v{r}, s{S_DELTA}, v{r}")
    rd = [(0, 1, j, V_W0) for j in range(8)] + [(0, 0, i, V_A0) for i in range(8)] if with_reads else []
    phase(b, swap, 1, rd, list(range(16)) if with_dma else [])
    if with_dma:
        b.emit(f"v_add_u32 v{V_OFF0}, 128, v{V_OFF0}")
        b.emit(f"v_add_u32 v{V_OFF1}, 128, v{V_OFF1}")
    if not last:
        b.emit(f"s_add_u32 s{S_DMA}, s{S_DMA}, s{S_DELTA}")
        b.emit(f"s_sub_u32 s{S_DELTA}, 0, s{S_DELTA}")


def step(b, swap, with_reads, with_dma, last):
    """One 64-deep K-step = 128 MFMAs (set 0 then set 1, fixed order: the results do not depend on the schedule).

    MFMA   1..31   ds_read K half 1 of the current stage -> set 1 (behind the odd MFMAs, W first)
           36      s_waitcnt lgkmcnt(0); s_barrier      every wave has read ALL of the current stage's buffer (its K half 0 was read
                                                        in the previous step) -> the buffer is free
           38..83  the 16 DMA pieces of K-step kt + 2 into that buffer, one behind every third MFMA
           86      s_waitcnt vmcnt(16); s_barrier       all but this step's 16 pieces have landed = stage kt + 1 is complete, for
                                                        every wave
           88..118 ds_read K half 0 of stage kt + 1 -> set 0 (free since MFMA 64), behind every second MFMA

    A piece issued in step kt is not waited for before MFMA 86 of step kt + 1: 130-176 MFMAs (2100-2800 cycles) of flight, where the
    first form of this loop (everything behind ONE mid-step barrier with vmcnt(0)) left 66-126."""
    if SCHED == 1:
        return step_one_barrier(b, swap, with_reads, with_dma, last)
    slots = {}
    rd1 = [(1, 1, j, V_W1) for j in range(8)] + [(1, 0, i, V_A1) for i in range(8)]
    for k, x in enumerate(rd1):
        slots.setdefault(R1 * k + 1, []).append(("r", x))
    if with_dma and not NODMA:
        slots.setdefault(B1, []).append(("bar1", None))
        for q in range(16):
            slots.setdefault(B1 + 2 + DSTEP * q, []).append(("d", q))
    b2 = B1 + 2 + DSTEP * 15 + 3 if with_dma else 66
    if not last:
        slots.setdefault(b2, []).append(("bar2", 16 if (with_dma and not NODMA) else 0))
        if with_reads:
            rd0 = [(0, 1, j, V_W0) for j in range(8)] + [(0, 0, i, V_A0) for i in range(8)]
            for k, x in enumerate(rd0):
                slots.setdefault(min(b2 + 1 + R0 * k, 127), []).append(("r", x))
    b.emit("s_waitcnt lgkmcnt(0)")
    n = 0
    for s_ in (0, 1):
        if s_ == 1 and not (with_dma and not NODMA and B1 < 64):
            b.emit("s_waitcnt lgkmcnt(0)")             # set 1 complete (no first barrier in this step, or it comes later)
        for i in range(8):
            for j in range(8):
                mfma(b, swap, s_, i, j)
                n += 1
                for kind, x in slots.get(n, []):
                    if kind == "r":
                        ds_read(b, *x)
                    elif kind == "d":
                        dma_piece(b, x)
                    elif kind == "bar1":
                        b.emit("s_waitcnt lgkmcnt(0)")
                        if not NOBAR:
                            b.emit("s_barrier")
                    else:
                        b.emit(f"s_waitcnt vmcnt({x})")
                        if not NOBAR:
                            b.emit("s_barrier")
                        for r in (V_A0, V_A1, V_W0, V_W1):          # read addresses move to the other stage
                            b.emit(f"v_add_u32 v{r}, s{S_DELTA}, v{r}")
    if with_dma:
        b.emit(f"v_add_u32 v{V_OFF0}, 128, v{V_OFF0}")
        b.emit(f"v_add_u32 v{V_OFF1}, 128, v{V_OFF1}")
    if not last:
        b.emit(f"s_add_u32 s{S_DMA}, s{S_DMA}, s{S_DELTA}")
        b.emit(f"s_sub_u32 s{S_DELTA}, 0, s{S_DELTA}")


def loop(swap):
    b = Body()
    e = b.emit
    tag = "s" if swap else "u"
    # ---- prologue: per-piece source bases ----
    e(f"s_mov_b32 s{S_M0}, m0")                                # M0 is compiler-reserved: saved here, restored at the end
    e(f"s_mov_b32 s{S_CNT}, %12")
    e(f"v_mov_b32 v{V_OFF0}, %4")
    e(f"v_mov_b32 v{V_OFF1}, %5")
    e(f"v_mov_b32 v{V_A0}, %0")
    e(f"v_mov_b32 v{V_A1}, %1")
    e(f"v_mov_b32 v{V_W0}, %2")
    e(f"v_mov_b32 v{V_W1}, %3")
    e(f"s_mov_b32 s{S_DMA}, %11")
    e(f"s_mov_b32 s{S_DELTA}, 0x10000")
    for q in range(16):
        e(f"s_add_u32 s{S_T0}, %8, {8 * q}")
        e(f"s_min_u32 s{S_T0}, s{S_T0}, %9")                   # row groups past the matrix end: its last 8 rows (never stored)
        e(f"s_mul_hi_u32 s{S_T1}, s{S_T0}, %10")
        e(f"s_mul_i32 s{S_T0}, s{S_T0}, %10")
        e(f"s_add_u32 s{SB + 2 * q}, %6, s{S_T0}")
        e(f"s_addc_u32 s{SB + 2 * q + 1}, %7, s{S_T1}")
    # ---- stage 0 (+ stage 1) in flight, accumulators cleared while they travel ----
    for q in range(16):
        dma_piece(b, q)
    e(f"v_add_u32 v{V_OFF0}, 128, v{V_OFF0}")
    e(f"v_add_u32 v{V_OFF1}, 128, v{V_OFF1}")
    for r in range(256):
        e(f"v_accvgpr_write_b32 a{r}, 0")
    e("s_waitcnt vmcnt(0)")
    e("s_barrier")
    e(f"s_cmp_lt_u32 s{S_CNT}, 2")
    e(f"s_cbranch_scc1 .Lg4{tag}_nos1_%=")
    e(f"s_add_u32 s{S_DMA}, s{S_DMA}, 0x10000")
    for q in range(16):
        dma_piece(b, q)
    e(f"s_sub_u32 s{S_DMA}, s{S_DMA}, 0x10000")
    e(f"v_add_u32 v{V_OFF0}, 128, v{V_OFF0}")
    e(f"v_add_u32 v{V_OFF1}, 128, v{V_OFF1}")
    e(f".Lg4{tag}_nos1_%=:")
    for j in range(8):
        ds_read(b, 0, 1, j, V_W0)
    for i in range(8):
        ds_read(b, 0, 0, i, V_A0)
    # ---- steps 0 .. nk-3: reads + DMA; step nk-2: reads only; step nk-1: nothing ----
    e(f"s_cmp_lt_u32 s{S_CNT}, 3")
    e(f"s_cbranch_scc1 .Lg4{tag}_tail_%=")
    e(f".Lg4{tag}_loop_%=:")
    step(b, swap, True, True, False)
    e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    e(f"s_cmp_gt_u32 s{S_CNT}, 2")
    e(f"s_cbranch_scc1 .Lg4{tag}_loop_%=")
    e(f".Lg4{tag}_tail_%=:")
    e(f"s_cmp_lt_u32 s{S_CNT}, 2")
    e(f"s_cbranch_scc1 .Lg4{tag}_last_%=")
    step(b, swap, True, False, False)
    e(f".Lg4{tag}_last_%=:")
    step(b, swap, False, False, True)
    e("s_nop 15")                                               # MFMA results -> v_accvgpr_read of the read-out macros
    e("s_nop 15")
    e(f"s_mov_b32 m0, s{S_M0}")
    return b.text()



def clobbers():
    c = [f"a{r}" for r in range(256)] + [f"v{r}" for r in range(V_OFF0, 256)]
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
