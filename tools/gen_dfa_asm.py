#!/usr/bin/env python3
"""Writes tokenizer_amd/csrc/dfa_scan_loop.inc: the automaton scan's loop over the pieces of one
wave block (kernels_dfa.hip: dfa_scan_asm_kernel), gfx950 assembly, for 1, 2 and 4 chains per lane.

Why a generator: the loop is bound by the number of instructions a wave issues (measured: 2.0 - 2.3 us
of the kernel per instruction of a chain-step at 100 MB, whatever the instruction's kind), so the step
is written out instruction by instruction, the C chains of a lane interleaved (every row below is
emitted once per chain, a result is only needed C rows later), and everything rare sits behind ONE
compare and branch per chain-step:
  * the state word's top byte holds kArr (bit 31: the entry arrived on is a virtual one) and partial (bit 30);
    the state a lane takes after a miss, M = code * stride, gets the top byte of the state left, shifted
    left by one, as its top byte -- its sign = that state was partial, a miss LDS may not decide (kArr drops
    out, and so does the mark one step later).  So the sign of the new state finds both kinds of lanes --
    after a miss the one, after a hit the other -- and the block behind the branch sorts them out:
    the global array is asked (lane by lane with scalar loads when the lanes are few -- a vector
    load's wait would also wait for the stream's stores, about 3 us -- all at once otherwise), arrivals
    go to their stream or counter.  The block comes back to where it left.
  * events go to the stream under exec = the lanes that missed (a store's cost grows with its active
    lanes: storing "nothing" from the other lanes to a trash entry cost 25 us of 90), plus lane 63:
    where that lane has no event it writes the entry behind the step's events, which nobody counts
    and the next step overwrites -- so every chain-step issues exactly one store, which the wait
    for the next piece (in issue order) counts on.
The text is plain assembly; operands in %[...] are those of the asm statement in kernels_dfa.hip.

One step of chain j, state word S (dfa.h: kArr | partial | 0 : 6 | label : 8 | LDS base : 16):
    T = S.base + code;  X = lds[T]                           (issued during the previous step)
    M = code * stride, its top byte = S's top byte << 1
    miss = label(X) != code;  S' = miss ? M : X
    miss: S.base to the miss stream (the state left)
Registers.  Chain j: v[40 + 12 j ..]: W0 W1 window (8 codes), N0 N1 next piece, Ta Xa / Tb Xb slot
index and slot word of the even / odd steps, M, R rank / store offset (M R also the pair a global slot
is loaded into), L label (then LDS address), Sb the state after the even steps (the operand S holds
it after the odd ones).  v20 scratch, v32 stride, v34 1, v36 cold_base, v37 2.  s[36 + 2 j ..] miss mask, s[44 + j] its count, s[48:49] lanes the global array
decides, s[56:57] those of them under cold_base, s[64:69] s[72:75] scratch, s70 the next piece's
number, s71 m0, s[78:79] lane 63.
"""
import os
import sys

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tokenizer_amd", "csrc", "dfa_scan_loop.inc")
NAMES = ["W0", "W1", "N0", "N1", "Ta", "Xa", "Tb", "Xb", "M", "R", "L", "Sb"]  # (register pairs start at even numbers)
SL, COLD = "s[48:49]", "s[56:57]"


def v(j, name):
    return "v%d" % (40 + 12 * j + NAMES.index(name))


def m(j):
    return "s[%d:%d]" % (36 + 2 * j, 37 + 2 * j)


def mlo(j):
    return "s%d" % (36 + 2 * j)


def mhi(j):
    return "s%d" % (37 + 2 * j)


def cnt(j):
    return "s%d" % (44 + j)


def gen(C, exp=()):
    main, rare = [], []
    G = lambda j: "%%[G%d]" % j

    def cur(j, k):  # the state before step k
        return "%%[S%d]" % j if k % 2 == 0 else v(j, "Sb")

    def nxt(j, k):  # ... after it
        return v(j, "Sb") if k % 2 == 0 else "%%[S%d]" % j

    def wsel(k):
        return ("W0" if k < 4 else "W1"), k & 3

    def issue(k, state):
        """the LDS reads of step k, for every chain; state(j) = the register of the state"""
        wn, sel = wsel(k)
        T, X = ("Ta", "Xa") if k % 2 == 0 else ("Tb", "Xb")
        for j in range(C):
            main.append("v_add_u32_sdwa %s, %s, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_%d src1_sel:WORD_0"
                        % (v(j, T), v(j, wn), state(j), sel))
        for j in range(C):
            main.append("v_lshlrev_b32 %s, 2, %s" % (v(j, "L"), v(j, T)))
        for j in range(C):
            main.append("ds_read_b32 %s, %s" % (v(j, X), v(j, "L")))

    def rare_block(j, k):
        """chain j at step k: lanes whose new state has something in its top byte"""
        wn, sel = wsel(k)
        T, X = ("Ta", "Xa") if k % 2 == 0 else ("Tb", "Xb")
        t2 = "%%=_%d_%d" % (k, j)
        r = [".Lrare%s:" % t2]
        if "NOSLOW" not in exp:
            r += [
                "v_cmp_gt_i32_e64 %s, 0, %s" % (SL, nxt(j, k)),          # the sign of the new state ...
                "s_and_b64 %s, %s, %s" % (SL, SL, m(j)),                 # ... after a miss: a miss LDS may not decide -- the row's copy in the global array (dfa.h)
                "s_cbranch_scc0 .Lra%s" % t2,
                "s_bcnt1_i32_b64 s64, %s" % SL,
                "s_cmp_gt_u32 s64, 5",
                "s_cbranch_scc1 .Lsv%s" % t2,
                # ---- a few lanes: one by one
                "s_mov_b32 s71, m0",
                "s_mov_b64 s[66:67], 0",                                  # lanes that hit there
                "s_mov_b64 %s, 0" % COLD,                                 # lanes under cold_base that missed there too
                ".Lsw%s:" % t2,
                "s_ff1_i32_b64 s64, %s" % SL,
                "s_bitset0_b64 %s, s64" % SL,
                "v_readlane_b32 s65, %s, s64" % cur(j, k),
                "v_readlane_b32 s68, %s, s64" % v(j, wn),
                "s_and_b32 s69, s65, 0xffff",
                "s_bfe_u32 s68, s68, %d" % (sel * 8 | 0x80000),
                "s_cmp_eq_u32 s69, %[coldbs]",
                "s_cbranch_scc1 .Lsc%s" % t2,
                "s_lshl_b32 s69, s69, 2",
                "s_load_dword s65, %[gof], s69",                          # the state's global base
                "s_waitcnt lgkmcnt(0)",
                "s_branch .Lsg%s" % t2,
                ".Lsc%s:" % t2,
                "v_readlane_b32 s65, %s, s64" % G(j),                     # ... carried under cold_base
                "s_bitset1_b64 %s, s64" % COLD,
                ".Lsg%s:" % t2,
                "s_add_u32 s65, s65, s68",                                # the entry's index
                "s_lshl_b32 s69, s65, 3",
                "s_load_dwordx2 s[72:73], %[slot], s69",
                "s_waitcnt lgkmcnt(0)",
                "s_bfe_u32 s69, s72, 0x80010",                            # its label
                "s_cmp_lg_u32 s69, s68",
                "s_cbranch_scc1 .Lsn%s" % t2,                             # a miss there too
                "s_mov_b32 m0, s64",                                      # (writelane takes one SGPR: the lane comes from m0)
                "v_writelane_b32 %s, s72, m0" % nxt(j, k),
                "v_writelane_b32 %s, s65, m0" % v(j, T),
                "v_writelane_b32 %s, s73, m0" % G(j),
                "s_bitset1_b64 s[66:67], s64",
                "s_bitset0_b64 %s, s64" % COLD,
                "s_bitset0_b64 %s, s64" % m(j),
                ".Lsn%s:" % t2,
                "s_cmp_lg_u64 %s, 0" % SL,
                "s_cbranch_scc1 .Lsw%s" % t2,
                "s_mov_b32 m0, s71",
                "s_mov_b64 exec, %s" % COLD,                              # the counters, for the lanes concerned
                "v_subrev_u32 v20, %%[hot], %s" % G(j),
                "v_lshlrev_b32 v20, 2, v20",
                "global_atomic_add v20, v34, %[cmiss]",
                "s_mov_b64 exec, s[66:67]",
                "s_branch .Lsh%s" % t2,
                # ---- many lanes: all at once
                ".Lsv%s:" % t2,
                "s_mov_b64 exec, %s" % SL,
                "v_cmp_eq_u32_sdwa %s, %s, v36 src0_sel:WORD_0 src1_sel:DWORD" % (COLD, cur(j, k)),
                "s_andn2_b64 exec, %s, %s" % (SL, COLD),
                "v_lshlrev_b32_sdwa %s, v37, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" % (v(j, "L"), cur(j, k)),
                "global_load_dword %s, %s, %%[gof]" % (G(j), v(j, "L")),
                "s_waitcnt vmcnt(0)",
                "s_mov_b64 exec, %s" % SL,
                "v_add_u32_sdwa %s, %s, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_%d src1_sel:DWORD" % (v(j, "L"), v(j, wn), G(j), sel),
                "v_lshlrev_b32 %s, 3, %s" % (v(j, X), v(j, "L")),
                "global_load_dwordx2 v[%s:%s], %s, %%[slot]" % (v(j, "M")[1:], v(j, "R")[1:], v(j, X)),
                "s_waitcnt vmcnt(0)",
                "v_bfe_u32 v20, %s, 16, 8" % v(j, "M"),
                "v_cmp_eq_u32_sdwa s[66:67], %s, v20 src0_sel:BYTE_%d src1_sel:DWORD" % (v(j, wn), sel),
                "s_andn2_b64 %s, %s, s[66:67]" % (m(j), m(j)),            # a hit there is no miss
                "s_andn2_b64 exec, %s, s[66:67]" % COLD,                  # still a miss, under cold_base: the state's own counter
                "v_subrev_u32 v20, %%[hot], %s" % G(j),
                "v_lshlrev_b32 v20, 2, v20",
                "global_atomic_add v20, v34, %[cmiss]",
                "s_mov_b64 exec, s[66:67]",
                "v_mov_b32 %s, %s" % (nxt(j, k), v(j, "M")),
                "v_mov_b32 %s, %s" % (v(j, T), v(j, "L")),
                "v_mov_b32 %s, %s" % (G(j), v(j, "R")),
                # ---- both: the hits there (exec), counted for the next choice of what LDS holds; arrivals among them
                ".Lsh%s:" % t2,
                "v_subrev_u32 v20, %%[hot], %s" % v(j, T),
                "v_lshlrev_b32 v20, 2, v20",
                "global_atomic_add v20, v34, %[chit]",
                "v_cmp_gt_i32_e64 s[74:75], 0, %s" % nxt(j, k),
                "s_mov_b64 exec, s[74:75]",
                "global_atomic_add v20, v34, %[carr]",
                "s_mov_b64 exec, -1",
                ".Lra%s:" % t2,
            ]
        if "NOARR" not in exp:
            r += [
                "v_cmp_gt_i32_e64 s[64:65], 0, %s" % nxt(j, k),             # the sign after a hit: an arrival on a virtual entry ...
                "v_cmp_gt_u32_e64 s[66:67], %%[hot], %s" % v(j, T),         # ... of LDS (the others were counted above)
                "s_andn2_b64 s[64:65], s[64:65], %s" % m(j),
                "s_and_b64 exec, s[64:65], s[66:67]",
                "v_mbcnt_lo_u32_b32 v20, exec_lo, 0",
                "v_mbcnt_hi_u32_b32 v20, exec_hi, v20",
                "v_lshl_add_u32 v20, v20, 1, %[cura]",
                "global_store_short v20, %s, %%[basea]" % v(j, T),
                "s_bcnt1_i32_b64 s68, exec",
                "s_lshl1_add_u32 %[cura], s68, %[cura]",
                "s_mov_b64 exec, -1",
            ]
        if "SEARCH" in exp:
            # the search automaton (dfa.h: BuildSearchDfa): no partial rows, no miss events; an arrival = some
            # pattern ends at this byte: the entry arrived on and the byte's place in the prepared corpus
            r = [".Lrare%s:" % t2,
                 "v_cmp_gt_i32_e64 s[64:65], 0, %s" % nxt(j, k),
                 "s_lshl_b32 s68, %[p], 3",
                 "s_add_u32 s68, s68, %d" % k,
                 # (loads and stores share one counter: the piece under way is waited for before the piece's first
                 # store, so that the wait at the piece's end never waits for stores)
                 "s_cmp_lg_u32 s77, 0",
                 "s_cbranch_scc1 .Lsk%s" % t2,
                 "s_waitcnt vmcnt(0)",
                 "s_mov_b32 s77, 1",
                 ".Lsk%s:" % t2,
                 "s_mov_b64 exec, s[64:65]",
                 "v_mbcnt_lo_u32_b32 v20, exec_lo, 0",
                 "v_mbcnt_hi_u32_b32 v20, exec_hi, v20",
                 "v_lshl_add_u32 v20, v20, 3, %[cura]",
                 "v_bfe_u32 %s, %s, 24, 6" % (v(j, "M"), nxt(j, k)),          # (dfa.h: SearchImage::kLenShift)
                 "v_lshl_or_b32 %s, %s, 16, %s" % (v(j, "M"), v(j, "M"), v(j, T)),
                 "v_add_u32 %s, s68, %%[P%d]" % (v(j, "R"), j),
                 "global_store_dwordx2 v20, v[%s:%s], %%[basea]" % (v(j, "M")[1:], v(j, "R")[1:]),
                 "s_bcnt1_i32_b64 s68, exec",
                 "s_lshl3_add_u32 %[cura], s68, %[cura]",
                 "s_mov_b64 exec, -1"]
        r.append("s_branch .Lbr%s" % t2)
        return r

    def step(k):
        nonlocal main, rare
        wn, sel = wsel(k)
        T, X = ("Ta", "Xa") if k % 2 == 0 else ("Tb", "Xb")
        mark = "NOSLOW" not in exp and "SEARCH" not in exp

        def after_miss():  # the state after a miss, marked where the state left is a partial one
            for j in range(C):
                main.append("v_mul_u32_u24_sdwa %s, %s, v32 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_%d src1_sel:DWORD"
                            % (v(j, "M"), v(j, wn), sel))
            if mark:
                for j in range(C):
                    main.append("v_lshlrev_b32_sdwa %s, v34, %s dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_3"
                                % (v(j, "M"), cur(j, k)))

        # (a mask written by a vector compare: two wait states before a vector instruction reads it; a register
        # written in part, dst_sel: one before it is read.  With fewer than three chains M's rows go between.)
        late = C < 3
        if not late:
            after_miss()
        main.append("s_waitcnt lgkmcnt(0)")
        for j in range(C):  # (the label is byte 2 of the slot word)
            main.append("v_cmp_ne_u32_sdwa %s, %s, %s src0_sel:BYTE_%d src1_sel:BYTE_2" % (m(j), v(j, wn), v(j, X), sel))
        if late:
            after_miss()
            if C == 1:
                main.append("s_nop 0")
        for j in range(C):
            main.append("v_cndmask_b32_e64 %s, %s, %s, %s" % (nxt(j, k), v(j, X), v(j, "M"), m(j)))
        # ---- anything rare
        if not ("NOSLOW" in exp and "NOARR" in exp):
            for j in range(C):
                t2 = "%%=_%d_%d" % (k, j)
                main.append("v_cmp_gt_i32_e64 vcc, 0, %s" % nxt(j, k))
                main.append("s_cbranch_vccnz .Lrare%s" % t2)
                main.append(".Lbr%s:" % t2)
                rare += rare_block(j, k)
        # ---- the next lookups under way
        if k < 7:
            issue(k + 1, lambda j: nxt(j, k))
        # ---- the states left by a miss: their base to the miss stream
        if "NOLOG" not in exp and "SEARCH" not in exp:
            for j in range(C):
                main.append("v_mbcnt_lo_u32_b32 %s, %s, 0" % (v(j, "R"), mlo(j)))
            for j in range(C):
                main.append("v_mbcnt_hi_u32_b32 %s, %s, %s" % (v(j, "R"), mhi(j), v(j, "R")))
            # (the batch as a buffer of 16-bit records: the store's index is the rank, its scalar offset the stream's fill in
            # bytes -- the address arithmetic a global store needs, "rank * 2 + fill", is the hardware's.  The fill moves on
            # chain by chain, so every chain's store goes with the fill before it: stores and scalar adds alternate.)
            for j in range(C):
                main.append("s_or_b64 exec, %s, s[78:79]" % m(j))
                main.append("buffer_store_short %s, %s, %%[srd], %%[curm] idxen" % (cur(j, k), v(j, "R")))
                main.append("s_bcnt1_i32_b64 %s, %s" % (cnt(j), m(j)))
                main.append("s_lshl1_add_u32 %%[curm], %s, %%[curm]" % cnt(j))
            main.append("s_mov_b64 exec, -1")

    # ---- the loop
    main += [
        "v_mov_b32 v32, %[stride]",
        "v_mov_b32 v34, 1",
        "v_mov_b32 v36, %[coldb]",
        "v_mov_b32 v37, 2",
        "s_mov_b32 s78, 0",
        "s_mov_b32 s79, 0x80000000",
        "s_mov_b32 s77, 0",
    ]
    for j in range(C):
        main.append("global_load_dwordx2 v[%s:%s], %%[voff], %%[src] offset:%d" % (v(j, "W0")[1:], v(j, "W1")[1:], 512 * j))
    main += ["s_waitcnt vmcnt(0)", ".Lpiece%=:", "s_add_u32 s70, %[p], 1", "s_cmp_lt_u32 s70, %[pieces]", "s_cbranch_scc0 .Lnopf%="]
    main.append("v_add_u32 %%[voff], %d, %%[voff]" % (512 * C))     # the next piece, under way during this one
    for j in range(C):
        main.append("global_load_dwordx2 v[%s:%s], %%[voff], %%[src] offset:%d" % (v(j, "N0")[1:], v(j, "N1")[1:], 512 * j))
    main.append(".Lnopf%=:")
    issue(0, lambda j: "%%[S%d]" % j)
    for k in range(8):
        step(k)
    stores = 0 if ("NOLOG" in exp or "SEARCH" in exp) else 8 * C
    if "SEARCH" in exp:
        main += ["s_cmp_lg_u32 s77, 0", "s_cbranch_scc1 .Lhave%=", "s_waitcnt vmcnt(0)", ".Lhave%=:", "s_mov_b32 s77, 0"]
    else:
        main.append("s_waitcnt vmcnt(%d)" % stores)  # (in issue order: the piece came before the body's stores, one per step and chain)
    for j in range(C):
        main.append("v_mov_b32 %s, %s" % (v(j, "W0"), v(j, "N0")))
        main.append("v_mov_b32 %s, %s" % (v(j, "W1"), v(j, "N1")))
    main += [
        "s_mov_b32 %[p], s70",
        "s_cmp_ge_u32 %[p], %[pieces]",
        "s_cbranch_scc1 .Lout%=",
        "s_cmp_gt_u32 %[curm], %[high]",
        "s_cbranch_scc1 .Lout%=",
        "s_cmp_gt_u32 %[cura], %[high]",
        "s_cbranch_scc0 .Lpiece%=",
        ".Lout%=:",
        "s_branch .Lend%=",
    ]
    return main + rare + [".Lend%=:"]


def clobbers(C):
    regs = ["v20", "v32", "v34", "v36", "v37"] + ["v%d" % r for r in range(40, 40 + 12 * C)]
    regs += ["s%d" % r for r in range(36, 36 + 2 * C)] + ["s%d" % (44 + j) for j in range(C)]
    regs += ["s48", "s49", "s56", "s57"] + ["s%d" % r for r in range(64, 76)] + ["s77", "s78", "s79"]
    return regs + ["vcc", "scc", "memory"]


def emit(f, name, lines):
    f.write("#define %s \\\n" % name)
    for n, l in enumerate(lines):
        f.write('\t"%s\\n"%s\n' % (l, " \\" if n + 1 < len(lines) else ""))


def main():
    with open(OUT, "w") as f:
        f.write("// dfa_scan_loop.inc -- GENERATED by tools/gen_dfa_asm.py (the schedule and the register plan are explained there); do not edit.\n")
        for C in (1, 2, 4):
            f.write("#if defined(DFA_EXP_NOSLOW) || defined(DFA_EXP_NOLOG) || defined(DFA_EXP_NOARR)\n")
            f.write("// (timing experiments of csrc/Makefile exp_dfa: parts of the step left out, results wrong)\n")
            first = True
            for combo in (("NOSLOW", "NOLOG", "NOARR"), ("NOSLOW",), ("NOLOG",), ("NOARR",)):
                cond = " && ".join(("defined(DFA_EXP_%s)" % x) if x in combo else ("!defined(DFA_EXP_%s)" % x) for x in ("NOSLOW", "NOLOG", "NOARR"))
                f.write("#%s %s\n" % ("if" if first else "elif", cond))
                first = False
                emit(f, "DFA_GEN_LOOP%d" % C, gen(C, combo))
            f.write("#endif\n#else\n")
            emit(f, "DFA_GEN_LOOP%d" % C, gen(C))
            f.write("#endif\n")
            f.write("#define DFA_GEN_CLOBBERS%d %s\n" % (C, ", ".join('"%s"' % r for r in clobbers(C))))
            emit(f, "DFA_GEN_SEARCH%d" % C, gen(C, ("SEARCH",)))
    print("wrote", os.path.normpath(OUT))


if __name__ == "__main__":
    sys.exit(main())
