#!/usr/bin/env python3
"""Emits mulut_amd/csrc/mulut_tube2_asm.inc: the hand-scheduled gfx950 instruction blocks of stage_tube2_kernel.

The kernel keeps the five table rows of a pass (5 x 32 bytes per lane) in VGPRs ABOVE the compiler's register budget
(__attribute__((amdgpu_waves_per_eu))), so a row read can stay in flight across asm statements: nothing the compiler
allocates can live there.  A block is one pass: per row  s_waitcnt -> 8 v_pk_mad_u16 -> the two ds_read_b128 that
refill the row's registers with the NEXT pass's row.  LDS returns in order, so the waits are plain counts.

    python tools/gen_tube2_asm.py [out]      (re-run after changing the register map; the output is committed)
"""
import os
import sys

DEBUG = int(os.environ.get("TUBE2_DEBUG", "0"))      # timing / debugging builds (written to tools/experiments/tube2_timing; -DMULUT_TUBE2_ASM_INC selects one): 1: s_nop 4 after every wait, 2: every wait drains (lgkmcnt(0)),
# 3: s_nop 4 at the end of every block, 4: no row reads (wrong results: VALU-only time), 5: no MACs (wrong results: LDS-only time)

WAIT_GROUPS = [[int(c) for c in g] for g in os.environ.get("TUBE2_WAITS", "012,34").split(",")]
ROW0 = 88          # v88..v127: rows 0..4, eight dwords each (LO plane x,y,z,w then HI plane x,y,z,w); tuples start on even registers
WAVES_PER_EU = 6   # the kernel is built with amdgpu_waves_per_eu(6, 6): a hard cap of 512 / 6 -> 84 registers for the register
                   # allocator (it was seen to use v0..v80).  amdgpu_num_vgpr does not bind it, and asm clobbers alone only keep LONG-lived
                   # values out: a temporary between two asm statements may still land in a clobbered register.
                   # tests/test_abi_cpu.py audits the ISA: nothing outside the asm blocks may name v88 or above.


def row(j, k):
    return "v%d" % (ROW0 + 8 * j + k)


def rowq(j, plane):
    b = ROW0 + 8 * j + 4 * plane
    return "v[%d:%d]" % (b, b + 3)


SDWA = "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_%d"


def addr_ops(half, base, s):
    """row byte offsets of the pass in half `half` of the packed pair values"""
    o = []
    if half == 0:
        o.append("v_and_b32 %%[a0], 0xffff, %%[%s]" % base)
    else:
        o.append("v_lshrrev_b32 %%[a0], 16, %%[%s]" % base)
    for j in range(3):
        o.append(("v_add_u32_sdwa %%[a%d], %%[a%d], %%[%s%d] " % (j + 1, j, s, j)) + SDWA % half)
    return o


def mac8(j, rev):
    """acc += row j * weight j.  rev: the pass of rotation r + 2 (weights in the high halves): element order reversed,
    i.e. dword k -> 3 - k, LO <-> HI plane, halves of a dword swapped (op_sel on src0)."""
    o = []
    for k in range(4):
        if not rev:
            o.append("v_pk_mad_u16 %%[l%d], %s, %%[w%d], %%[l%d] op_sel:[0,0,0] op_sel_hi:[1,0,1]" % (k, row(j, k), j, k))
            o.append("v_pk_mad_u16 %%[h%d], %s, %%[w%d], %%[h%d] op_sel:[0,0,0] op_sel_hi:[1,0,1]" % (k, row(j, 4 + k), j, k))
        else:
            o.append("v_pk_mad_u16 %%[l%d], %s, %%[w%d], %%[l%d] op_sel:[1,1,0] op_sel_hi:[0,1,1]" % (3 - k, row(j, 4 + k), j, 3 - k))
            o.append("v_pk_mad_u16 %%[h%d], %s, %%[w%d], %%[h%d] op_sel:[1,1,0] op_sel_hi:[0,1,1]" % (3 - k, row(j, k), j, 3 - k))
    return o


def row_loads(j):
    a = "a%d" % (j if j < 4 else 0)
    lo, hi = ("ilo", "ihi") if j < 4 else ("ilo4", "ihi4")
    return ["ds_read_b128 %s, %%[%s] offset:%%[%s]" % (rowq(j, 0), a, lo),
            "ds_read_b128 %s, %%[%s] offset:%%[%s]" % (rowq(j, 1), a, hi)]


def nb_loads(anchor):
    """the six neighbour codes of a rotation pair: the + offsets go straight into the outputs, the - offsets into temporaries"""
    o = []
    for k, nm in enumerate(("pb", "pc", "pd")):
        o.append("ds_read_u16 %%[%s], %%[win] offset:%%[n%d]" % (nm, 2 * k))
        o.append("ds_read_u16 %%[t%d], %%[win] offset:%%[n%d]" % (k, 2 * k + 1))
    if anchor:
        o.append("ds_read_u16 %[ca], %[win] offset:%[nan]")
    return o


def unpack(anchor):
    o = ["v_lshl_or_b32 %[pb], %[t0], 16, %[pb]",
         "v_lshl_or_b32 %[pc], %[t1], 16, %[pc]",
         "v_lshl_or_b32 %[pd], %[t2], 16, %[pd]"]
    if anchor:
        o.append("v_lshl_or_b32 %[ca], %[ca], 16, %[ca]")
    return o


QMAX = 14          # LDS operations a wave may have outstanding: lgkmcnt is a 4-bit counter (15); one is left for an LDS access of
                   # the compiler's own code between two blocks (the parked output rows)


class Queue:
    """The wave's outstanding LDS operations, oldest first (LDS returns in order).  need() emits the counted wait that
    guarantees the named operations have landed; issue() first makes room so that the counter can never saturate."""

    def __init__(self, ops, out):
        self.q = list(ops)
        self.out = out

    def need(self, tags):
        idx = [i for i, t in enumerate(self.q) if t in tags]
        if not idx:
            return
        last = max(idx)
        self.out.append("s_waitcnt lgkmcnt(%d)" % (0 if DEBUG == 2 else len(self.q) - 1 - last))
        if DEBUG == 1:
            self.out.append("s_nop 4")
        self.q = self.q[last + 1:]

    def issue(self, tags):
        if len(self.q) + len(tags) > QMAX:
            keep = QMAX - len(tags)
            self.out.append("s_waitcnt lgkmcnt(%d)" % keep)
            self.q = self.q[len(self.q) - keep:]
        self.q += tags


def rows_tags(p):
    return ["%s%d%s" % (p, j, h) for j in range(5) for h in ("l", "h")]


def block(q, rev, nn, loads, addr_half, cur, nxt):
    """One pass on the queue q (which holds the reads of this pass's rows, tags cur + row + plane).  rev: reversed (second) pass of
    the pair.  nn: neighbour codes fetched in this block (0, 6, or 7 with the anchor): issued after the first wait, merged at the
    end -- they land in the block's own output operands, so nothing of theirs is in flight when the block ends.  loads: refill a
    row's registers with the next pass's row (tags nxt...) right after its MACs (addresses from half addr_half of base / s0..s2)."""
    o = q.out
    if loads:
        o += addr_ops(addr_half, "base", "s")
    for j in range(5):
        # rows are waited for in groups (every s_waitcnt is an instruction of the wave's issue budget): the reads were issued a
        # whole pass ago, only the last ones can still be on their way
        grp = [g for g in WAIT_GROUPS if g[0] == j]
        if grp:
            q.need(["%s%d%s" % (cur, r, h) for r in grp[0] for h in "lh"])
        if j == 0 and nn:
            q.issue(["n%d" % i for i in range(nn)])
            o += nb_loads(nn == 7)
        if DEBUG != 5:
            o += mac8(j, rev)
        if loads:
            q.issue(["%s%dl" % (nxt, j), "%s%dh" % (nxt, j)])
            if DEBUG != 4:
                o += row_loads(j)
    if nn:
        q.need(["n%d" % i for i in range(7)])
        o += unpack(nn == 7)


def cstr(name, lines):
    if DEBUG == 3:
        lines = list(lines) + ["s_nop 4"]
    s = "#define %s \\\n" % name
    s += " \\\n".join('    "%s\\n\\t"' % l for l in lines)
    return s + "\n"


def main():
    out = ["// GENERATED by tools/gen_tube2_asm.py -- do not edit.  Instruction blocks of stage_tube2_kernel (mulut_kernels.hip).",
           "// Private registers (never allocated by the compiler: amdgpu_waves_per_eu(TUBE2_WAVES_PER_EU) caps its budget below them):",
           "//   v%d..v%d rows 0..4 of the pass in flight (8 dwords each)" % (ROW0, ROW0 + 39),
           "#define TUBE2_ROW0 %d" % ROW0, "#define TUBE2_WAVES_PER_EU %d" % WAVES_PER_EU,
           "#define TUBE2_CLOBBERS " + ", ".join('"v%d"' % r for r in range(ROW0, 128)) + ', "memory"', ""]
    # prologue: anchor + neighbours of pair 0, neighbours of pair 1, the ten row reads of the very first pass
    out.append(cstr("TUBE2_ASM_LOAD_NB_ANCHOR", nb_loads(True) + ["s_waitcnt lgkmcnt(0)"] + unpack(True)))
    out.append(cstr("TUBE2_ASM_LOAD_NB", nb_loads(False) + ["s_waitcnt lgkmcnt(0)"] + unpack(False)))
    p1 = addr_ops(0, "base", "s")
    for j in range(5):
        p1 += row_loads(j) if DEBUG != 4 else []
    out.append(cstr("TUBE2_ASM_FIRST_ROWS", p1))
    # a pair = block A (first pass, rotation r: refills with the pair's second pass) + block B (second pass, rotation r + 2: fetches
    # and merges the neighbour codes of the pair after next, refills with the next pair's first pass).  B starts on the queue A
    # leaves and must leave exactly the ten row reads A expects.
    qa = Queue(rows_tags("A"), [])
    block(qa, False, 0, True, 1, "A", "B")
    out.append(cstr("TUBE2_ASM_A", qa.out))
    for nn in (0, 6, 7):
        qb = Queue(qa.q, [])
        block(qb, True, nn, True, 0, "B", "A")
        assert qb.q == rows_tags("A"), qb.q
        out.append(cstr("TUBE2_ASM_B_N%d" % nn, qb.out))
    ql = Queue(qa.q, [])
    block(ql, True, 0, False, 0, "B", "A")
    assert ql.q == [], ql.q
    out.append(cstr("TUBE2_ASM_B_LAST", ql.out))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if DEBUG or os.environ.get("TUBE2_TAG"):     # timing-only generations: built with -DMULUT_TUBE2_ASM_INC='"<path>"'
        path = os.path.join(root, "tools", "experiments", "tube2_timing", "mulut_tube2_asm_dbg%s.inc" % os.environ.get("TUBE2_TAG", DEBUG))
    else:
        path = os.path.join(root, "mulut_amd", "csrc", "mulut_tube2_asm.inc")
    if len(sys.argv) > 1:       # another destination (tests/test_abi_cpu.py compares it with the committed file)
        path = sys.argv[1]
    with open(path, "w") as f:
        f.write("\n".join(out))
    print("wrote", path)


if __name__ == "__main__":
    main()
