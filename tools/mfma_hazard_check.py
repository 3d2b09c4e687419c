"""Static wait-state check of the hand-register MFMA kernels, on the gfx950 disassembly of a built object.

Why: the MFMAs of mlp_bf16.hip / mlp_f16s*.hip / dgrad_f16s.hip are `asm volatile` statements.  hipcc allocates their VGPR operands, but its
hazard recogniser only inserts the software wait states of the CDNA3/4 ISA guide (section 4.5 "manually inserted wait states"; LLVM
GCNHazardRecognizer::checkMAIVALUHazards / checkMAILdStHazards, gfx940 + gfx950 rows) around instructions it KNOWS to be matrix instructions.
For an asm MFMA nothing is inserted, so these are the kernels' own obligation:

  RAW / WAW   an MFMA of P passes writes its vdst P + 4 wait states after issue (gfx950 XDL shapes: P + 3, + 1 for P != 2; the f32-input
              shapes: P + 2 -- which is exactly what hipcc pads its own 32x32x2 read-outs with, `s_nop 15; s_nop 1`).  A VALU / VMEM / LDS
              instruction that reads OR WRITES a register of that tuple earlier sees stale data -- or is itself overwritten when the matrix
              result lands.  The second case is what took down round 3's `-DMN_F16S_NOPACK` ablation build: with the packing elided the
              accumulators were dead on arrival, the allocator reused their registers at once (`v_mfma v[54:57]` ... `v_lshl_add_u64
              v[54:55]` ... `global_load_lds_dwordx4 v[54:55]`, and `v_mfma v[12:15]` ... barrier ... `v_add_u32 v12, 0x8000, v74` =
              the ring's fetch offset), and the late matrix write turned an address into float bits: a page fault.
  WAR on C    the pipe reads srcC late: a VALU write of srcC needs 7 (8-pass) / 13 (16-pass) wait states after the MFMA.
  MFMA A / B  an MFMA reading another MFMA's vdst as A or B needs the same P + 4 (C -> C of the same shape is forwarded by hardware).
  operands    an AGPR written by `v_accvgpr_write` (a packed fragment) may be an MFMA operand 2 wait states later at the earliest.
Both register files are tracked: VGPR destinations (the asm MFMAs) and AGPR destinations (the builtin MFMAs of the fp32 / training kernels,
which hipcc pads itself -- they ride along as a check of the checker -- and their explicit `v_accvgpr_read` read-outs).

Counting: every instruction is one wait state (one pass = 4 clocks), `s_nop N` is N + 1.  An intervening MFMA counts as its ISSUE INTERVAL, P / 2
wait states (16x16x32 f16 / bf16: 8 passes of latency, a new one every 4 passes = 16 clocks, which is what the 2.5 PFLOP/s peak is made of and
what tools/mfma_probe*.hip measured): the in-order wave cannot issue the instruction behind it earlier than that, whatever the pipe was doing.
`--strict` counts it as ONE like LLVM's recogniser does (which never relies on the pipe's rate); the kernels' packing schedule is built on the
issue interval ("a tile's first stage is >= 8 MFMA issues behind the MFMA that finished it"), so strict mode does not gate at zero: its
per-kernel counts -- the number of pairs that are safe only BECAUSE of the issue interval -- are pinned in tests/test_packing_cpu.py
(STRICT_RELIANCE), so that a compiler upgrade which makes the kernels lean on the pipe's rate more than today is a red CPU test.

    python tools/mfma_hazard_check.py OBJECT_OR_SO [kernel-substring] [--strict]     # exit 1 if any pair is inside its window
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
AREG = re.compile(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]")
PASSES = {"16x16x32": 8, "32x32x16": 16, "32x32x2": 16, "16x16x4": 8, "32x32x8": 16, "16x16x16": 8, "4x4x4": 2, "32x32x1": 16, "16x16x1": 8}
# instructions whose first operand is not a vector destination
NO_VDST = ("global_store", "ds_write", "ds_store", "global_load_lds", "buffer_store", "flat_store", "scratch_store", "s_", "v_cmp", "v_accvgpr_write",
           "v_readfirstlane", "v_readlane", "global_atomic", "buffer_atomic", "ds_add", "ds_max", "ds_min", "ds_or", "ds_and", "global_wb", "global_inv", "buffer_wbl2", "buffer_inv")


def device_asm(path: str) -> str:
    """Disassembly of the gfx950 code object inside a host object / shared library built by hipcc."""
    work = tempfile.mkdtemp()
    try:
        local = os.path.join(work, "k.o")
        shutil.copy(path, local)
        r = subprocess.run([OBJDUMP, "--offloading", local], capture_output=True, cwd=work)
        if r.returncode != 0 and not any(f.endswith("gfx950") for f in os.listdir(work)):
            return ""                                           # no offload bundle at all
        dev = [f for f in os.listdir(work) if f.endswith("gfx950")]
        if not dev:
            return ""                                           # a host-only object (pack.cpp.o): no device code, nothing to check
        assert len(dev) == 1, os.listdir(work)
        return subprocess.run([OBJDUMP, "-d", os.path.join(work, dev[0])], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(work)


def _regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def _aregs(tok):
    """AGPRs named by an operand, as negative numbers - 1 - N (one register space for both files in check())."""
    out = set()
    for m in AREG.finditer(tok):
        if m.group(1) is not None:
            out.add(-1 - int(m.group(1)))
        else:
            out.update(-1 - r for r in range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def _allregs(tok):
    return _regs(tok) | _aregs(tok)


def kernels_of(asm: str, want: str = ""):
    ks, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            cur = m.group(1)
            ks[cur] = []
            continue
        if cur is None or not line.startswith("\t"):
            continue
        body = line.split("//")[0].strip()
        if body:
            op, _, rest = body.partition(" ")
            ks[cur].append((op, [o.strip() for o in rest.split(",")] if rest else [], body))
    return {k: v for k, v in ks.items() if want in k}


def _passes(op):
    m = re.search(r"_(\d+x\d+x\d+)_?", op)
    return PASSES.get(m.group(1), 16) if m else 16


def _issue_interval(op):
    """Wait states an MFMA holds the in-order wave before the next instruction can issue behind it: P / 2 for the f16 / bf16 shapes of
    gfx950 (16x16x32: 16 clocks, 32x32x16: 32), P for the f32-input shapes (32x32x2: 64 clocks, 16x16x4: 32) -- MI355X_MICROARCH.md,
    per-instruction cycle constants."""
    P = _passes(op)
    return P if op.endswith("_f32") and re.search(r"x\d+_?f32$", op) else max(1, P // 2)


def check(ins, strict=False):
    """-> (number of MFMAs, list of violation strings) for one kernel's instruction list."""
    bad, n_mfma = [], 0
    for i, (op, ops, body) in enumerate(ins):
        if not op.startswith("v_mfma"):
            continue
        n_mfma += 1
        P = _passes(op)
        dst = _allregs(ops[0])                                # VGPR tuple (the asm MFMAs) or AGPR tuple (builtin MFMAs: hipcc pads those itself)
        srcc = _regs(ops[3]) if len(ops) > 3 and ops[3].startswith("v") else set()
        xdl = not re.search(r"x\d+_?f32$", op)                # f16 / bf16 / i8 shapes run on the XDL pipe; the f32-input shapes are "SMFMA"
        need_rw = P + 4 if xdl else P + 2                     # LLVM: GFX940_XDL_N_PassWriteVgprVALU...(gfx950) = P + 3 + 1; SMFMA{16x16,32x32}WriteVgprVALU... = 10, 18
        need_war = 7 if P == 8 else 13 if P == 16 else P - 1
        ws = 0
        for j in range(i + 1, len(ins)):
            if ws >= need_rw:
                break
            op2, ops2, body2 = ins[j]
            if op2 == "s_nop":
                ws += int(ops2[0], 0) + 1
                continue
            if op2.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                break                                             # straight-line check; the kernels' loops close over >> 20 instructions
            writes = _regs(ops2[0]) if ops2 and ops2[0].startswith("v") and not op2.startswith(NO_VDST) else set()
            if op2.startswith("v_accvgpr_write"):
                writes = _aregs(ops2[0])
            reads = set()
            for k, o in enumerate(ops2):
                if not (k == 0 and writes):
                    reads |= _allregs(o)
            if op2.startswith("v_mfma"):
                writes = _allregs(ops2[0])
                ab = _allregs(ops2[1]) | _allregs(ops2[2])
                c2 = _allregs(ops2[3]) if len(ops2) > 3 else set()
                if dst & ab:
                    bad.append(f"MFMA A/B reads vdst +{ws}ws (<{need_rw}): [{body}] -> [{body2}]")
                if (dst & c2) and (c2 != dst or _passes(op2) != P):
                    bad.append(f"MFMA C overlaps vdst partially +{ws}ws: [{body}] -> [{body2}]")
                if dst & writes and writes != dst:
                    bad.append(f"MFMA vdst overlaps vdst partially +{ws}ws: [{body}] -> [{body2}]")
            else:
                if dst & writes:
                    bad.append(f"WAW +{ws}ws (<{need_rw}): [{body}] -> [{body2}]")
                if dst & reads:
                    bad.append(f"RAW +{ws}ws (<{need_rw}): [{body}] -> [{body2}]")
                if srcc and not (srcc & dst) and (srcc & writes) and ws < need_war:
                    bad.append(f"WAR(C) +{ws}ws (<{need_war}): [{body}] -> [{body2}]")
            ws += 1 if (strict or not op2.startswith("v_mfma")) else _issue_interval(op2)
    # an AGPR written by v_accvgpr_write may be an MFMA's operand 2 wait states later at the earliest (cdna_hip_programming.md 5.7 item 2:
    # "v_accvgpr_write -> MFMA operand: s_nop 1"); the kernels write their fragments a whole group ahead
    for i, (op, ops, body) in enumerate(ins):
        if not op.startswith("v_accvgpr_write") or not ops:      # (a VGPR written by an ordinary VALU instruction is interlocked: hipcc itself
            continue                                             #  builds the asm MFMAs' C tuples with v_mov right in front of them)
        w = _aregs(ops[0])
        ws = 0
        for j in range(i + 1, min(i + 4, len(ins))):
            if ws >= 2:
                break
            op2, ops2, body2 = ins[j]
            if op2 == "s_nop":
                ws += int(ops2[0], 0) + 1
                continue
            if op2.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                break
            if op2.startswith("v_mfma") and w & (_allregs(ops2[1]) | _allregs(ops2[2]) | (_allregs(ops2[3]) if len(ops2) > 3 else set())):
                bad.append(f"VALU write -> MFMA operand +{ws}ws (<2): [{body}] -> [{body2}]")
            ws += 1
    return n_mfma, bad


def main(argv):
    strict = "--strict" in argv
    argv = [a for a in argv if a != "--strict"]
    asm = open(argv[1]).read() if argv[1].endswith(".s") else device_asm(argv[1])
    if not asm:
        print(f"{argv[1]}: no device code")
        return 0
    total = 0
    for name, ins in kernels_of(asm, argv[2] if len(argv) > 2 else "").items():
        n, bad = check(ins, strict)
        if n == 0:
            continue
        for b in bad[:12]:
            print("   ", b)
        print(f"{name[:110]}: {n} MFMAs, {len(ins)} instructions, {len(bad)} pairs inside their window")
        total += len(bad)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
