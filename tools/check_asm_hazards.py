"""Scan gfx950 device assembly for the wait-state rules the compiler cannot keep around inline assembly.

The compiler's hazard recognizer inserts the s_nop's the hardware needs between its own instructions; it does not look
inside an inline-assembly statement.  Two rules of the CDNA ISA matter for the statements this library contains:

  A. a vector-memory store of more than 64 bits reads its data registers up to two wait states after it issues: no VALU
     instruction may write one of them inside that window (every path out of the store is followed, branches included);
  B. a vector-memory instruction that reads an SGPR written by a VALU instruction (v_readfirstlane, v_readlane, a
     compare or carry with an SGPR destination) needs five wait states after that write (straight-line code before the
     statement is examined).

usage: hipcc <flags> --cuda-device-only -S -o unit.s unit.hip ; python tools/check_asm_hazards.py unit.s [...]
Exit status 1 when a violation is found.
"""
import re
import sys

REG = re.compile(r"\b([vs])(\d+)\b|\b([vs])\[(\d+):(\d+)\]")
TWO_DEST = re.compile(r"v_(add_co|sub_co|subrev_co|addc_co|subb_co|subbrev_co|div_scale|mad_u64_u32|mad_i64_i32)")


def regs(text, kind):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) == kind:
            out.add(int(m.group(2)))
        elif m.group(3) == kind:
            out.update(range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def wait_states(ins):
    m = re.match(r"s_nop\s+(\d+)", ins)
    return int(m.group(1)) + 1 if m else 1


def dest_text(ins):
    """Destination operand(s) of a VALU instruction: the first operand, two for carries, divides and 64-bit mads."""
    parts = ins.split(None, 1)
    if len(parts) < 2:
        return ""
    ops = [o.strip() for o in parts[1].split(",")]
    return ",".join(ops[:2 if TWO_DEST.match(parts[0]) else 1])


def parse(path):
    """-> list of (line number, instruction, inside inline asm, kernel), labels: name -> index of the next instruction"""
    prog, labels = [], {}
    in_asm, kernel = False, "?"
    with open(path) as f:
        for ln, raw in enumerate(f, 1):
            if ";;#ASMSTART" in raw:
                in_asm = True
                continue
            if ";;#ASMEND" in raw:
                in_asm = False
                continue
            m = re.match(r"^([A-Za-z_.$][\w.$]*):", raw)
            if m:
                labels[(kernel if m.group(1).startswith(".L") else "", m.group(1))] = len(prog)
                if not m.group(1).startswith(".L"):
                    kernel = m.group(1)
                    labels[(kernel, m.group(1))] = len(prog)
                continue
            if not raw.startswith("\t"):
                continue
            ins = raw.split(";")[0].strip()
            if not ins or ins.startswith("."):
                continue
            prog.append((ln, ins, in_asm, kernel))
    return prog, labels


def follow(prog, labels, i, data, budget, store, path, found, depth=0):
    """Rule A along every path that starts at instruction i with `budget` wait states still to pass."""
    while i < len(prog) and budget > 0 and depth < 8:
        ln, ins, _, kernel = prog[i]
        if ins.startswith("v_") and regs(dest_text(ins), "v") & data:
            found.add((path, ln, kernel, ins, store[1], store[0], budget))
            return
        if ins.startswith("s_endpgm"):
            return
        budget -= wait_states(ins)
        m = re.match(r"s_(c?branch\w*)\s+(\S+)", ins)
        if m:
            tgt = labels.get((kernel, m.group(2)))
            if tgt is not None and budget > 0:
                follow(prog, labels, tgt, data, budget, store, path, found, depth + 1)
            if m.group(1) == "branch":
                return
        i += 1


def scan(path):
    prog, labels = parse(path)
    found = set()
    nb = 0
    for i, (ln, ins, in_asm, kernel) in enumerate(prog):
        if not in_asm or not re.match(r"(global|buffer|flat|scratch)_", ins):
            continue
        # rule B
        need = regs(ins, "s")
        ws, k = 0, i - 1
        while k >= 0 and ws < 5 and prog[k][3] == kernel:
            prev = prog[k][1]
            if prev.startswith("v_") and regs(dest_text(prev), "s") & need:
                print("%s:%d: [B] %s: '%s' reads an SGPR that '%s' wrote %d wait state(s) earlier" % (path, ln, kernel, ins, prev, ws))
                nb += 1
            ws += wait_states(prev)
            k -= 1
        # rule A
        m = re.match(r"(global|buffer|flat)_store_dwordx([34])\s+(.*)", ins)
        if m:
            ops = [o.strip() for o in m.group(3).split(",")]
            data = regs(ops[0] if m.group(1) == "buffer" else ops[1], "v")
            follow(prog, labels, i + 1, data, 2, (ln, ins), path, found)
    for p, ln, kernel, ins, store, sln, budget in sorted(found):
        print("%s:%d: [A] %s: '%s' writes data of '%s' (line %d), %d wait state(s) short" % (p, ln, kernel, ins, store, sln, budget))
    return nb + len(found)


def main():
    total = 0
    for p in sys.argv[1:]:
        total += scan(p)
    print("%d violation(s)" % total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
