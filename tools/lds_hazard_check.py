#!/usr/bin/env python3
"""Static check of hand-counted LDS reads in a kernel's ISA (conv_regw.hip issues its fragment reads as inline-asm `ds_read_b128`
whose destination the compiler believes is complete at once, and waits for them with hand-counted `s_waitcnt lgkmcnt(n)`).

For every kernel of an assembly file (`hipcc -S --cuda-device-only`): walk the instructions in program order, keep the FIFO of outstanding
LGKM operations (LDS and scalar-memory instructions; LDS results return in order), retire entries at every `s_waitcnt ... lgkmcnt(n)`, and
report any instruction that reads or overwrites a register that an OUTSTANDING `ds_read` is still going to write.  Branch targets reset
the FIFO conservatively to "whatever was outstanding at the jump" -- the kernels checked here keep the window inside straight-line code
between barriers, so a label is treated as a join with the fall-through state (enough for the loops in conv_regw.hip; a hazard across
a back edge is reported by the second pass over a loop body because the FIFO at the bottom is carried to the label once).

usage: lds_hazard_check.py file.s [kernel-name-substring]      exit status 1 when a hazard is found
"""
import re
import sys

REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def check_kernel(lines):
    """lines: instruction strings of one kernel.  Returns a list of (line index, instruction, outstanding read) hazards."""
    fifo = []            # entries: None (an LGKM op without a vector destination) or (dest register set, index, text)
    hazards = []
    carried = {}         # label -> FIFO snapshot at the first jump to it (back edges)
    for i, ins in enumerate(lines):
        ins = ins.split(";")[0].strip()
        if not ins:
            continue
        if ins.endswith(":"):
            continue
        op, _, rest = ins.partition(" ")
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", rest)
            if m:
                n = int(m.group(1))
                while len(fifo) > n:
                    fifo.pop(0)
            continue
        pending = [e for e in fifo if e is not None]
        if pending and not op.startswith("s_"):               # every vector / LDS / memory instruction: sources and destinations alike
            touched = regs(rest)
            for dest, j, text in pending:
                if touched & dest:
                    hazards.append((i, ins, text))
        if op.startswith("ds_read") or op.startswith("ds_load"):
            dest = regs(rest.split(",")[0])
            fifo.append((dest, i, ins))
        elif op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
            fifo.append(None)
        elif op in ("s_barrier",):
            pass
    return hazards


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        body.append(line)
        if "s_endpgm" in line:
            yield name, body
            name = None


def main():
    path, want = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    bad = 0
    n = 0
    for name, body in kernels(path):
        if want and want not in name:
            continue
        n += 1
        hz = check_kernel(body)
        reads = sum(1 for l in body if "ds_read" in l)
        print(f"{name[:90]}: {len(body)} lines, {reads} LDS reads, {len(hz)} hazards")
        for i, ins, text in hz[:8]:
            print(f"    line {i}: `{ins}` touches the destination of outstanding `{text}`")
        bad += len(hz)
    print(f"{n} kernels checked, {bad} hazards")
    return 1 if bad or not n else 0


if __name__ == "__main__":
    sys.exit(main())
