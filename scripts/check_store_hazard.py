"""ISA lint of the SHIPPED library: no vector-memory store of more than 64 bits may be followed within two issue slots by an
instruction that writes one of its data VGPRs.

Why (measured on MI355X, scripts/micro/t_store_hazard.hip -> profiles/r03_t_store_hazard.txt): behind `buffer_store_dwordx4` /
`global_store_dwordx4` the hardware reads the data registers late enough that a VALU write issued right behind the store can
reach memory instead of the stored value -- 0.4 % of the stores with an SGPR soffset, 22 % with a literal one, under
back-pressure; ONE wait state still loses 0.5-1.7 % of the literal-soffset / global forms, TWO are always enough.  hipcc pads one
wait state, and none at all when a buffer store's soffset is a register.  The kernels' own 16-byte buffer stores go through
`buffer_store_b128_sreg` (csrc/common.h), which carries its wait states; this script checks what the compiler scheduled around
every other wide store.

Second check (scan_counted_waits): no register spill (scratch_*) inside a loop that waits with a counted `s_waitcnt vmcnt(n > 0)`
-- the kernels that count their vector-memory instructions (expand_res_bf16, conv_bal_bf16, bottleneck128_bf16, the fp32
quarter tiles) would wait for the wrong instruction.

usage: check_store_hazard.py [libposerisk_hip.so]   (exit status 1 and a listing when a violation is found)
Extracts the gfx950 code objects from the library's offload bundles and disassembles them with llvm-objdump."""
import os
import re
import struct
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
NEED = 2   # wait states between the store and a write of its data registers


def code_objects(so_path):
    """The device ELFs (triples naming gfx950) of every offload bundle in the file."""
    blob = open(so_path, "rb").read()
    out, pos = [], 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return out
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if "gfx950" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos = q


def vregs(operand):
    """VGPR numbers named by one operand ('v7', 'v[4:7]'), else ()."""
    m = re.fullmatch(r"v(\d+)", operand)
    if m:
        return (int(m.group(1)),)
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return tuple(range(int(m.group(1)), int(m.group(2)) + 1))
    return ()


def written(mn, ops):
    """VGPRs an instruction writes (conservative for the few two-destination forms)."""
    if mn.startswith(("v_swap", "v_permlane16_swap", "v_permlane32_swap")):
        return set(r for o in ops[:2] for r in vregs(o))
    if mn.startswith(("v_", "ds_read", "ds_load", "global_load", "flat_load", "scratch_load")):
        return set(vregs(ops[0])) if ops else set()
    if mn.startswith("buffer_load") and "lds" not in ops:
        return set(vregs(ops[0])) if ops else set()
    if mn.startswith(("buffer_atomic", "global_atomic", "flat_atomic")) and ("sc0" in ops or "glc" in ops):
        return set(vregs(ops[0])) if ops else set()
    return set()


def disassemble(so_path):
    """-> {function: [(addr, mnemonic, operand words, text, branch target addr or None)]} over all gfx950 code objects."""
    funcs = {}
    for n_obj, elf in enumerate(code_objects(so_path)):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf)
            f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        func, start = "?", 0
        for line in txt.splitlines():
            m = re.match(r"^([0-9a-f]+) <(.+)>:", line)
            if m:
                start, func = int(m.group(1), 16), f"{m.group(2)} [{n_obj}]"
                funcs.setdefault(func, [])
                continue
            code, _, comment = line.partition("//")
            s = code.strip()
            if not s or s.endswith(":"):
                continue
            parts = s.split(None, 1)
            mn = parts[0]
            ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", parts[1])] if len(parts) > 1 else []
            ops = [w for o in ops for w in o.split()]          # 'v0 offen offset:16' -> separate words
            ma = re.match(r"\s*([0-9A-Fa-f]+):", comment)
            addr = int(ma.group(1), 16) if ma else None
            mt = re.search(r"\+0x([0-9a-fA-F]+)>", comment) if mn.startswith(("s_cbranch", "s_branch")) else None
            funcs.setdefault(func, []).append((addr, mn, ops, s, start + int(mt.group(1), 16) if mt else None))
    return funcs


def store_hazards(ins):
    """Wide stores of one function whose data registers are written within NEED issue slots, along EVERY path: a
    conditional branch costs one slot and is followed both ways (fall-through and target), `s_branch` to its target --
    that a taken branch costs at least one slot is all that is assumed."""
    index = {a: i for i, (a, *_r) in enumerate(ins) if a is not None}
    bad = []
    for k, (_a, mn, ops, s, _t) in enumerate(ins):
        if not re.match(r"(buffer|global|flat|scratch)_store_(dwordx[34]|b96|b128)", mn):
            continue
        data = set(vregs(ops[0])) if mn.startswith("buffer") else set(vregs(ops[1])) if len(ops) > 1 else set()
        work, seen, hit = [(k + 1, 0)], set(), None
        while work and hit is None:
            j, states = work.pop()
            while states < NEED and j < len(ins) and (j, states) not in seen:
                seen.add((j, states))
                _a2, mn2, ops2, s2, tgt = ins[j]
                if mn2 in ("s_endpgm", "s_setpc_b64"):
                    break
                if mn2.startswith("s_cbranch") or mn2 == "s_branch":
                    if tgt in index:
                        work.append((index[tgt], states + 1))
                    if mn2 == "s_branch":
                        break
                    states, j = states + 1, j + 1
                    continue
                if written(mn2, ops2) & data:
                    hit = s2
                    break
                states += int(ops2[0], 0) + 1 if mn2 == "s_nop" and ops2 else 1
                j += 1
        if hit is not None:
            bad.append((s, hit))
    return bad


def scan(so_path):
    """-> list of (function, store line, offending line)."""
    return [(fn, a, b) for fn, ins in disassemble(so_path).items() for a, b in store_hazards(ins)]


def counted_wait_hazards(ins):
    """A kernel that waits with a COUNTED `s_waitcnt vmcnt(n > 0)` knows how many vector-memory instructions are younger
    than the load it waits for.  A register spill (`scratch_store` / `scratch_load`) in the same loop is one more such
    instruction than it counted.  That cannot let the wait pass EARLY -- loads return in order, so while the awaited load is
    outstanding so are the n counted younger ones, n + 1 > n -- but it makes the wait (and the compiler's own vmcnt(0) behind a
    scratch_load) wait for more than was meant: ADVISORY, reported and not failed on.  (Today: the wide conv_bal_bf16 variants
    reload one spilled address register per chunk.)
    -> list of (scratch instruction, counted wait) where the spill sits in the wait's INNERMOST loop (the span of a backward
    branch), i.e. runs every time the counting does; a spill in an outer loop (once per tile or chunk) is not flagged."""
    loops = [(t, a) for a, mn, _o, _s, t in ins if t is not None and a is not None and t <= a]
    waits = [(a, s) for a, mn, ops, s, _t in ins if mn == "s_waitcnt" and a is not None and
             any(re.fullmatch(r"vmcnt\((\d+)\)", w) and int(w[6:-1]) > 0 for w in ops)]
    spills = [(a, s) for a, mn, _o, s, _t in ins if mn.startswith("scratch_") and a is not None]
    out = []
    for wa, ws in waits:
        inner = min(((hi - lo, lo, hi) for lo, hi in loops if lo <= wa <= hi), default=None)   # the wait's innermost loop
        if inner:
            out += [(ss, ws) for sa, ss in spills if inner[1] <= sa <= inner[2]]
    return out


def scan_counted_waits(so_path):
    """-> list of (function, scratch instruction, counted wait)."""
    out = []
    for fn, ins in disassemble(so_path).items():
        out += [(fn, a, b) for a, b in sorted(set(counted_wait_hazards(ins)))]
    return out


def permlane_swap_groups(ins):
    """v_permlane32_swap_b32 / v_permlane16_swap_b32 of one function as runs of consecutive swaps:
    -> list of (first swap's text, number of swaps in the run, wait states directly in front of the run).
    csrc/common.h::acc32_regs converts a 32 x 32 tile's accumulators with ONE asm statement of 8 swaps behind `s_nop 15` +
    `s_nop 3` (20 wait states: the matrix pipe's write -> VALU read distance, which hipcc does not pad for asm operands).  A run
    of another length or with a shorter pad did not come from that helper -- e.g. from the builtins, which hipcc 7.2
    miscompiles (second result dropped, calls merged: profiles/r06_experiments.txt 1b)."""
    out, i = [], 0
    while i < len(ins):
        if "permlane32_swap" in ins[i][1] or "permlane16_swap" in ins[i][1]:
            j = i
            while j < len(ins) and ("permlane32_swap" in ins[j][1] or "permlane16_swap" in ins[j][1]):
                j += 1
            pad, k = 0, i - 1
            while k >= 0 and ins[k][1] == "s_nop":
                pad += int(ins[k][2][0]) + 1
                k -= 1
            out.append((ins[i][3], j - i, pad))
            i = j
        else:
            i += 1
    return out


def scan_permlane_swaps(so_path):
    """-> (number of well-formed groups, list of (function, first swap, swaps in the run, wait states in front) that are not)."""
    good, bad = 0, []
    for fn, ins in disassemble(so_path).items():
        for text, n, pad in permlane_swap_groups(ins):
            if n == 8 and pad >= 18:
                good += 1
            else:
                bad.append((fn, text, n, pad))
    return good, bad


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "poserisk_release_amd", "libposerisk_hip.so")
    objs = code_objects(path)
    bad = scan(path)
    print(f"{path}: {len(objs)} gfx950 code objects, {len(bad)} wide stores with a data register written within {NEED} issue slots")
    for fn, a, b in bad:
        print(f"  {fn[:70]}\n      {a}\n      {b}")
    spill = scan_counted_waits(path)
    print(f"advisory: {len(spill)} register spills in a loop that waits with a counted vmcnt (a performance smell, not a hazard: see "
          f"counted_wait_hazards)")
    for fn, a, b in spill:
        print(f"  {fn[:70]}\n      {a}\n      {b}")
    sys.exit(1 if bad else 0)
