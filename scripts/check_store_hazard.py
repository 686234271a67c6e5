"""ISA lint of the SHIPPED library: no vector-memory store of more than 64 bits may be followed within two issue slots by an
instruction that writes one of its data VGPRs.

Why (measured on MI355X, scripts/micro/t_store_hazard.hip -> profiles/r03_t_store_hazard.txt): behind `buffer_store_dwordx4` /
`global_store_dwordx4` the hardware reads the data registers late enough that a VALU write issued right behind the store can
reach memory instead of the stored value -- 0.4 % of the stores with an SGPR soffset, 22 % with a literal one, under
back-pressure; ONE wait state still loses 0.5-1.7 % of the literal-soffset / global forms, TWO are always enough.  hipcc pads one
wait state, and none at all when a buffer store's soffset is a register.  The kernels' own 16-byte buffer stores go through
`buffer_store_b128_sreg` (csrc/common.h), which carries its wait states; this script checks what the compiler scheduled around
every other wide store.

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


def scan(so_path):
    """-> list of (function, store line, offending line)."""
    bad = []
    for elf in code_objects(so_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf)
            f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        func, ins = "?", []
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                func = m.group(1)
                continue
            s = line.split("//")[0].strip()
            if not s or s.endswith(":"):
                continue
            parts = s.split(None, 1)
            mn = parts[0]
            ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", parts[1])] if len(parts) > 1 else []
            ops = [w for o in ops for w in o.split()]          # 'v0 offen offset:16' -> separate words
            ins.append((func, mn, ops, s))
        for k, (fn, mn, ops, s) in enumerate(ins):
            if not re.match(r"(buffer|global|flat|scratch)_store_(dwordx[34]|b96|b128)", mn):
                continue
            data = set(vregs(ops[0])) if mn.startswith("buffer") else set(vregs(ops[1])) if len(ops) > 1 else set()
            states, j = 0, k + 1
            while states < NEED and j < len(ins) and ins[j][0] == fn:
                _, mn2, ops2, s2 = ins[j]
                if mn2 in ("s_endpgm", "s_branch", "s_setpc_b64") or mn2.startswith("s_cbranch"):
                    break                                         # the next block is scanned as written; a taken branch costs more than two slots
                if written(mn2, ops2) & data:
                    bad.append((fn, s, s2))
                    break
                states += int(ops2[0], 0) + 1 if mn2 == "s_nop" and ops2 else 1
                j += 1
    return bad


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "poserisk_release_amd", "libposerisk_hip.so")
    objs = code_objects(path)
    bad = scan(path)
    print(f"{path}: {len(objs)} gfx950 code objects, {len(bad)} wide stores with a data register written within {NEED} issue slots")
    for fn, a, b in bad:
        print(f"  {fn[:70]}\n      {a}\n      {b}")
    sys.exit(1 if bad else 0)
