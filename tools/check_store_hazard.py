#!/usr/bin/env python3
"""
Static check of the built library for the gfx950 store-data hazard (profiles/r04_store_data_hazard_plain.txt): an
8- / 12- / 16-byte VMEM store whose data registers a VALU instruction rewrites within the next WINDOW instructions.
hipcc leaves that unprotected when the store's soffset is an SGPR (16 bytes) and always for 8 bytes; the hardware
reads the first data register late, so the overwrite can overtake the store.  The epilogues pin their data registers
(conv3d_epilogue.h epi_store_*); this looks at EVERY kernel of libddpm3d.so, so that a new store site or a compiler
change cannot bring the pattern back unnoticed (tests/test_host_cpu.py runs it).

    python tools/check_store_hazard.py [path/to/libddpm3d.so]      # exit 1 and a listing if any site is found
"""

import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
WINDOW = 2          # instructions behind the store that must not write its data registers

# (scratch_store: the compiler's own spill stores -- fallback paths only -- are rewritten behind the store in every
# program hipcc builds; they are listed with --all, not counted)
STORE = re.compile(r"^\s*(buffer_store_dwordx[234]|global_store_dwordx[234]|flat_store_dwordx[234]|scratch_store_dwordx[234])\s+(.*)$")
VREG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def code_objects(so, tmp):
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so, os.path.join(tmp, "fat.bin")], check=True)
    data = open(os.path.join(tmp, "fat.bin"), "rb").read()
    out = []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data):
        p = m.start()
        cnt = struct.unpack_from("<Q", data, p + 24)[0]
        q = p + 32
        for _ in range(cnt):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            q += 24
            tgt = data[q:q + tl].decode()
            q += tl
            if "gfx950" in tgt and size:
                f = os.path.join(tmp, "co_%d.o" % len(out))
                open(f, "wb").write(data[p + off:p + off + size])
                out.append(f)
    return out


def data_regs(mnemonic, operands):
    """register range of the store's data operand"""
    ops = [o.strip() for o in operands.split(",")]
    # buffer_store: vdata, vaddr, srsrc, soffset ...; global/flat/scratch_store: vaddr, vdata, saddr
    data = ops[0] if mnemonic.startswith("buffer") else ops[1]
    m = VREG.fullmatch(data)
    if not m:
        return None
    return (int(m.group(1)), int(m.group(2))) if m.group(1) else (int(m.group(3)),) * 2


def written_regs(line):
    t = line.strip()
    m = re.match(r"^(v_\S+)\s+([^,]+)", t)
    if not m or m.group(1).startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_nop")):
        return None
    r = VREG.fullmatch(m.group(2).strip())
    if not r:
        return None
    return (int(r.group(1)), int(r.group(2))) if r.group(1) else (int(r.group(3)),) * 2


def scan(listing):
    sites = []
    kernel = "?"
    lines = listing.split("\n")
    insts = []          # (kernel, text)
    for ln in lines:
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            kernel = m.group(1)
            continue
        t = ln.split("//")[0].strip()
        if t and not t.endswith(":"):
            insts.append((kernel, t))
    for i, (k, t) in enumerate(insts):
        m = STORE.match(t)
        if not m:
            continue
        rng = data_regs(m.group(1), m.group(2))
        if rng is None:
            continue
        for j in range(1, WINDOW + 1):
            if i + j >= len(insts) or insts[i + j][0] != k:
                break
            nxt = insts[i + j][1]
            if nxt.startswith(("s_nop", "s_waitcnt")):          # an explicit wait state in between ends the window
                break
            if nxt.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
                break
            w = written_regs(nxt)
            if w and w[0] <= rng[1] and w[1] >= rng[0]:
                sites.append((k, t, nxt, j))
                break
    return sites


def main():
    args = [a for a in sys.argv[1:] if a != "--all"]
    everything = "--all" in sys.argv[1:]
    so = args[0] if args else os.path.join(ROOT, "3d-denoising-diffusion-model_amd", "csrc", "libddpm3d.so")
    with tempfile.TemporaryDirectory() as tmp:
        sites = []
        n = 0
        for f in code_objects(so, tmp):
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f], capture_output=True, text=True, check=True).stdout
            n += len(re.findall(r"^[0-9a-f]+ <.+>:$", dis, flags=re.M))
            sites += [x for x in scan(dis) if everything or not x[1].startswith("scratch_store")]
    print("%s: %d kernels, %d wide stores whose data registers are rewritten within %d instructions" % (os.path.basename(so), n, len(sites), WINDOW))
    for k, st, nx, j in sites[:40]:
        print("  %s\n      %s\n      +%d: %s" % (k, st, j, nx))
    return 1 if sites else 0


if __name__ == "__main__":
    sys.exit(main())
