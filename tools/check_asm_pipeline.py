#!/usr/bin/env python
"""Static safety check of the hand-counted load pipelines (csrc/lfdm_device.h: lfdm_gload_f4 / lfdm_vmwait).

The loads of those pipelines are issued by inline asm; hipcc believes their destination registers hold the data the moment the
asm statement ends.  Under register pressure it may therefore copy (spill to an AGPR) or even re-use such a register while the
load is still in flight - the copy is stale, and a re-used register is overwritten when the data lands (seen once: a pointer
temporary clobbered by a late load -> HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION).  This script compiles a .hip file to gfx950
assembly and walks every kernel: an asm-issued load is pending until a `s_waitcnt vmcnt(N)` retires it (loads return in order:
a wait leaves the N newest pending); NO other instruction may name a pending destination register, and nothing may be pending
at a label / branch.  Exit status 1 and a report per violation.  Run by tests/test_host_and_abi.py (no GPU needed)."""
import os
import re
import subprocess
import sys
import tempfile

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form", "-S", "--cuda-device-only"]
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check(path):
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.run([HIPCC] + FLAGS + [path, "-o", asm], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lines = open(asm).read().splitlines()
    problems, kernel, pending, in_asm, n_loads = [], None, [], False, 0
    for no, raw in enumerate(lines, 1):
        line = raw.split(";")[0].strip() if not raw.strip().startswith(";;#") else raw.strip()
        if raw.strip().startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if raw.strip().startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line:
            continue
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel, pending = m.group(1), []
            continue
        if kernel is None:
            continue
        if line.startswith("s_endpgm"):
            if pending:
                problems.append((kernel, no, "loads still pending at s_endpgm", raw.strip()))
            kernel = None
            continue
        if re.match(r"^\.?\w+:$", line) or line.startswith("s_cbranch") or line.startswith("s_branch"):
            if pending:
                problems.append((kernel, no, "asm loads pending across control flow", raw.strip()))
                pending = []
            continue
        w = re.match(r"s_waitcnt\b(.*)", line)
        if w:
            v = re.search(r"vmcnt\((\d+)\)", w.group(1))
            if v:
                keep = int(v.group(1))
                pending = pending[len(pending) - keep:] if keep < len(pending) else pending
            continue
        if in_asm and line.startswith("global_load_dwordx4"):
            dst = regs_of(line.split(",")[0])
            busy = set().union(*[p for p in pending]) if pending else set()
            if dst & busy:
                problems.append((kernel, no, "asm load into a register that is still pending", raw.strip()))
            pending.append(dst)
            n_loads += 1
            continue
        if pending:
            busy = set().union(*pending)
            touched = regs_of(line)
            if not in_asm and re.match(r"(global|buffer|flat|scratch)_load", line):
                # a compiler-issued load joins the in-order queue; conservatively treat its destination as pending too
                pending.append(regs_of(line.split(",")[0]))
                touched = regs_of(",".join(line.split(",")[1:]))
            if touched & busy:
                problems.append((kernel, no, "instruction names a register of an in-flight asm load: v%s" % sorted(touched & busy)[:4], raw.strip()))
    return problems, n_loads


def main():
    rc = 0
    for path in sys.argv[1:]:
        problems, n = check(path)
        print("%s: %d asm-issued loads, %d problem(s)" % (os.path.basename(path), n, len(problems)))
        for k, no, what, text in problems[:40]:
            print("  %s line %d: %s\n      %s" % (k[:60], no, what, text))
        rc |= 1 if problems else 0
    return rc


if __name__ == "__main__":
    sys.exit(main())
