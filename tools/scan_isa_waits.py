#!/usr/bin/env python
"""Static scan of the gfx950 ISA hipcc emits for csrc/*.hip for the two wait signatures that cost this project time in round 6 (no GPU needed):

  header   a loop whose FIRST wait is `s_waitcnt vmcnt(0)` although the loop issues global / buffer loads: every trip starts by draining all loads in
           flight - the prefetch the source meant is gone (conv_wino.hip before round 6: the prologue requested the filter fragments before the patch, the
           loop body the other way round, and the compiler's wait where both orders merge was the conservative one: 2.4 ms per video).
  lone     a global / buffer load with no other load in the three instructions before it and an `s_waitcnt vmcnt(0)` within the four after it - one load,
           one round trip.  Several of them in one kernel, usually behind `s_cbranch_execz`, are guarded loads (`if (ok) v = *p;`) that the compiler
           could not batch: the linear attention's token rows (eight round trips in a row per tile, 1.4 ms per video), the attention backward's bias and
           rotary loads (22 % of that kernel), conv_smalln's staging loop.

It reports candidates, not verdicts: the same signature measured neutral where other waves of the workgroup cover the round trips (HISTORY.md, round 6).
Usage: tools/scan_isa_waits.py [file.hip ...]      (default: every csrc/*.hip; ~10 s per file)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form", "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include")]
LOAD = re.compile(r"\b(global|buffer)_load")


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:110] or name
    except OSError:
        return name


def functions(lines):
    marks = [(m.group(1), i) for i, l in enumerate(lines) for m in [re.match(r"^(_Z\w+):", l)] if m]
    marks.append((None, len(lines)))
    for (name, a), (_, b) in zip(marks, marks[1:]):
        yield name, [l for l in lines[a:b] if l.strip() and not l.strip().startswith(";")]


def scan(path):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        r = subprocess.run([HIPCC] + FLAGS + ["-o", out, path], capture_output=True, text=True)
        if r.returncode != 0:
            print("%s: hipcc failed\n%s" % (path, r.stderr[-400:]))
            return
        lines = open(out).read().split("\n")
    base = os.path.basename(path)
    for name, body in functions(lines):
        header, lone, guarded = [], 0, 0
        for j, l in enumerate(body):
            if "Loop Header" in l:
                label = l.split(":")[0].strip()
                # the loop's blocks: the header's own and every block LLVM annotates "in Loop: Header=<this>" (a rotated loop's latch stands BEFORE its header)
                tag = "Header=" + label.lstrip(".L")
                starts = [k for k, x in enumerate(body) if x.startswith(".LBB") and (k == j or tag in x)]
                labels = [k for k, x in enumerate(body) if x.startswith(".LBB")] + [len(body)]
                seg = []
                for k in starts:
                    seg += body[k:min(n for n in labels if n > k)]
                first_wait = next((x for x in body[j + 1:j + 8] if "vmcnt" in x or LOAD.search(x) or "v_mfma" in x or "ds_" in x), "")
                if len(starts) >= 1 and "vmcnt(0)" in first_wait and any(LOAD.search(x) for x in seg):
                    header.append("%s (%d instructions, %d loads, %d MFMAs)" % (label, len(seg), sum(bool(LOAD.search(x)) for x in seg), sum("v_mfma" in x for x in seg)))
            if LOAD.search(l) and not any(LOAD.search(x) for x in body[max(0, j - 3):j]):
                nxt = body[j + 1:j + 5]
                if any("vmcnt(0)" in x for x in nxt) and not any(LOAD.search(x) for x in nxt[:2]):
                    lone += 1
                    guarded += any("s_cbranch_execz" in x for x in body[max(0, j - 4):j])
        if header or lone >= 3:
            print("%-22s %s" % (base, demangle(name)))
            for h in header:
                print("    header: loop %s starts with s_waitcnt vmcnt(0)" % h)
            if lone >= 3:
                print("    lone:   %d loads each followed by vmcnt(0) (%d behind s_cbranch_execz)" % (lone, guarded))


if __name__ == "__main__":
    for f in (sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "cvpr23_lfdm_amd", "csrc", "*.hip")))):
        scan(f)
