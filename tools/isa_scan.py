#!/usr/bin/env python3
"""Scan the compiler's output of the HIP sources for the patterns that cost time this round:

  * scalar (kernarg) re-loads inside loops        -- `s_load_dword*` between a label and its backward branch
  * an LDS round trip in front of an MFMA          -- `s_waitcnt lgkmcnt(0)` directly before a `v_mfma`
  * register spills                                -- `private_seg_size` > 0
  * register counts (occupancy)

    python tools/isa_scan.py [file.hip ...]        (default: every csrc/*.hip; needs hipcc, no GPU)
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "maskcyclegan-vc_amd", "csrc")


def scan(asm):
    rows = []
    for m in re.finditer(r"^(_Z\S+):\s*; @.*?\n(.*?)s_endpgm", asm, re.S | re.M):
        name, body = m.group(1), m.group(2).split("\n")
        labels = {mm.group(1): i for i, l in enumerate(body) for mm in [re.match(r"^(\.LBB\d+_\d+):", l)] if mm}
        inloop = set()
        for i, l in enumerate(body):
            mm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
                inloop.update(range(labels[mm.group(1)], i + 1))
        loads = sum(1 for k in inloop if re.search(r"\bs_load_dword", body[k]))
        # (r6) vector-memory loads inside loops and full drains (`s_waitcnt vmcnt(0)`) inside loops: a register-prefetch ring that the compiler
        # drains at the loop's back edge, or a conditional load after which it stops counting (bf16_trunk_layer_kernel's first form)
        vml = sum(1 for k in inloop if re.search(r"\b(global_load|buffer_load|scratch_load)", body[k]) and "lds" not in body[k])
        vm0 = sum(1 for k in inloop if re.search(r"s_waitcnt.*vmcnt\(0\)", body[k]))
        ins = [l.strip() for l in body if l.strip() and not l.strip().startswith((";", "."))]
        mfma = sum(l.startswith("v_mfma") for l in ins)
        w0 = sum(1 for i, l in enumerate(ins) if l.startswith("v_mfma") and i and ins[i - 1].startswith("s_waitcnt") and "lgkmcnt(0)" in ins[i - 1])

        def field(key):
            mm = re.search(re.escape(name) + r"\." + key + r", (\d+)", asm)
            return int(mm.group(1)) if mm else -1
        short = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)
        short = re.sub(r"Ev4Twin.*|Ev12Bf16.*", "", short)[:64]
        rows.append((loads, w0, mfma, field("private_seg_size"), field("num_vgpr"), field("num_agpr"), short, vml, vm0))
    return rows


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    full = os.environ.get("ISA_SCAN_ALL", "0") != "0"
    print("%-66s %9s %12s %8s %6s %6s %12s" % ("kernel", "loop s_ld", "mfma|wait0", "scratch", "vgpr", "agpr", "loop ld|vm0"))
    for f in files:
        with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
            r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I", CSRC, f,
                                "-o", tmp.name], capture_output=True, text=True)
            if r.returncode:
                print("%s: %s" % (f, r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "failed"))
                continue
            rows = scan(open(tmp.name).read())
        for loads, w0, mfma, scr, vg, ag, short, vml, vm0 in sorted(rows, reverse=True):
            if full or loads or scr > 0 or (mfma and w0 * 2 >= mfma) or vm0:
                print("%-66s %9d %6d|%-5d %8d %6d %6d %6d|%-5d" % (short, loads, mfma, w0, scr, vg, ag, vml, vm0))


if __name__ == "__main__":
    main()
