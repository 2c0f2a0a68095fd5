#!/usr/bin/env python3
"""Audit of the device ISA for one hipcc defect found in round 4: register copies of a live-range split placed
at the top of a join block IN FRONT of the `s_or_b64 exec, exec, sN` that re-enables the lanes which skipped the branch -- the copies
then run for the lanes of the branch only and the other lanes are left with whatever the destination held (letkf_wave_kernel<16, 11,
true>: the run scheduler's `pend` counter, garbage in lanes 32..63, the wave walked on into points that were not its own).

Usage: isa_exec_audit.py file.s [...]   (the device assembly: hipcc -save-temps=obj, or --cuda-device-only -S).  Prints every JOIN
block -- a label that the skip branch (s_cbranch_execz) of a divergent `if` jumps to -- in which an instruction that depends on exec
(vector ALU, LDS, global, scratch) stands between the label and the block's exec restore; exit code 1 if there is one.  (A then-block
that merely falls through into the restore is not a join block: its instructions are meant for its own lanes.)"""
import os
import re
import sys

LABEL = re.compile(r"^([.\w$]+):")
# the exec restores of a join block: after an `if` (s_or_b64 exec, exec, saved) and in front of an `else` (s_or_saveexec_b64 sN, sM);
# `s_or_saveexec_b64 sN, -1` is whole-wave mode around an SGPR-spill register, not a join
RESTORE = re.compile(r"^\s+(s_or_b64 exec, exec, |s_or_saveexec_b64 s\[\d+:\d+\], s\[)")
EXEC_FREE = ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32")


def join_labels(path):
    """per kernel: the labels a skip branch of a divergent `if` jumps to (s_cbranch_execz after s_and_saveexec): join blocks"""
    tg, kernel = {}, None
    with open(path) as f:
        for line in f:
            m = LABEL.match(line)
            if m and not m.group(1).startswith(".L"):
                kernel = m.group(1)
            m = re.match(r"\s+s_cbranch_execz (\S+)", line)
            if m:
                tg.setdefault(kernel, set()).add(m.group(1))
    return tg


def audit(path):
    joins = join_labels(path)
    return [h for h in audit_all(path) if h[1] in joins.get(h[0], ())]


def audit_all(path):
    hits = []
    kernel = None
    since = []          # instructions since the last label
    label = None
    with open(path) as f:
        for no, line in enumerate(f, 1):
            code = line.split(";")[0].rstrip()
            if not code.strip():
                continue
            m = LABEL.match(code)
            if m:
                label = m.group(1)
                if not label.startswith(".L"):
                    kernel = label
                since = []
                continue
            if not code.startswith("\t") or code.strip().startswith("."):
                continue
            if RESTORE.match(code):
                bad = [s for s in since if (s[1].startswith(("v_", "ds_", "global_", "scratch_", "buffer_", "flat_")) and not s[1].startswith(EXEC_FREE))]
                # only the FIRST restore after a label is the block's prologue; later ones close regions opened inside the block
                if bad and not any(RESTORE.match("\t" + s[1]) for s in since):
                    hits.append((kernel, label, no, bad))
                since.append((no, code.strip()))
                continue
            op = code.strip()
            # a branch or a saveexec ends the prologue: what follows belongs to the block's own regions
            if op.startswith(("s_cbranch", "s_branch", "s_and_saveexec", "s_or_saveexec", "s_andn2_saveexec")) or "exec" in op.split(" ")[1:2]:
                since.append((no, "s_or_b64 exec, exec, (end of prologue)"))   # sentinel: stops the check for this block
                continue
            since.append((no, op))
    return hits


def audit_loop_exits(path):
    """the same defect behind a loop: exec-dependent instructions between the loop's back branch (s_cbranch_execnz) and the exec
    restore of the exit block would run with no lane enabled"""
    hits, prev, kernel = [], [], None
    with open(path) as f:
        for no, line in enumerate(f, 1):
            code = line.split(";")[0].rstrip()
            if not code.strip():
                continue
            m = LABEL.match(code)
            if m:
                if not m.group(1).startswith(".L"):
                    kernel = m.group(1)
                prev = []
                continue
            op = code.strip()
            if not code.startswith("\t") or op.startswith("."):
                continue
            if RESTORE.match(code):
                bad = []
                for n, x in reversed(prev):
                    if x.startswith("s_cbranch_execnz"):
                        if bad:
                            hits.append((kernel, "(loop exit)", no, bad[::-1]))
                        break
                    if x.startswith(("s_cbranch", "s_branch")) or "exec" in x:
                        break
                    if x.startswith(("v_", "ds_", "global_", "scratch_", "buffer_", "flat_")) and not x.startswith(EXEC_FREE):
                        bad.append((n, x))
            prev.append((no, op))
    return hits


def main():
    args = [a for a in sys.argv[1:] if a != "-q"]
    quiet = len(args) != len(sys.argv) - 1
    if len(args) == 2 and args[0] == "--dir":
        # every unit's assembly that is present beside the objects: a unit without it was not compiled in this tree (its object
        # arrived built -- gpurun ships objects, not their assembly -- and was audited where it was built)
        import glob
        args = sorted(glob.glob(os.path.join(args[1], "*-hip-amdgcn-amd-amdhsa-gfx950.s")))
        if not args:
            print("isa_exec_audit: no device assembly beside the objects (built elsewhere): nothing to audit")
            return 0
    rc = 0
    total = 0
    for p in args:
        hits = audit(p) + audit_loop_exits(p)
        total += len(hits)
        if hits or not quiet:
            print(f"{p}: {len(hits)} suspicious block prologue(s)")
        for kernel, label, no, bad in hits:
            rc = 1
            print(f"  {kernel} {label} (restore at line {no}):")
            for n, s in bad[:8]:
                print(f"      {n}: {s}")
    print(f"isa_exec_audit: {len(args)} unit(s), {total} suspicious block prologue(s)")
    return rc


if __name__ == "__main__":
    sys.exit(main())
